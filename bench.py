#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its headline config.

  python bench.py --gpus N --steps K --warmup W

A step = one full render of the workload (default: config 3 = 256^3 sigma_t grid + 256^3 linear RIF,
RK4 eikonal stepping on the trilinear field (BRICK27 records), HG g=0.8, 512^2 x 256 spp, half-voxel steps, ratio-tracking
transmittance, box filter) with all inputs resident in HBM.  N > 1 (launched by torch.distributed.run) shards
sample indices across ranks (weak scaling: every rank renders 512^2 x 256 spp of a 512^2 x (256 N) spp job)
and sum-reduces the film with RCCL inside the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (guides/MI355X_MICROARCH.md); ~6300 measured by a float4 copy


def build_workload(name, res, size, spp):
    import numpy as np
    from mitsubaer_amd import params as P, synth
    common = dict(width=size, height=size, rfilter=P.FILTER_BOX, rfilter_param=0.5, tr_estimator=P.TR_RATIO,
                  phase=P.PHASE_HG, g=0.8, density_scale=4.0, albedo=[0.9, 0.9, 0.9])
    dens = synth.density_field(res)
    if name == "cfg2":
        p = P.SceneParams(density=dens, rif_mode=P.RIF_CONST, **common)
        desc = "%d^3 sigma_t grid, constant RIF (straight rays), HG g=0.8, %d^2 x %d spp" % (res, size, spp)
    elif name == "cfg3":
        rres = int(os.environ.get("BENCH_RIF_RES", res))
        p = P.SceneParams(density=dens, rif_mode=P.RIF_TRILINEAR, rif=synth.linear_rif(rres), stepper=P.STEP_RK4,
                          stepsize=0.5 * 2.0 / (res - 1), **common)
        desc = "%d^3 sigma_t grid + %d^3 linear-gradient RIF, RK4 eikonal curved rays, HG g=0.8, %d^2 x %d spp" % (res, res, size, spp)
    elif name == "cfg4":
        p = P.SceneParams(density=dens, rif_mode=P.RIF_TRILINEAR, rif=synth.radial_rif(res), stepper=P.STEP_RK4,
                          stepsize=0.5 * 2.0 / (res - 1), **common)
        desc = "%d^3 sigma_t grid + %d^3 radial RIF, RK4 eikonal curved rays, HG g=0.8, %d^2 x %d spp" % (res, res, size, spp)
    elif name == "cfg5":
        # emissive heterogeneous medium + RGB albedo grid + curved-ray luminaire sampling of a point emitter (A12)
        g = np.linspace(0.0, 1.0, res, dtype=np.float32)
        alb = np.empty((res, res, res, 3), np.float32)
        alb[..., 0] = 0.55 + 0.4 * g[None, None, :]; alb[..., 1] = 0.55 + 0.4 * g[None, :, None]; alb[..., 2] = 0.55 + 0.4 * g[:, None, None]
        common.pop("albedo")
        p = P.SceneParams(density=dens, rif_mode=P.RIF_TRILINEAR, rif=synth.linear_rif(res), stepper=P.STEP_RK4,
                          stepsize=0.5 * 2.0 / (res - 1), albedo_mode=P.ALBEDO_GRID, albedo_grid=alb, env_radiance=[0, 0, 0],
                          emission=[0.2, 0.12, 0.06], point_position=[0.2, 0.3, -0.1], point_intensity=[1.0, 0.8, 0.5], **common)
        desc = "%d^3 sigma_t + RGB albedo grids, emissive medium, %d^3 linear RIF, RK4 curved rays + curved-ray point-emitter NEE, %d^2 x %d spp" % (res, res, size, spp)
    else:
        raise SystemExit("unknown workload %s" % name)
    return p, desc


def algorithmic_bytes(c, p):
    """B_alg = 4*[T_rif*E*C_steps + 8*C_tent + 24*C_real*[albedo gridded]] + 40*C_paths."""
    from mitsubaer_amd import capi, params as P
    t_rif = 0 if p.rif_mode == P.RIF_CONST else (8 if p.rif_mode == P.RIF_TRILINEAR else 64)
    e = 2 if p.stepper == P.STEP_VERLET else 4
    alb = 24 if p.albedo_mode == P.ALBEDO_GRID else 0
    return 4.0 * (t_rif * e * float(c[capi.C_STEPS]) + 8.0 * float(c[capi.C_TENTATIVE]) + alb * float(c[capi.C_REAL])) \
        + 40.0 * float(c[capi.C_PATHS])


def host_cores():
    """Threads the CPU baseline may use: the affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    return n


def cpu_baseline(p, target_seconds=20.0):
    """The CPU oracle (kind 'port': the restatement of the reference's routines) on a bounded sample of the same
    workload: whole image, few spp, all host cores."""
    from oracle import orc
    cores = host_cores()
    orc.build()
    t0 = time.time()
    orc.render(p, 0, 1, 0, nthreads=cores)                                # calibration: one sample per pixel
    per_spp = max(time.time() - t0, 1e-3)
    spp = int(max(1, min(64, round(target_seconds / per_spp))))
    t0 = time.time()
    _, c = orc.render(p, 0, spp, 0, nthreads=cores)
    dt = time.time() - t0
    paths = p.width * p.height * spp
    return {"value": paths / dt / 1e6, "unit": "Mpaths/s", "cores": cores, "kind": "port",
            "sample": "%dx%d x %d spp of the same scene (%.1f s), oracle/libmer_oracle.so fp32, std::thread over rows" % (p.width, p.height, spp, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--layout", default="auto", choices=["auto", "dense", "cell8", "brick27", "brick125"],
                    help="HBM layout of the trilinear RIF (auto: brick27 up to 2^28 nodes, cell8 above)")
    ap.add_argument("--shard", default="samples", choices=["samples", "tiles"])
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--device", type=int, default=None, help="force this GPU index for every rank (rehearsal on one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    args = ap.parse_args()

    import numpy as np
    import torch
    from mitsubaer_amd import capi, dist as mdist
    rank, world, local = mdist.init_process_group(args.backend)
    if args.device is not None:
        local = args.device
    if world != args.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    p, desc = build_workload(args.workload, args.res, args.size, args.spp)
    ctx = capi.Context(local)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    if args.layout == "auto":
        args.layout = "brick27" if args.res ** 3 <= 1 << 28 else "cell8"
    layout = {"cell8": capi.LAYOUT_CELL8, "brick27": capi.LAYOUT_BRICK27, "brick125": capi.LAYOUT_BRICK125, "dense": capi.LAYOUT_DENSE}[args.layout]
    sc, vols = ctx.upload_scene(p, layout=layout)
    film = torch.zeros((p.height, p.width, 5), dtype=torch.float32, device=dev)
    # weak scaling: every rank renders args.spp samples per pixel of a (spp * world)-sample job
    sh = mdist.shard_args(args.shard, rank, world, args.spp * world) if args.shard == "samples" else \
        mdist.shard_args(args.shard, rank, world, args.spp * world)

    def step(seed):
        film.zero_()
        ctx.render(sc, film.data_ptr(), sh["spp_begin"], sh["spp_count"], seed=seed, spp_stride=sh["spp_stride"],
                   tile_rank=sh["tile_rank"], tile_count=sh["tile_count"])
        mdist.reduce_film(film)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(1000 + i)
    barrier()
    ctx.counters_reset()
    kernel_ms = []; march_ms = []; event_ms = []; passes = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
        # HIP events recorded by libmer around the render kernel on the launch stream; reading them waits for
        # this step's kernel only (steps are dependent through the film anyway)
        kernel_ms.append(ctx.last_kernel_ms())
        n_, m_, e_ = ctx.last_render_stats()
        passes.append(n_); march_ms.append(m_); event_ms.append(e_)
    barrier()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(tmax.item())
    counters = ctx.counters().astype(np.float64)
    paths_rank = float(counters[capi.C_PATHS])
    # one more step OUTSIDE the timed region with a single pipeline: the dominant kernel's launch duration when its launches do not
    # share the chip with those of other pipelines (reported as roofline.single_pipeline, next to the figures of the timed region)
    solo = None
    if ctx.get_option("pipes") > 1 and not os.environ.get("BENCH_NO_SOLO_STEP"):
        with ctx.options(pipes=1):
            ctx.counters_reset()
            t1 = time.perf_counter(); step(args.steps); torch.cuda.synchronize(); solo_wall = time.perf_counter() - t1
            n1, m1, e1 = ctx.last_render_stats()
            c1 = ctx.counters().astype(np.float64)
            solo = (n1, m1, e1, c1, solo_wall)
    ct = torch.tensor(counters, dtype=torch.float64, device=dev)
    mdist.reduce_counters(ct)
    total_paths = float(ct[capi.C_PATHS].item())

    if rank == 0:
        k_ms = float(np.mean(kernel_ms))                                  # whole render (all wavefront passes)
        m_ms = float(np.mean(march_ms)); e_ms = float(np.mean(event_ms)); n_pass = float(np.mean(passes))
        b_alg = algorithmic_bytes(counters, p) / max(args.steps, 1)      # per step (one full render), this rank
        # the dominant kernel is K_march: it performs every field fetch; the film write (40 B/path) is K_event's
        b_march = (b_alg - 40.0 * paths_rank / max(args.steps, 1))
        per_launch_bytes = b_march / max(n_pass, 1.0)
        per_launch_ms = m_ms / max(n_pass, 1.0)
        achieved = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0     # 0: MER_NO_PASS_EVENTS A/B runs
        achieved_step = b_alg / (k_ms * 1e-3) / 1e9
        lane_eff = float(counters[capi.C_ACTIVE_LANES] / max(counters[capi.C_LOOP_ITERS], 1.0))
        name, cus, hbm = ctx.device_info()
        out = {
            "metric": "Mpaths/sec (curved-ray heterogeneous volume); roofline = achieved algorithmic HBM GB/s vs peak",
            "value": total_paths / elapsed / 1e6, "unit": "Mpaths/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": {"cfg2": "configs[1]: ", "cfg3": "configs[2]: ", "cfg4": "configs[3]: ", "cfg5": "configs[4]: "}[args.workload] + desc, "grid": args.res, "film": [p.width, p.height], "spp_per_gpu": args.spp, "pipelines_per_gpu": ctx.get_option("pipes"),
                       "stepper": "rk4", "rif_interp": "trilinear", "layout": args.layout, "shard": args.shard,
                       "stepsize": p.stepsize, "estimator": "volpath + delta tracking on eikonal rays, ratio-tracking NEE",
                       "device": name, "cus": cus},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel": "mer::march_kernel<curved, %s trilinear, rk4, grid>" % args.layout,
                         "kernel_avg_launch_ms": per_launch_ms, "launches_per_step": n_pass, "algorithmic_bytes_per_launch": per_launch_bytes,
                         "kernel_ms_per_step": m_ms, "event_kernel_ms_per_step": e_ms,
                         "whole_step": {"ms": k_ms, "algorithmic_bytes": b_alg, "achieved": achieved_step, "frac": achieved_step / HBM_PEAK_GBS},
                         "counters_per_launch": {"paths": paths_rank / args.steps, "steps": counters[capi.C_STEPS] / args.steps,
                                                 "tentative": counters[capi.C_TENTATIVE] / args.steps, "real": counters[capi.C_REAL] / args.steps},
                         "active_lane_fraction": lane_eff,
                         # the render runs as several independent pipelines on their own streams (launch_render): launches of different
                         # pipelines execute side by side, so one launch's duration is stretched by its neighbours and `achieved` (bytes of
                         # ONE launch / its duration, the rocprofv3 figure) understates the chip's rate by up to that factor;
                         # whole_step is the aggregate: all algorithmic bytes of the step / its wall time
                         "concurrent_pipelines": ctx.get_option("pipes")},
        }
        if solo is not None:
            n1, m1, e1, c1, solo_wall = solo
            b1 = algorithmic_bytes(c1, p) - 40.0 * float(c1[capi.C_PATHS])
            out["roofline"]["single_pipeline"] = {
                "what": "one extra untimed step with MER_PIPES=1: K_march launches that have the chip to themselves",
                "kernel_avg_launch_ms": m1 / max(n1, 1), "launches_per_step": n1, "algorithmic_bytes_per_launch": b1 / max(n1, 1),
                "achieved": b1 / max(n1, 1) / (m1 / max(n1, 1) * 1e-3) / 1e9 if m1 > 0 else 0.0,
                "frac": (b1 / max(n1, 1) / (m1 / max(n1, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS) if m1 > 0 else 0.0,
                "kernel_ms_per_step": m1, "event_kernel_ms_per_step": e1, "step_wall_ms": solo_wall * 1e3,
                "active_lane_fraction": float(c1[capi.C_ACTIVE_LANES] / max(c1[capi.C_LOOP_ITERS], 1.0))}
        # HBM traffic cannot be counted inside this process: it comes from separate `rocprofv3 --pmc` passes of this very
        # command (FETCH_SIZE / WRITE_SIZE, gfx950 read correction applied), committed under profiles/
        tf = os.path.join(ROOT, "profiles", "round1", "pmc_traffic_cfg3_n1.json")
        if args.workload == "cfg3" and args.res == 256 and args.size == 512 and args.spp == 256 and args.layout == "brick27" and os.path.exists(tf):
            try:
                t = json.load(open(tf))["march_kernel"]
                out["roofline"]["traffic"] = t["traffic_bytes_per_launch_with_gfx950_x2_read_correction"]
                out["roofline"]["traffic_source"] = "profiles/round1/pmc_traffic_cfg3_n1.json (separate rocprofv3 --pmc passes; bytes per K_march launch)"
            except Exception:
                pass
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(p, args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
