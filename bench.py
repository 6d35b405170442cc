#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its headline config.

  python bench.py --gpus N --steps K --warmup W

A step = one full render of the workload (default: configs[2] = 256^3 sigma_t grid + 256^3 linear RIF, RK4 eikonal stepping on the
trilinear field, HG g=0.8, 512^2 x 256 spp, half-voxel steps, ratio-tracking transmittance, box filter) with all inputs resident in HBM.

N > 1: one process per GPU.  Launched by torch.distributed.run (the driver's way) the ranks come from the environment; launched as
plain `python bench.py --gpus N` this process starts that launcher itself -- N ranks, before it touches a GPU -- and relays rank 0's
line.  A world size that differs from --gpus is an error.  --scaling weak (default): every rank renders --spp samples per pixel of a
(spp x N)-sample job, sample-interleaved.  --scaling strong: the job is fixed (--spp samples per pixel of the whole film) and is cut
into 32x32 image tiles dealt round-robin to the ranks (the reference's block partition, src/librender/renderproc.cpp:79,142-149).
Either way the film is sum-reduced with one RCCL all-reduce inside the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (guides/MI355X_MICROARCH.md); ~6300 measured by a float4 copy
# wave64 VALU instructions the chip can issue per second: 256 CUs x 4 SIMDs x 2.4 GHz / 2.3 cycles per independent v_fma_f32
# (profiles/round1/ubench_valu_issue_rate.txt: 2.25-2.40 cycles measured; a DEPENDENT v_fma_f32 issues every 4.2-4.4 cycles)
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.3
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "round3", "hbm_traffic.json")
PMC_PASSES = {      # one rocprofv3 --pmc pass each (the counters of a pass must fit the hardware's counter slots; gpurun allows --pmc with --kernel-trace only)
    "rd": "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum",
    "write": "WRITE_SIZE TCC_EA0_RDREQ_DRAM_sum",
    "sq": "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY",
}


def source_hash():
    """sha256 over the kernel sources (csrc/*.hpp, *.hip): a committed counter pass is only valid for the kernels it was taken on"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(ROOT, "mitsubaer_amd", "csrc", "*.h*"))):
        h.update(os.path.basename(fn).encode()); h.update(open(fn, "rb").read())
    return h.hexdigest()[:16]


def pmc_sums(outdir, family="march_kernel"):
    """{counter: sum over the dispatches of the kernel family} from the counter_collection CSV of one rocprofv3 pass"""
    import csv
    import glob
    import re
    acc = {}
    files = sorted(glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    n = 0
    for fn in files[-1:]:
        for r in csv.DictReader(open(fn)):
            m = re.match(r"(?:void )?(?:mer::)?([A-Za-z_0-9]+)", r["Kernel_Name"])
            if m and m.group(1) == family:
                acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]); n += 1
    return acc if n else None


def dominant_family(workload):
    """kernel family whose counters the roofline is quoted on: K_march for curved rays; for straight rays K_event, which runs the walks itself
    (option inline_walks, the default) -- unless the two-kernel form is forced"""
    return "event_kernel" if workload == "cfg2" else "march_kernel"


def live_pmc(workload, res, size, spp, layout, options, budget_s=240.0):
    """Counter passes of THIS workload taken in THIS run: for every entry of PMC_PASSES one child process `rocprofv3 --pmc <counters>
    --kernel-trace -- python3 bench.py <one single-pipeline step>`, started before this process touches the GPU (a GPU process must not
    exec; and the profiler's counters need the chip to themselves).  Returns the traffic entry (as profiles/round3/hbm_traffic.json
    holds them) or (None, reason)."""
    import shutil
    import tempfile
    if os.environ.get("MER_BENCH_PMC_CHILD") or os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD") or os.environ.get("ROCP_TOOL_LIBRARIES") or os.environ.get("ROCPROF_OUTPUT_PATH"):
        return None, "this process is itself being profiled"
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    t0 = time.time()
    got = {}
    child = None
    opts = ",".join(filter(None, [options, "pipes=1"]))
    for name, ctrs in PMC_PASSES.items():
        left = budget_s - (time.time() - t0)
        if left < 20:
            return None, "counter passes exceeded their %d s budget" % budget_s
        out = tempfile.mkdtemp(prefix="mer_pmc_%s_" % name, dir="/tmp")
        cmd = [exe, "--pmc"] + ctrs.split() + ["--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
               "--workload", workload, "--res", str(res), "--size", str(size), "--spp", str(spp), "--layout", layout, "--steps", "1", "--warmup", "0",
               "--no-cpu-baseline", "--no-solo-step", "--no-target-512", "--no-live-pmc", "--options", opts]
        env = dict(os.environ); env["TMPDIR"] = "/tmp"; env["MER_BENCH_PMC_CHILD"] = "1"
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=left)
        except subprocess.TimeoutExpired:
            shutil.rmtree(out, ignore_errors=True)
            return None, "counter pass '%s' timed out" % name
        sums = pmc_sums(out, "march_kernel" if "inline_walks=0" in opts else dominant_family(workload))
        shutil.rmtree(out, ignore_errors=True)
        if r.returncode != 0 or not sums:
            return None, "counter pass '%s' failed (rc %d): %s" % (name, r.returncode, r.stderr.decode(errors="replace")[-200:])
        got[name] = sums
        if child is None:
            try:
                child = json.loads(r.stdout.decode().strip().splitlines()[-1])
            except Exception:
                return None, "counter pass '%s' printed no bench line" % name
    c = child["counters_per_step"]
    steps = c["eikonal_steps"] if c["eikonal_steps"] > 0 else c["tentative_collisions"]
    wave_steps = max(c["lane_slots"] / 64.0, 1.0)
    rd, wr, sq = got["rd"], got["write"], got["sq"]
    read_b = 128 * rd.get("TCC_EA0_RDREQ_128B_sum", 0.0) + 64 * rd.get("TCC_EA0_RDREQ_64B_sum", 0.0) + 32 * rd.get("TCC_EA0_RDREQ_32B_sum", 0.0)
    write_b = 1024 * wr.get("WRITE_SIZE", 0.0)
    return {"method": "rocprofv3 --pmc passes taken by this bench run (one single-pipeline step each, child processes): read = 128*TCC_EA0_RDREQ_128B + "
                      "64*RDREQ_64B + 32*RDREQ_32B, write = WRITE_SIZE (KiB); fabric bytes, Infinity-Cache hits included",
            "live": True, "seconds": time.time() - t0, "layout": layout, "options": opts, "source_hash": source_hash(),
            "unit_steps": steps, "wave_steps": wave_steps, "read_bytes": read_b, "write_bytes": write_b,
            "hbm_bytes_per_eikonal_step": (read_b + write_b) / max(steps, 1.0),
            "valu_inst_per_wave_step": sq.get("SQ_INSTS_VALU", 0.0) / max(wave_steps, 1.0),
            "vmem_rd_inst_per_wave_step": sq.get("SQ_INSTS_VMEM_RD", 0.0) / max(wave_steps, 1.0),
            "wave_cycles_waiting_for_issue": sq.get("SQ_WAIT_INST_ANY", 0.0) / max(sq.get("SQ_WAVE_CYCLES", 0.0), 1.0)}, None


def workload_tag(name, res, size):
    return "%s_%d_film%d" % (name, res, size)


def scene_params(name, res, size, with_fields):
    """SceneParams of a BASELINE config; with_fields: numpy grids (the CPU baseline needs them; the GPU side of large grids uses fields
    generated in HBM by mer_synth_field_dev, see upload())."""
    import numpy as np
    from mitsubaer_amd import params as P, synth
    common = dict(width=size, height=size, rfilter=P.FILTER_BOX, rfilter_param=0.5, tr_estimator=P.TR_RATIO,
                  phase=P.PHASE_HG, g=0.8, density_scale=4.0, albedo=[0.9, 0.9, 0.9])
    dens = synth.density_field(res) if with_fields else None
    step = 0.5 * 2.0 / (res - 1)
    if name == "cfg2":
        p = P.SceneParams(density=dens, rif_mode=P.RIF_CONST, **common)
        desc = "configs[1]: %d^3 sigma_t grid, constant RIF (straight rays), HG g=0.8" % res
    elif name == "cfg3":
        p = P.SceneParams(density=dens, rif_mode=P.RIF_TRILINEAR, rif=synth.linear_rif(res) if with_fields else None, stepper=P.STEP_RK4, stepsize=step, **common)
        desc = "configs[2]: %d^3 sigma_t grid + %d^3 linear-gradient RIF, RK4 eikonal curved rays, HG g=0.8" % (res, res)
    elif name == "cfg4":
        p = P.SceneParams(density=dens, rif_mode=P.RIF_TRILINEAR, rif=synth.radial_rif(res) if with_fields else None, stepper=P.STEP_RK4, stepsize=step, **common)
        desc = "configs[3]: %d^3 sigma_t grid + %d^3 radial RIF, RK4 eikonal curved rays, HG g=0.8" % (res, res)
    elif name == "cfg5":
        # emissive heterogeneous medium + RGB albedo grid + curved-ray luminaire sampling of a point emitter (A12)
        g = np.linspace(0.0, 1.0, res, dtype=np.float32)
        alb = np.empty((res, res, res, 3), np.float32)
        alb[..., 0] = 0.55 + 0.4 * g[None, None, :]; alb[..., 1] = 0.55 + 0.4 * g[None, :, None]; alb[..., 2] = 0.55 + 0.4 * g[:, None, None]
        common.pop("albedo")
        p = P.SceneParams(density=synth.density_field(res), rif_mode=P.RIF_TRILINEAR, rif=synth.linear_rif(res), stepper=P.STEP_RK4, stepsize=step,
                          albedo_mode=P.ALBEDO_GRID, albedo_grid=alb, env_radiance=[0, 0, 0], emission=[0.2, 0.12, 0.06],
                          point_position=[0.2, 0.3, -0.1], point_intensity=[1.0, 0.8, 0.5], **common)
        desc = "configs[4]: %d^3 sigma_t + RGB albedo grids, emissive medium, %d^3 linear RIF, RK4 curved rays + curved-ray point-emitter NEE" % (res, res)
    else:
        raise SystemExit("unknown workload %s" % name)
    return p, desc


def build_workload(name, res, size, spp):
    """(SceneParams with numpy fields, description) -- used by the full-size tests"""
    p, desc = scene_params(name, res, size, with_fields=True)
    return p, desc + ", %d^2 x %d spp" % (size, spp)


def upload(ctx, name, res, p, layout):
    """-> (scene desc, volumes).  Grids of 512^3 and more are generated on the device (mer_synth_field_dev: the same formulas as
    mitsubaer_amd/synth.py) instead of 0.5 - 4 GiB numpy arrays per field."""
    from mitsubaer_amd import capi
    if p.density is not None or name in ("cfg5",):
        return ctx.upload_scene(p, layout=layout)
    dl = capi.LAYOUT_CELL8 if layout in (capi.LAYOUT_BRICK27, capi.LAYOUT_BRICK125, capi.LAYOUT_AUTO) else layout
    dens = ctx.synth_volume(0, res, layout=dl)
    rif = None
    if name in ("cfg3", "cfg4"):
        rif = ctx.synth_volume(1 if name == "cfg3" else 2, res, layout=layout)
    return ctx.scene_desc(p, dens, None, rif), [v for v in (dens, rif) if v is not None]


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


def committed_traffic(tag, layout, options):
    """Fallback when the counters could not be taken in this run: the committed pass of this workload (profiles/round3/hbm_traffic.json,
    scratch/pmc_traffic.py) -- used only if it was taken on THESE kernel sources, THIS RIF layout and THESE options; (entry, None) or (None, why)."""
    try:
        e = json.load(open(TRAFFIC_FILE)).get(tag)
    except Exception:
        e = None
    if not e:
        return None, "no counter pass committed for %s" % tag
    want = ",".join(filter(None, [options, "pipes=1"]))
    if e.get("source_hash") != source_hash():
        return None, "the committed counter pass of %s was taken on other kernel sources (%s, now %s)" % (tag, e.get("source_hash"), source_hash())
    if e.get("layout") != layout or e.get("options") != want:
        return None, "the committed counter pass of %s was taken with layout %s, options %s" % (tag, e.get("layout"), e.get("options"))
    return e, None


def roofline_block(ctx, capi, p, tag, solo, timed, layout_name, traffic=None, traffic_why=None):
    """roofline of the dominant kernel (K_march).  `solo` = (passes, march_ms, event_ms, counters, wall_s) of one step rendered as ONE
    pipeline -- its launches have the chip to themselves, so bytes / duration is a chip-level rate; `timed` = the same for the timed
    region's average step (concurrent pipelines: only the aggregate over the step means anything there)."""
    n1, m1, e1, c1, wall1 = solo
    steps = float(c1[capi.C_STEPS]); tent = float(c1[capi.C_TENTATIVE]); paths = float(c1[capi.C_PATHS])
    b_alg = algorithmic_bytes(c1, p) - 40.0 * paths                       # K_march performs every field fetch; the film write is K_event's
    launches = max(n1, 1)
    inline = p.rif_mode == 0 and ctx.get_option("inline_walks") == 1 and p.sigma_mode == 1 and p.method == 0
    if inline:
        # straight rays: K_event runs the walks itself (persistent lanes) -- it IS the kernel that fetches the field; K_march's launches are empty
        b_alg = algorithmic_bytes(c1, p)
    dom_ms = e1 if inline else m1          # device time of the dominant kernel in the single-pipeline step
    launch_ms = dom_ms / launches
    tr = traffic
    curved = p.rif_mode != 0
    unit_steps = steps if curved else tent           # the unit the counter bytes are quoted per: eikonal steps (curved rays) or tentative collisions (straight)
    out = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "kernel": ("mer::march_kernel<curved, %s trilinear, rk4, grid>" % layout_name) if curved else
                     ("mer::event_kernel<straight, grid sigma_t, inline walks> (one persistent launch does most of a step: per-launch averages mean little here)" if inline
                      else "mer::march_kernel<straight, grid sigma_t (cell8 records)>"),
           "kernel_avg_launch_ms": launch_ms, "launches_per_step": n1, "kernel_ms_per_step": dom_ms, "march_kernel_ms_per_step": m1, "event_kernel_ms_per_step": e1,
           "step_wall_ms": wall1 * 1e3, "pipelines": 1,
           "what": "one untimed step rendered as ONE pipeline (option pipes = 1): K_march's launches have the chip to themselves, HIP events on the launch stream",
           "algorithmic_bytes_per_launch": b_alg / launches,
           "algorithmic_GBps": b_alg / max(dom_ms, 1e-9) / 1e6,
           "counters_per_step": {"paths": paths, "eikonal_steps": steps, "tentative_collisions": tent, "real_collisions": float(c1[capi.C_REAL])},
           "active_lane_fraction": float(c1[capi.C_ACTIVE_LANES] / max(c1[capi.C_LOOP_ITERS], 1.0))}
    if tr:
        # fabric-side bytes per unit step (rocprofv3 --pmc: TCC_EA0_RDREQ by request size + WRITE_SIZE) x this step's unit steps
        hbm = tr["hbm_bytes_per_eikonal_step"] * unit_steps
        out.update({"achieved": hbm / max(dom_ms, 1e-9) / 1e6, "traffic": hbm / launches,
                    "reuse_factor": b_alg / max(hbm, 1.0), "bytes_per_unit_step": tr["hbm_bytes_per_eikonal_step"],
                    "traffic_source": ("counter passes taken inside this run (%.0f s): " % tr.get("seconds", 0) if tr.get("live") else
                                       "extrapolated from the committed counter pass %s [%s] (same kernel sources %s, layout, options): " % (os.path.relpath(TRAFFIC_FILE, ROOT), tag, tr.get("source_hash")))
                                      + tr.get("method", "")})
        out["frac"] = out["achieved"] / HBM_PEAK_GBS
    else:
        # no valid counter pass: the algorithmic rate is an UPPER bound of the HBM rate only when nothing is reused, which is not the case
        # here (register cell cache, L2, Infinity Cache) -- report no fraction rather than a wrong one
        out.update({"achieved": None, "frac": None, "traffic": None, "traffic_source": traffic_why or "no counter pass"})
    out["note"] = ("achieved / traffic are FABRIC bytes (what L2 requests beyond itself, Infinity-Cache hits included), an upper bound of the DRAM bytes: at 256^3 the "
                   "working set (~0.3 GiB) is about the size of the 256 MiB Infinity Cache, so the DRAM share is unknown there; the 512^3 block (2.5 GiB) can be read as HBM")
    if tr and tr.get("valu_inst_per_wave_step") and curved:
        wave_steps = float(c1[capi.C_LOOP_ITERS]) / 64.0
        ginst = tr["valu_inst_per_wave_step"] * wave_steps / max(dom_ms, 1e-9) / 1e6
        out["valu"] = {"inst_per_wave_step": tr["valu_inst_per_wave_step"], "achieved_Ginst_s": ginst, "peak_Ginst_s": VALU_PEAK_GINST,
                       "frac": ginst / VALU_PEAK_GINST,
                       "note": "peak = independent v_fma_f32 issue (2.3 cycles per wave64 instruction, measured); a dependent one issues every ~4.3 cycles"}
    out["limiter"] = (tr or {}).get("limiter", "see DESIGN.md section 4")
    if float(c1[capi.C_CONNECT_UNITS]) > 0:
        out["dominant_kernel_note"] = ("this workload's device time is dominated by mer::connect_stage_kernel (curved-ray connection solver, ~90 %: "
                                       "profiles/round3/stats/rocprofv3_kernel_stats_cfg5_256_spp128.csv); the block above is K_march's share only")
    if timed is not None:
        k_ms, m_ms, e_ms, n_pass, counters, pipes = timed
        b_step = algorithmic_bytes(counters, p)
        ws = {"pipelines": pipes, "ms": k_ms, "kernel_ms_summed_over_pipelines": m_ms, "event_kernel_ms_summed_over_pipelines": e_ms,
              "launches": n_pass, "algorithmic_bytes": b_step, "algorithmic_GBps": b_step / max(k_ms, 1e-9) / 1e6,
              "note": "timed region: concurrent pipelines stretch one another's launches, so only the aggregate over the step is a chip-level figure"}
        if tr:
            hb = tr["hbm_bytes_per_eikonal_step"] * float(counters[capi.C_STEPS] if curved else counters[capi.C_TENTATIVE])
            ws.update({"hbm_bytes": hb, "achieved": hb / max(k_ms, 1e-9) / 1e6, "frac": hb / max(k_ms, 1e-9) / 1e6 / HBM_PEAK_GBS})
        out["whole_step"] = ws
    return out


def spawn_ranks(args):
    """plain `python bench.py --gpus N`: start N ranks with torch.distributed.run before this process touches a GPU, relay rank 0."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


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
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--shard", default=None, choices=["samples", "tiles"], help="default: samples for weak scaling, tiles for strong")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--device", type=int, default=None, help="force this GPU index for every rank (rehearsal on one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-target-512", action="store_true", help="skip the 512^3 block of the default run")
    ap.add_argument("--no-solo-step", action="store_true", help="skip the extra single-pipeline step (profiled runs)")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not take the roofline's counter passes (child rocprofv3 runs) inside this run; fall back to the committed pass")
    ap.add_argument("--options", default="", help="mer_context_set_option pairs, e.g. pipes=1,ksteps=96")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.layout == "auto":
        args.layout = "brick27" if args.res ** 3 <= 1 << 28 else "cell8"
    # the roofline's byte counters, taken by child processes BEFORE this one touches the GPU (N = 1 only)
    single = args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1
    traffic = traffic_why = traffic5 = traffic5_why = None
    want_512 = args.workload == "cfg3" and args.res == 256 and single and not args.no_target_512 and not args.no_cpu_baseline
    if single and not args.no_solo_step:
        if not args.no_live_pmc:
            import torch  # noqa: F401  -- pages the image in (1-2 min on a fresh box) so that the children's imports are fast; importing touches no GPU
            traffic, traffic_why = live_pmc(args.workload, args.res, args.size, args.spp, args.layout, args.options)
            if want_512 and traffic:
                traffic5, traffic5_why = live_pmc("cfg3", 512, args.size, args.spp, "brick27", args.options, budget_s=120.0)
        if not traffic:
            why_live = traffic_why or "live counter passes disabled (--no-live-pmc)"
            traffic, traffic_why = committed_traffic(workload_tag(args.workload, args.res, args.size), args.layout, args.options)
            traffic_why = "%s; %s" % (why_live, traffic_why) if traffic_why else None
        if want_512 and not traffic5:
            why_live = traffic5_why or "live counter passes not taken"
            traffic5, traffic5_why = committed_traffic(workload_tag("cfg3", 512, args.size), "brick27", args.options)
            traffic5_why = "%s; %s" % (why_live, traffic5_why) if traffic5_why else None

    import numpy as np
    import torch
    from mitsubaer_amd import capi, dist as mdist
    rank, world, local = mdist.init_process_group(args.backend)
    if world != args.gpus:
        print("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if args.device is not None:
        local = args.device
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    shard_mode = args.shard or ("tiles" if args.scaling == "strong" else "samples")

    small = args.res < 512 or args.workload == "cfg5"
    p, desc = scene_params(args.workload, args.res, args.size, with_fields=small)
    desc += ", %d^2 x %d spp" % (args.size, args.spp)
    ctx = capi.Context(local)
    for kv in filter(None, args.options.split(",")):
        k, v = kv.split("="); ctx.set_option(k, int(v))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    layout = {"cell8": capi.LAYOUT_CELL8, "brick27": capi.LAYOUT_BRICK27, "brick125": capi.LAYOUT_BRICK125, "dense": capi.LAYOUT_DENSE}[args.layout]
    sc, vols = upload(ctx, args.workload, args.res, p, layout)
    film = torch.zeros((p.height, p.width, 5), dtype=torch.float32, device=dev)
    spp_job = args.spp * world if args.scaling == "weak" else args.spp
    sh = mdist.shard_args(shard_mode, rank, world, spp_job)

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
        # HIP events recorded by libmer around the render on the launch stream; reading them waits for this step's kernels only
        kernel_ms.append(ctx.last_kernel_ms())
        n_, m_, e_ = ctx.last_render_stats()
        passes.append(n_); march_ms.append(m_); event_ms.append(e_)
    barrier()
    elapsed = time.perf_counter() - t0
    tl = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    per_rank_ms = [elapsed / max(args.steps, 1) * 1e3]
    # ranks that joined the RCCL communicator (backend "nccl" is RCCL on ROCm); null under any other backend -- a gloo rehearsal says so
    backend_used = torch.distributed.get_backend() if world > 1 else None
    rccl_ranks = None
    if world > 1:
        gathered = [torch.zeros_like(tl) for _ in range(world)]
        torch.distributed.all_gather(gathered, tl)
        per_rank_ms = [float(g.item()) / max(args.steps, 1) * 1e3 for g in gathered]
        elapsed = max(float(g.item()) for g in gathered)                  # MAX over ranks
        rccl_ranks = torch.distributed.get_world_size() if backend_used == "nccl" else None
    counters = ctx.counters().astype(np.float64)
    # one more step OUTSIDE the timed region as a single pipeline: per-launch figures of the dominant kernel
    solo = None
    if not args.no_solo_step:             # every rank (N > 1: rank 0's block is printed; the others keep the GPUs in step until the next collective)
        with ctx.options(pipes=1):
            ctx.counters_reset()
            mer_sh = mdist.shard_args("samples", 0, 1, args.spp)
            film2 = torch.zeros_like(film)
            t1 = time.perf_counter()
            ctx.render(sc, film2.data_ptr(), mer_sh["spp_begin"], mer_sh["spp_count"], seed=args.steps, spp_stride=1)
            torch.cuda.synchronize(); solo_wall = time.perf_counter() - t1
            n1, m1, e1 = ctx.last_render_stats()
            solo = (n1, m1, e1, ctx.counters().astype(np.float64), solo_wall)
            del film2
    ct = torch.tensor(counters, dtype=torch.float64, device=dev)
    mdist.reduce_counters(ct)
    total_paths = float(ct[capi.C_PATHS].item())

    if rank == 0:
        name, cus, hbm = ctx.device_info()
        pipes = ctx.get_option("pipes")
        slots = ctx.get_option("nslots") or cus * 2048 * 4              # the library's default: 4 x the resident lanes of the chip
        tile_share = sh["tile_count"]
        tag = workload_tag(args.workload, args.res, args.size)
        out = {
            "metric": "Mpaths/sec (curved-ray heterogeneous volume); roofline = measured HBM GB/s of the dominant kernel vs peak",
            "value": total_paths / elapsed / 1e6, "unit": "Mpaths/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic", "backend": backend_used,
            "config": {"workload": desc, "grid": args.res, "film": [p.width, p.height],
                       "spp_job": spp_job, "spp_per_gpu": args.spp if args.scaling == "weak" else None,
                       "pipelines_per_gpu": pipes, "stepper": "rk4", "rif_interp": "trilinear", "layout": args.layout, "shard": shard_mode,
                       "stepsize": p.stepsize, "estimator": "volpath + delta tracking on eikonal rays, ratio-tracking NEE",
                       "device": name, "cus": cus, "backend": backend_used,
                       # paths of one rank's step per path-state slot: below ~4 the render is one generation of paths and its time is the tail of the
                       # longest paths, not the steady rate (configs[3] at 8 spp: 4; at 128 spp: 64)
                       "paths_per_slot": round(p.width * p.height * sh["spp_count"] / max(tile_share, 1) / max(slots, 1), 2)},
            "rccl_ranks": rccl_ranks, "per_rank_ms_per_step": per_rank_ms,
            # device counters of rank 0, per step of the timed region (inputs of SURVEY section 8d's byte formula)
            "counters_per_step": {"paths": counters[capi.C_PATHS] / max(args.steps, 1), "eikonal_steps": counters[capi.C_STEPS] / max(args.steps, 1),
                                  "tentative_collisions": counters[capi.C_TENTATIVE] / max(args.steps, 1), "real_collisions": counters[capi.C_REAL] / max(args.steps, 1),
                                  "lane_slots": counters[capi.C_LOOP_ITERS] / max(args.steps, 1), "active_lane_steps": counters[capi.C_ACTIVE_LANES] / max(args.steps, 1),
                                  "connections": counters[capi.C_NEE] / max(args.steps, 1) if any(p.point_intensity) else 0.0,
                                  "connect_units": counters[capi.C_CONNECT_UNITS] / max(args.steps, 1), "connect_steps": counters[capi.C_CONNECT_STEPS] / max(args.steps, 1),
                                  "connect_lane_slots": counters[capi.C_CONNECT_LANE_SLOTS] / max(args.steps, 1),
                                  "side_walks_spawned": counters[capi.C_SIDE_SPAWNED] / max(args.steps, 1), "side_walks_in_the_paths_lane": counters[capi.C_SIDE_INLINE] / max(args.steps, 1)},
        }
        if solo is not None:
            timed = (float(np.mean(kernel_ms)), float(np.mean(march_ms)), float(np.mean(event_ms)), float(np.mean(passes)),
                     counters / max(args.steps, 1), pipes)
            out["roofline"] = roofline_block(ctx, capi, p, tag, solo, timed, args.layout, traffic, traffic_why)
        if not args.no_cpu_baseline and world == 1:
            if p.density is None:
                pc, _ = scene_params(args.workload, args.res, args.size, with_fields=True)
            else:
                pc = p
            out["cpu_baseline"] = cpu_baseline(pc, args.cpu_seconds)
        # the north star's target volume (512^3) in the same run: two steps + its CPU baseline on a bounded sample
        if want_512:
            for v in vols:
                v.destroy()
            vols = []
            p5, d5 = scene_params("cfg3", 512, args.size, with_fields=False)
            sc5, v5 = upload(ctx, "cfg3", 512, p5, capi.LAYOUT_BRICK27)
            film.zero_(); ctx.render(sc5, film.data_ptr(), 0, args.spp, seed=50); torch.cuda.synchronize()       # warm-up
            ts = time.perf_counter()
            for i in range(2):
                film.zero_(); ctx.render(sc5, film.data_ptr(), 0, args.spp, seed=60 + i)
            torch.cuda.synchronize(); dt5 = (time.perf_counter() - ts) / 2
            with ctx.options(pipes=1):
                ctx.counters_reset(); film.zero_(); t1 = time.perf_counter()
                ctx.render(sc5, film.data_ptr(), 0, args.spp, seed=70); torch.cuda.synchronize(); w5 = time.perf_counter() - t1
                n5, m5, e5 = ctx.last_render_stats(); c5 = ctx.counters().astype(np.float64)
            t512 = {"workload": d5 + ", %d^2 x %d spp" % (args.size, args.spp), "value": p5.width * p5.height * args.spp / dt5 / 1e6, "unit": "Mpaths/s",
                    "ms_per_step": dt5 * 1e3, "steps": 2,
                    "roofline": roofline_block(ctx, capi, p5, workload_tag("cfg3", 512, args.size), (n5, m5, e5, c5, w5), None, "brick27", traffic5, traffic5_why)}
            pc5, _ = scene_params("cfg3", 512, args.size, with_fields=True)
            t512["cpu_baseline"] = cpu_baseline(pc5, 12.0)
            t512["gpu_over_cpu"] = t512["value"] / t512["cpu_baseline"]["value"]
            out["target_512"] = t512
            for v in v5:
                v.destroy()
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
