#!/usr/bin/env python3
"""Predicts the 8-GPU strong-scaling behaviour of a job from ONE GPU: renders each of the `world` shards of the job one after another
and prints per-shard milliseconds, max / mean, and the scaling the slowest shard implies (sum of shards / (world x slowest shard): what
N GPUs would reach if nothing but the imbalance and the per-shard tail cost them anything).

  python scratch/shard_balance.py --workload cfg3 --res 256 --size 512 --spp 256 --world 8
  python scratch/shard_balance.py --workload cfg4 --res 1024 --size 1024 --spp 128 --world 8      # spp of the JOB; tile shards render all of them

Modes: tiles (diagonal deal, the default), tiles_plain (option tile_deal = 0: row-major round robin = whole tile columns), samples.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--modes", default="tiles,tiles_plain,samples")
    ap.add_argument("--options", default="")
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from mitsubaer_amd import capi, dist as mdist
    ctx = capi.Context(0)
    for kv in filter(None, args.options.split(",")):
        k, v = kv.split("="); ctx.set_option(k, int(v))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    small = args.res < 512
    p, desc = bench.scene_params(args.workload, args.res, args.size, with_fields=small)
    layout = capi.LAYOUT_BRICK27 if args.res ** 3 <= 1 << 28 else capi.LAYOUT_CELL8
    sc, vols = bench.upload(ctx, args.workload, args.res, p, layout)
    film = torch.zeros((p.height, p.width, 5), dtype=torch.float32, device="cuda")

    def timed(sh):
        film.zero_(); torch.cuda.synchronize()
        ctx.counters_reset()
        t0 = time.perf_counter()
        ctx.render(sc, film.data_ptr(), sh["spp_begin"], sh["spp_count"], seed=1, spp_stride=sh["spp_stride"], tile_rank=sh["tile_rank"], tile_count=sh["tile_count"])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        c = ctx.counters()
        return dt * 1e3, float(c[capi.C_PATHS]), float(c[capi.C_STEPS])

    whole = mdist.shard_args("samples", 0, 1, args.spp)
    timed(mdist.shard_args("samples", 0, 1, max(1, args.spp // 8)))        # warm-up (allocations, code objects)
    t_whole, paths_whole, steps_whole = timed(whole)
    out = {"workload": desc + ", %d^2 x %d spp" % (args.size, args.spp), "world": args.world, "whole_job_ms": t_whole,
           "whole_job_Mpaths_s": paths_whole / t_whole / 1e3, "modes": {}}
    print("%s\nwhole job on one GPU: %.1f ms, %.1f Mpaths/s" % (out["workload"], t_whole, out["whole_job_Mpaths_s"]))
    for mode in args.modes.split(","):
        with ctx.options(tile_deal=0 if mode == "tiles_plain" else 1):
            ms, st = [], []
            for r in range(args.world):
                sh = mdist.shard_args("samples" if mode == "samples" else "tiles", r, args.world, args.spp)
                t, _, s = timed(sh)
                ms.append(t); st.append(s)
        ms = np.array(ms); st = np.array(st)
        res = {"shard_ms": [round(float(x), 2) for x in ms], "max_over_mean_ms": float(ms.max() / ms.mean()),
               "max_over_mean_eikonal_steps": float(st.max() / max(st.mean(), 1.0)),
               "implied_scaling_sum_over_world_max": float(ms.sum() / (args.world * ms.max())),
               "speedup_vs_whole_job": float(t_whole / ms.max())}
        out["modes"][mode] = res
        print("%-12s shard ms %s\n             max/mean %.3f (work: %.3f)  efficiency sum/(N max) %.3f  speed-up vs whole job %.2fx of %d" % (
            mode, " ".join("%.1f" % x for x in ms), res["max_over_mean_ms"], res["max_over_mean_eikonal_steps"],
            res["implied_scaling_sum_over_world_max"], res["speedup_vs_whole_job"], args.world))
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
