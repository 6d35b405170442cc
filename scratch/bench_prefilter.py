#!/usr/bin/env python3
"""K_prefilter (mer_volume_build_spline) timing: N^3 float32 -> cubic-B-spline coefficients, inputs resident in HBM.
Algorithmic bytes: 3 passes x (4 B read + 4 B write) per voxel."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitsubaer_amd import capi
ctx = capi.Context(0)
for N in [int(a) for a in sys.argv[1:]] or [256, 512]:
    v = ctx.synth_volume(2, N)
    ctx.synchronize()
    t0 = time.time(); v.build_spline(); ctx.synchronize(); dt = time.time() - t0
    alg = 24.0 * N ** 3
    print(json.dumps({"kernel": "K_prefilter", "N": N, "ms": dt * 1e3, "algorithmic_GB": alg / 1e9, "achieved_GBs": alg / dt / 1e9, "frac_of_8TBs": alg / dt / 8e12}))
    v.destroy()
