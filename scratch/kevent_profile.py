#!/usr/bin/env python3
"""Section profile of K_event (library built with -DMER_PROFILE: csrc/mer_wavefront.hpp PROF marks, mer_debug_prof in mer_render_brick.hip).
   MER_LIB=.../libmer_prof.so python scratch/kevent_profile.py [pipes]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mitsubaer_amd import capi
pipes = int(sys.argv[1]) if len(sys.argv) > 1 else 1
p, desc = bench.build_workload("cfg3", 256, 512, 256)
ctx = capi.Context(0, pipes=pipes)
sc, vols = bench.upload(ctx, "cfg3", 256, p, capi.LAYOUT_BRICK27)
film = torch.zeros((p.height, p.width, 5), dtype=torch.float32, device="cuda:0")
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.render(sc, film.data_ptr(), 0, 256, seed=1); torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
assert ctx.lib.mer_debug_prof(out, 1) == 0
t0 = time.perf_counter(); ctx.render(sc, film.data_ptr(), 0, 256, seed=2); torch.cuda.synchronize(); dt = time.perf_counter() - t0
assert ctx.lib.mer_debug_prof(out, 0) == 0
names = ["prologue: list lookup + record load", "regeneration (ring pop, camera ray, begin)", "real collision + luminaire-sample spawn", "TR_DONE (in-lane walks' ends)",
         "phase sample + look-up spawn", "after look-up: roulette + next free flight (begin)", "other events / loop overhead", "park: record store", "pushes + counters", "flush"]
tot = float(sum(out[k] for k in range(10)))
n, m, e = ctx.last_render_stats()
print("%s, pipes=%d: render %.1f ms, K_event %.1f ms (HIP events), %d waves with work, %.0f cycles (s_memtime ticks) per wave" % (desc, pipes, dt * 1e3, e, out[15], tot / max(out[15], 1)))
for k in range(10):
    print("  %-52s %5.1f %%   %8.0f ticks per wave" % (names[k], 100.0 * out[k] / tot, out[k] / max(out[15], 1)))
