#!/bin/bash
# K_event at 3 / 4 waves per SIMD (168 / 128 VGPR) under 4 concurrent pipelines
mkdir -p gpurun_out/ew
for w in 3 4; do
  MER_LIB=$PWD/mitsubaer_amd/libmer_ew$w.so timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ew/cfg3_ew$w.json 2> gpurun_out/ew/cfg3_ew$w.err || exit 1
done
