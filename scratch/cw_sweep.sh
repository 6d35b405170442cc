#!/bin/bash
# K_connect at 3 / 4 waves per SIMD (configs[4])
mkdir -p gpurun_out/cw
for w in 3 4; do
  BENCH_NO_SOLO_STEP=1 MER_LIB=$PWD/mitsubaer_amd/libmer_cw$w.so timeout -k 10 200 python bench.py --workload cfg5 --spp 32 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/cw/cfg5_cw$w.json 2> gpurun_out/cw/cfg5_cw$w.err || exit 1
done
BENCH_NO_SOLO_STEP=1 timeout -k 10 200 python bench.py --workload cfg5 --spp 32 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/cw/cfg5_base.json 2> gpurun_out/cw/cfg5_base.err || exit 1
