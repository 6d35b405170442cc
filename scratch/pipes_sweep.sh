#!/bin/bash
mkdir -p gpurun_out/pipes
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pipes/tests.log 2>&1 || { tail -40 gpurun_out/pipes/tests.log | cut -c1-300; exit 1; }
tail -2 gpurun_out/pipes/tests.log
for n in 1 2 3 4; do
  MER_PIPES=$n timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pipes/cfg3_p$n.json 2> gpurun_out/pipes/cfg3_p$n.err || exit 1
done
for n in 1 2; do
  MER_PIPES=$n timeout -k 10 200 python bench.py --res 512 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pipes/cfg3_512_p$n.json 2> gpurun_out/pipes/cfg3_512_p$n.err || exit 1
  MER_PIPES=$n timeout -k 10 120 python bench.py --workload cfg2 --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pipes/cfg2_p$n.json 2> gpurun_out/pipes/cfg2_p$n.err || exit 1
done
