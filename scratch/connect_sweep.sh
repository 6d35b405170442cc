#!/bin/bash
mkdir -p gpurun_out/cev
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/cev/tests.log 2>&1 || { tail -30 gpurun_out/cev/tests.log | cut -c1-300; exit 1; }
tail -2 gpurun_out/cev/tests.log
for m in 1 4 8 16; do
  MER_CONNECT_EVERY=$m timeout -k 10 200 python bench.py --workload cfg5 --spp 32 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/cev/cfg5_m$m.json 2> gpurun_out/cev/cfg5_m$m.err || exit 1
done
