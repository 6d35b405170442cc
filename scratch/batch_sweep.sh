#!/bin/bash
# tentative-collision batching in K_march: MER_ARRIVE_BATCH = 1 (off), 2, 4 (libmer.so), 8
mkdir -p gpurun_out/batch
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/batch/tests.log 2>&1 || { tail -30 gpurun_out/batch/tests.log | cut -c1-300; exit 1; }
tail -2 gpurun_out/batch/tests.log
for b in 1 2 4 8; do
  lib=$PWD/mitsubaer_amd/libmer_b$b.so; [ $b -eq 4 ] && lib=$PWD/mitsubaer_amd/libmer.so
  MER_LIB=$lib timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/batch/cfg3_b$b.json 2> gpurun_out/batch/cfg3_b$b.err || exit 1
  MER_LIB=$lib timeout -k 10 200 python bench.py --res 512 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/batch/cfg3_512_b$b.json 2> gpurun_out/batch/cfg3_512_b$b.err || exit 1
done
