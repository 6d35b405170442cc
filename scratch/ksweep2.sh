#!/bin/bash
mkdir -p gpurun_out/ks2
for K in 48 64 96 128 192; do
  MER_KSTEPS=$K timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ks2/cfg3_K$K.json 2> gpurun_out/ks2/cfg3_K$K.err || exit 1
done
for K in 64 96; do
  MER_KSTEPS=$K timeout -k 10 200 python bench.py --res 512 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ks2/cfg3_512_K$K.json 2> gpurun_out/ks2/cfg3_512_K$K.err || exit 1
done
