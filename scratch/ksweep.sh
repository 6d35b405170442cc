#!/bin/bash
# sweep K (eikonal steps per lane per pass) on the bench workload; results under gpurun_out/ksweep/
mkdir -p gpurun_out/ksweep
for K in 64 96 128 160 192 256; do
  MER_KSTEPS=$K timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ksweep/cfg3_K$K.json 2> gpurun_out/ksweep/cfg3_K$K.err || exit 1
done
for K in 96 160; do
  MER_KSTEPS=$K timeout -k 10 200 python bench.py --res 512 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ksweep/cfg3_512_K$K.json 2> gpurun_out/ksweep/cfg3_512_K$K.err || exit 1
done
