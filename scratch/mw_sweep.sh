#!/bin/bash
mkdir -p gpurun_out/mw
for w in 3 5; do
  MER_LIB=$PWD/mitsubaer_amd/libmer_mw$w.so timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/mw/cfg3_mw$w.json 2> gpurun_out/mw/cfg3_mw$w.err || exit 1
  MER_PIPES=1 MER_LIB=$PWD/mitsubaer_amd/libmer_mw$w.so timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/mw/cfg3_mw${w}_p1.json 2> gpurun_out/mw/cfg3_mw${w}_p1.err || exit 1
done
