#!/bin/bash
mkdir -p gpurun_out/pipes2
for cfg in "2 4194304" "4 4194304" "4 8388608" "3 6291456" "2 8388608"; do
  set -- $cfg
  MER_PIPES=$1 MER_NSLOTS=$2 timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pipes2/cfg3_p$1_s$2.json 2> gpurun_out/pipes2/cfg3_p$1_s$2.err || exit 1
done
MER_PIPES=4 MER_NSLOTS=4194304 timeout -k 10 200 python bench.py --res 512 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pipes2/cfg3_512_p4_s4M.json 2> gpurun_out/pipes2/cfg3_512_p4.err || exit 1
MER_PIPES=4 timeout -k 10 200 python bench.py --res 512 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pipes2/cfg3_512_p4_s2M.json 2> gpurun_out/pipes2/cfg3_512_p4b.err || exit 1
