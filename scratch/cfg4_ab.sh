#!/bin/bash
# 1024^3 fields on one GPU (configs[3] at 8 spp): RIF layout x march-list sorting
mkdir -p gpurun_out/cfg4ab
run() { tag=$1; shift; env "$@" timeout -k 10 400 python bench.py --workload cfg4 --res 1024 --size 1024 --spp 8 --steps 2 --warmup 1 --no-cpu-baseline $LAY > gpurun_out/cfg4ab/$tag.json 2> gpurun_out/cfg4ab/$tag.err || exit 1; }
LAY="--layout brick27" run brick_nosort MER_MQ_SORT=0 || exit 1
LAY="--layout cell8" run cell8_sort MER_MQ_SORT=1 || exit 1
LAY="--layout cell8" run cell8_nosort MER_MQ_SORT=0 || exit 1
LAY="--layout cell8" run cell8_nosort_k64 MER_MQ_SORT=0 MER_KSTEPS=64 || exit 1
