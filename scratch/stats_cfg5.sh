#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_stats; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c5
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python3 $R/bench.py --workload cfg5 --spp 128 --steps 2 --warmup 1 --no-target-512 --no-cpu-baseline > $O/cfg5_256_spp128_bench.json 2> $O/cfg5_stderr.txt
cp $(find /tmp/prof_c5 -name "*kernel_stats.csv" | head -1) $O/cfg5_256_spp128_kernel_stats.csv
head -5 $O/cfg5_256_spp128_kernel_stats.csv | cut -c1-170; tail -1 $O/cfg5_256_spp128_bench.json | cut -c1-200
