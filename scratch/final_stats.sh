#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench commands (GPU box, repo root) -> gpurun_out/final_stats/<tag>_{kernel_stats.csv,bench.json}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_stats; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift
  rm -rf /tmp/prof_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $R/bench.py "$@" > $O/${tag}_bench.json 2> $O/${tag}_stderr.txt
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${tag}_kernel_stats.csv
  echo "$tag rc=$? $(tail -c 300 $O/${tag}_bench.json | head -c 10)"; tail -1 $O/${tag}_bench.json | cut -c1-260
}
run cfg3_256_default
run cfg3_512 --res 512 --no-target-512 --no-cpu-baseline
run cfg4_1024_spp64 --workload cfg4 --res 1024 --size 1024 --spp 64 --steps 2 --warmup 1 --no-target-512 --no-cpu-baseline
run cfg5_256_spp128 --workload cfg5 --spp 128 --steps 2 --warmup 1 --no-target-512 --no-cpu-baseline
run cfg2_256 --workload cfg2 --spp 64 --no-target-512 --no-cpu-baseline
