#!/bin/bash
# rocprofv3 kernel stats of single-pipeline steps with and without the spatial sort of the march list
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/msort_stats; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift
  rm -rf /tmp/prof_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $R/bench.py --no-live-pmc --no-cpu-baseline --no-target-512 --no-solo-step "$@" > $O/${tag}_bench.json 2> $O/${tag}_stderr.txt
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${tag}_kernel_stats.csv
  echo "== $tag"; head -8 $O/${tag}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
}
run s0_256 --options pipes=1,march_sort=0
run s2_256 --options pipes=1,march_sort=2
run s3_256 --options pipes=1,march_sort=3
run s0_512 --res 512 --steps 2 --options pipes=1,march_sort=0
run s3_512 --res 512 --steps 2 --options pipes=1,march_sort=3
