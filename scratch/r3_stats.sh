#!/bin/bash
# rocprofv3 --kernel-trace --stats of the round-3 bench commands (GPU box, repo root) -> gpurun_out/r3_stats/<tag>_{kernel_stats.csv,bench.json}
# pipes=1 runs: the per-kernel average of K_march is a chip-level figure only when its launches have the chip to themselves
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_stats; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift
  rm -rf /tmp/prof_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $R/bench.py --no-live-pmc --no-cpu-baseline --no-target-512 "$@" > $O/${tag}_bench.json 2> $O/${tag}_stderr.txt
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${tag}_kernel_stats.csv
  echo "$tag: $(python3 -c "import json,sys; d=json.loads(open('$O/${tag}_bench.json').read().strip().splitlines()[-1]); print('%.1f Mpaths/s %.1f ms' % (d['value'], d['ms_per_step']))" 2>&1 | tail -1)"
}
run cfg3_256_default
run cfg3_256_pipes1 --options pipes=1 --no-solo-step
run cfg3_512_pipes1 --res 512 --options pipes=1 --no-solo-step --steps 2
run cfg4_1024_spp128_pipes1 --workload cfg4 --res 1024 --size 1024 --spp 128 --steps 1 --warmup 1 --options pipes=1 --no-solo-step
run cfg5_256_spp128 --workload cfg5 --spp 128 --steps 2 --warmup 1
run cfg2_256 --workload cfg2 --spp 64
run cfg2_256_pipes1 --workload cfg2 --spp 64 --options pipes=1 --no-solo-step
run cfg2_256_two_kernels --workload cfg2 --spp 64 --options inline_walks=0
