#!/bin/bash
# usage: scratch/pmc_pass.sh <tag> <bench args quoted> <counters...>   (run from the repo root on the GPU box)
tag=$1; shift; bargs=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py $bargs --steps 1 --warmup 0 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
echo "pass $tag rc=$?"
cd $GRAFT_REPO_ROOT
python3 scratch/pmc_sum.py $out > gpurun_out/pmc_$tag.txt 2>&1
find $out -name "*.csv" -size +1M -delete
exit 0
