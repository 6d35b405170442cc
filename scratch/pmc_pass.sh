#!/bin/bash
# One rocprofv3 counter pass (run from the repo root ON THE GPU BOX): scratch/pmc_pass.sh <outdir under gpurun_out> <counters (space separated, quoted)> -- <program and args>
# Counters are collected in their own run with --kernel-trace only (the pool refuses --pmc together with the hip/hsa/memory trace domains).
out=$GRAFT_REPO_ROOT/gpurun_out/$1; ctr=$2; shift; shift; [ "$1" == "--" ] && shift
rm -rf $out; mkdir -p $out
prog=$1; shift
case "$prog" in /*) ;; *) prog=$GRAFT_REPO_ROOT/$prog;; esac
cd /tmp && export TMPDIR=/tmp
if [[ "$prog" == *.py ]]; then
  timeout -k 5 ${PMC_TIMEOUT:-900} rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -- python3 $prog "$@" > $out/stdout.txt 2> $out/stderr.txt
else
  timeout -k 5 ${PMC_TIMEOUT:-900} rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -- $prog "$@" > $out/stdout.txt 2> $out/stderr.txt
fi
rc=$?
cd $GRAFT_REPO_ROOT
python3 scratch/pmc_sum.py $out > $out/summary.txt 2>&1
find $out -name "*.csv" -size +1M -delete
echo "pmc pass $out [$ctr] rc=$rc"
exit 0
