#!/bin/bash
# Round-end evidence run (on the GPU box, from the repo root): GPU tests, smoke, bench line, rocprofv3 kernel stats.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
python bench.py > $O/bench_cfg3_n1.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg3_n1_under_rocprof.json 2> $O/rocprof.err; echo "rocprof rc=$?"
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/rocprofv3_kernel_stats_cfg3_n1.csv
find $O/prof -name "*.csv" -size +2M -delete
python bench.py --workload cfg5 --spp 32 --steps 1 --warmup 1 > $O/bench_cfg5_n1.json 2>> $O/bench.err; echo "cfg5 rc=$?"
python bench.py --workload cfg2 --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg2_n1.json 2>> $O/bench.err; echo "cfg2 rc=$?"
