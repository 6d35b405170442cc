#!/bin/bash
# Round-end evidence run (on the GPU box, from the repo root): GPU tests, smoke, bench lines, rocprofv3 kernel stats, PMC traffic.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/gpu_tests.log; [ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc"; tail -2 $O/smoke.log; [ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
export BENCH_NO_SOLO_STEP=1      # profiled runs: the timed steps only (no extra single-pipeline step)
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_pmc_fetch.json 2> $O/pmc_fetch.err; rc=$?; echo "pmc fetch rc=$rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_pmc_write.json 2> $O/pmc_write.err; rc=$?; echo "pmc write rc=$rc"; [ $rc -eq 0 ] || exit $rc
cd $R
python scratch/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/bench_pmc_fetch.json $O/pmc_traffic_cfg3_n1.json; rc=$?; echo "traffic rc=$rc"; [ $rc -eq 0 ] || exit $rc
# bench.py reads the HBM traffic of the dominant kernel from profiles/round1/: refresh it before the bench lines are written
cp $O/pmc_traffic_cfg3_n1.json $R/profiles/round1/pmc_traffic_cfg3_n1.json
unset BENCH_NO_SOLO_STEP
python bench.py > $O/bench_cfg3_n1.json 2> $O/bench.err; rc=$?; echo "bench rc=$rc"; [ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
export BENCH_NO_SOLO_STEP=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg3_n1_under_rocprof.json 2> $O/rocprof.err; rc=$?; echo "rocprof rc=$rc"; [ $rc -eq 0 ] || exit $rc
cd $R; unset BENCH_NO_SOLO_STEP
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/rocprofv3_kernel_stats_cfg3_n1.csv
find $O -name "*.csv" -size +2M -delete
python bench.py --res 512 --steps 2 --warmup 1 > $O/bench_cfg3_512_n1.json 2>> $O/bench.err; rc=$?; echo "512 rc=$rc"; [ $rc -eq 0 ] || exit $rc
python bench.py --workload cfg5 --spp 32 --steps 1 --warmup 1 > $O/bench_cfg5_n1.json 2>> $O/bench.err; rc=$?; echo "cfg5 rc=$rc"; [ $rc -eq 0 ] || exit $rc
python bench.py --workload cfg2 --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg2_n1.json 2>> $O/bench.err; rc=$?; echo "cfg2 rc=$rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --workload cfg4 --res 1024 --size 1024 --spp 8 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cfg4_1gpu_8spp.json 2>> $O/bench.err; rc=$?; echo "cfg4 rc=$rc"; [ $rc -eq 0 ] || exit $rc
