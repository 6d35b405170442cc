#!/bin/bash
# Final plain runs of round 2 (GPU box, repo root) -> gpurun_out/final_runs/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_runs; mkdir -p $O
set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json | cut -c1-200
python bench.py --workload cfg5 --spp 512 --steps 1 --warmup 1 --no-target-512 > $O/bench_cfg5_spp512.json 2> $O/cfg5.err; tail -1 $O/bench_cfg5_spp512.json | cut -c1-200
python bench.py --workload cfg4 --res 1024 --size 1024 --spp 128 --steps 1 --warmup 1 --no-target-512 --no-cpu-baseline > $O/bench_cfg4_1024_spp128.json 2> $O/cfg4.err; tail -1 $O/bench_cfg4_1024_spp128.json | cut -c1-200
python bench.py --workload cfg2 --spp 64 --no-target-512 > $O/bench_cfg2.json 2> $O/cfg2.err; tail -1 $O/bench_cfg2.json | cut -c1-200
python bench.py --gpus 2 --backend gloo --device 0 --steps 2 --warmup 1 --no-target-512 --no-cpu-baseline > $O/bench_2ranks_one_gpu_gloo.json 2> $O/2ranks.err; tail -1 $O/bench_2ranks_one_gpu_gloo.json | cut -c1-300
python bench.py --gpus 2 --backend gloo --device 0 --scaling strong --steps 2 --warmup 1 --no-target-512 --no-cpu-baseline > $O/bench_2ranks_strong_gloo.json 2> $O/2ranks_s.err; tail -1 $O/bench_2ranks_strong_gloo.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_p1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p1 -- python3 $R/bench.py --options pipes=1 --no-target-512 --no-cpu-baseline > $O/bench_pipes1_under_rocprof.json 2> $O/p1.err
cp $(find /tmp/prof_p1 -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_cfg3_256_pipes1.csv
tail -1 $O/bench_pipes1_under_rocprof.json | cut -c1-200; head -3 $O/rocprofv3_kernel_stats_cfg3_256_pipes1.csv
