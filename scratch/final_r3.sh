#!/bin/bash
# Round-3 final measurement batch (GPU box, repo root): tests, smoke, bench lines, rocprofv3 stats, committed counter passes.
O=gpurun_out/final_r3; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
python bench.py --workload cfg2 --spp 64 --cpu-seconds 8 > $O/bench_cfg2.json 2>/dev/null
python bench.py --workload cfg5 --spp 512 --steps 2 --warmup 1 --cpu-seconds 8 > $O/bench_cfg5_spp512.json 2>/dev/null
python bench.py --workload cfg5 --spp 128 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cfg5_spp128.json 2>/dev/null
python bench.py --workload cfg4 --res 1024 --size 1024 --spp 128 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_cfg4_1024_spp128.json 2>/dev/null
python bench.py --workload cfg4 --res 1024 --size 1024 --spp 8 --steps 2 --warmup 1 --no-cpu-baseline --no-solo-step > $O/bench_cfg4_1024_spp8.json 2>/dev/null
python bench.py --res 512 --no-cpu-baseline --steps 2 > $O/bench_cfg3_512.json 2>/dev/null
for f in bench_default bench_cfg2 bench_cfg5_spp512 bench_cfg5_spp128 bench_cfg4_1024_spp128 bench_cfg4_1024_spp8 bench_cfg3_512; do python -c "
import json; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$f: %.1f Mpaths/s %.1f ms frac %s' % (d['value'], d['ms_per_step'], r.get('frac')))"; done
timeout -k 10 600 scratch/r3_stats.sh
export PMC_TIMEOUT=300
bash scratch/pmc_all.sh cfg3_256 --res 256 --no-target-512 > $O/pmc.log 2>&1
bash scratch/pmc_all.sh cfg3_512 --res 512 --no-target-512 >> $O/pmc.log 2>&1
bash scratch/pmc_all.sh cfg4_1024 --workload cfg4 --res 1024 --size 1024 --spp 128 --no-target-512 >> $O/pmc.log 2>&1
bash scratch/pmc_all.sh cfg2_256 --workload cfg2 --spp 64 --no-target-512 >> $O/pmc.log 2>&1
grep -c "rc=0" $O/pmc.log
