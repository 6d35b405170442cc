#!/bin/bash
# K_connect: further solver units in the same launch while a lane has taken fewer than connect_budget steps (0 = one unit per launch); x connect_launches
python -m pytest tests/test_gpu_render.py tests/test_gpu_leaf.py -x -q -m gpu -k "point or connect" 2>&1 | tail -2
for opt in "connect_budget=0" "connect_budget=150" "connect_budget=300" "connect_budget=600" "connect_budget=300,connect_launches=1" "connect_budget=600,connect_launches=1" "connect_budget=1200,connect_launches=1"; do
echo "== $opt"
python bench.py --workload cfg5 --spp 128 --steps 2 --warmup 1 --no-cpu-baseline --no-target-512 --options $opt 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['counters_per_step']
print('cfg5 value %.1f Mpaths/s  ms %.1f  lane eff %.3f units/conn %.2f' % (d['value'], d['ms_per_step'], c['connect_steps']/c['connect_lane_slots'], c['connect_units']/c['connections']))"
done
