#!/bin/bash
# configs[4] (curved-ray point-emitter NEE): K_connect launches per pass / pipelines / sample counts.  usage: scratch/ab_cfg5.sh "<bench args>" ...
for a in "$@"; do
  echo "== cfg5 $a"
  python bench.py --workload cfg5 --steps 1 --warmup 1 --no-cpu-baseline --no-solo-step $a 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['counters_per_step']
print('value %.2f Mpaths/s  ms %.0f  connections %.3g  units/conn %.1f  steps/unit %.0f  K_connect active-lane fraction %.2f  solver steps/s %.3g' % (d['value'], d['ms_per_step'], c['connections'], c['connect_units']/max(c['connections'],1), c['connect_steps']/max(c['connect_units'],1), c['connect_steps']/max(c['connect_lane_slots'],1), c['connect_steps']/d['ms_per_step']*1e3))"
done
