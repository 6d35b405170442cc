#!/bin/bash
# connection request lists sorted into 8 (ships) / 16 / 32 distance classes
for lib in libmer.so libmer_q16.so libmer_q32.so; do
echo "== $lib"
MER_LIB=$PWD/mitsubaer_amd/$lib python bench.py --workload cfg5 --spp 128 --steps 2 --warmup 1 --no-cpu-baseline --no-target-512 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['counters_per_step']
print('cfg5 value %.1f Mpaths/s  ms %.1f  lane eff %.3f' % (d['value'], d['ms_per_step'], c['connect_steps']/c['connect_lane_slots']))"
done
