#!/bin/bash
# headline bench against the slot count: per pipeline a whole number of resident K_march rounds (1280 blocks x 256 lanes = 327680) or the default 524288
for opt in "pipes=4" "pipes=4,nslots=1310720" "pipes=4,nslots=2621440" "pipes=4,nslots=3932160" "pipes=3,nslots=1966080" "pipes=2,nslots=1310720" "pipes=4,nslots=2621440,ksteps=96" "pipes=4,nslots=2621440,ksteps=160"; do
echo "== $opt"
python bench.py --no-cpu-baseline --no-target-512 --no-solo-step --steps 3 --warmup 1 --options $opt 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.1f Mpaths/s  ms %.1f' % (d['value'], d['ms_per_step']))"
done
