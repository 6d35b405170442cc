#!/bin/bash
# quick A/B across grid sizes for the library given in MER_LIB (default libmer.so): scratch/ab_quick.sh [bench options]
for cfg in "--res 256" "--res 512" "--workload cfg4 --res 1024 --size 1024 --spp 8"; do [ -n "$AB_SKIP_1024" ] && [[ "$cfg" == *1024* ]] && continue
  echo "== ${MER_LIB:-libmer.so} $cfg $*"
  python bench.py $cfg --no-cpu-baseline --no-target-512 --steps 2 --warmup 1 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{})
print('value %.1f Mpaths/s  ms %.1f  solo march ms %.1f  event ms %.1f  launches %s  active %.3f' % (d['value'], d['ms_per_step'], r.get('kernel_ms_per_step',0), r.get('event_kernel_ms_per_step',0), r.get('launches_per_step'), r.get('active_lane_fraction',0)))"
done
