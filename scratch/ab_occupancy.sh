#!/bin/bash
# K_march occupancy capped BELOW its 5 waves per SIMD by requesting unused dynamic LDS (march_lds_kb): 33 KiB -> 4 blocks per CU, 41 -> 3, 54 -> 2
for kb in 0 33 41 54; do
  for cfg in "--res 256" "--res 512" "--workload cfg4 --res 1024 --size 1024 --spp 32"; do
    echo "== march_lds_kb=$kb $cfg"
    python bench.py $cfg --no-cpu-baseline --no-target-512 --steps 2 --warmup 1 --options march_lds_kb=$kb 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{})
print('value %.1f Mpaths/s  ms %.1f  solo march ms %.1f  launches %s' % (d['value'], d['ms_per_step'], r.get('kernel_ms_per_step',0), r.get('launches_per_step')))"
  done
done
