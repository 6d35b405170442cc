#!/bin/bash
# A/B of K_march occupancy variants (libmer_w6.so / libmer_w8.so built with -DMER_MARCH_WAVES=6/8 for the cell8 / brick groups)
for lib in libmer.so libmer_w6.so libmer_w8.so; do
  export MER_LIB=$PWD/mitsubaer_amd/$lib
  for cfg in "--res 256" "--res 512" "--workload cfg4 --res 1024 --size 1024 --spp 8"; do
    echo "== $lib $cfg"
    python bench.py $cfg --no-cpu-baseline --no-target-512 --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{})
print('value %.1f Mpaths/s  ms %.1f  solo march ms %.1f  event ms %.1f  launches %s' % (d['value'], d['ms_per_step'], r.get('kernel_ms_per_step',0), r.get('event_kernel_ms_per_step',0), r.get('launches_per_step')))"
  done
done
