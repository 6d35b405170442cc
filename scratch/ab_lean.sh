#!/bin/bash
# K_march with the never-changing hot words left in the slot (80 VGPR, 6 waves/SIMD, no spills; libmer_lean.so: -DMER_MARCH_LEAN) against 86 VGPR / 5 waves
MER_LIB=$PWD/mitsubaer_amd/libmer_lean.so python -m pytest tests/test_gpu_render.py tests/test_gpu_fullsize.py -x -q -m gpu -k "not 1024" 2>&1 | tail -3
for lib in libmer.so libmer_lean.so; do
  for cfg in "--res 256" "--res 512" "--workload cfg4 --res 1024 --size 1024 --spp 32"; do
    echo "== $lib $cfg"
    MER_LIB=$PWD/mitsubaer_amd/$lib python bench.py $cfg --no-cpu-baseline --no-target-512 --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{})
print('value %.1f Mpaths/s  ms %.1f  solo march ms %.1f  launches %s' % (d['value'], d['ms_per_step'], r.get('kernel_ms_per_step',0), r.get('launches_per_step')))"
  done
done
