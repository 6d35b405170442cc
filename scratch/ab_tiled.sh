#!/bin/bash
# CELL8 records in 2x2x2-tile order (libmer_tiled.so: -DMER_CELL8_TILED) against x-major order, --layout cell8
MER_LIB=$PWD/mitsubaer_amd/libmer_tiled.so python -m pytest tests/test_gpu_render.py tests/test_gpu_leaf.py -x -q -m gpu -k "cell8 or bit_identical or global_load or lookup" 2>&1 | tail -3
for lib in libmer.so libmer_tiled.so; do
  for cfg in "--res 256" "--res 512" "--workload cfg4 --res 1024 --size 1024 --spp 32"; do
    echo "== $lib $cfg"
    MER_LIB=$PWD/mitsubaer_amd/$lib python bench.py $cfg --layout cell8 --no-cpu-baseline --no-target-512 --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{})
print('value %.1f Mpaths/s  ms %.1f  solo march ms %.1f  launches %s' % (d['value'], d['ms_per_step'], r.get('kernel_ms_per_step',0), r.get('launches_per_step')))"
  done
done
