#!/bin/bash
# BRICK27 cell changes fetched as four x-rows of the brick (12 floats, same 4 load instructions): an x move inside the brick needs no memory access (libmer_xrow.so)
MER_LIB=$PWD/mitsubaer_amd/libmer_xrow.so python -m pytest tests/test_gpu_render.py tests/test_gpu_fullsize.py -x -q -m gpu -k "not 1024" 2>&1 | tail -2
for lib in libmer.so libmer_xrow.so; do
  for cfg in "--res 256" "--res 512"; do
    echo "== $lib $cfg"
    MER_LIB=$PWD/mitsubaer_amd/$lib python bench.py $cfg --no-cpu-baseline --no-target-512 --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{})
print('value %.1f Mpaths/s  ms %.1f  solo march ms %.1f  launches %s' % (d['value'], d['ms_per_step'], r.get('kernel_ms_per_step',0), r.get('launches_per_step')))"
  done
done
