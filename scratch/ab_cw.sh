#!/bin/bash
# K_connect at 2 / 3 (ships) / 4 waves per SIMD after the scratch copy of the kernel arguments was removed
for lib in libmer.so libmer_c4.so libmer_c5.so libmer_c6.so; do
echo "== $lib"
MER_LIB=$PWD/mitsubaer_amd/$lib python bench.py --workload cfg5 --spp 128 --steps 2 --warmup 1 --no-cpu-baseline --no-target-512 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg5 value %.1f Mpaths/s  ms %.1f' % (d['value'], d['ms_per_step']))"
done
