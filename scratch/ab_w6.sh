#!/bin/bash
B="python bench.py --no-cpu-baseline --no-target-512 --no-solo-step --no-live-pmc --steps 3 --warmup 1"
P='import json,sys; d=json.loads(sys.stdin.read()); print("%.1f Mpaths/s %.1f ms" % (d["value"], d["ms_per_step"]))'
for lib in libmer.so libmer_w6.so; do for opt in "march_sort=0" "march_sort=2" "march_sort=2,pipes=1" "march_sort=0,pipes=1"; do
  echo -n "$lib 256^3 256spp $opt: "; MER_LIB=$PWD/mitsubaer_amd/$lib $B --options $opt 2>/dev/null | python -c "$P"
done; done
for lib in libmer.so libmer_w6.so; do for opt in "march_sort=0" "march_sort=3"; do
  echo -n "$lib 512^3 256spp $opt: "; MER_LIB=$PWD/mitsubaer_amd/$lib $B --res 512 --steps 2 --options $opt 2>/dev/null | python -c "$P"
done; done
