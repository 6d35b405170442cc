#!/bin/bash
B="python bench.py --no-cpu-baseline --no-target-512 --no-solo-step --no-live-pmc --warmup 1"
P='import json,sys; d=json.loads(sys.stdin.read()); print("%.1f Mpaths/s %.1f ms" % (d["value"], d["ms_per_step"]))'
for lib in libmer.so libmer_w6.so; do for opt in "march_sort=0" "march_sort=2" "march_sort=2,pipes=1" "march_sort=0,pipes=1"; do
  echo -n "$lib 256^3 256spp $opt: "; MER_LIB=$PWD/mitsubaer_amd/$lib $B --steps 3 --options $opt 2>/dev/null | python -c "$P"
done; done
for opt in "march_sort=0" "march_sort=3" "march_sort=2" "march_sort=3,pipes=1" "march_sort=0,pipes=1"; do
  echo -n "512^3 256spp $opt: "; $B --res 512 --steps 2 --options $opt 2>/dev/null | python -c "$P"
done
for opt in "march_sort=0" "march_sort=3" "march_sort=4"; do
  echo -n "1024^3 128spp $opt: "; $B --workload cfg4 --res 1024 --size 1024 --spp 128 --steps 1 --options $opt 2>/dev/null | python -c "$P"
done
for opt in "march_sort=0" "march_sort=3"; do
  echo -n "1024^3 8spp $opt: "; $B --workload cfg4 --res 1024 --size 1024 --spp 8 --steps 2 --options $opt 2>/dev/null | python -c "$P"
  echo -n "256^3 32spp $opt: "; $B --spp 32 --steps 3 --options $opt 2>/dev/null | python -c "$P"
done
