#!/bin/bash
B="python bench.py --no-cpu-baseline --no-target-512 --no-solo-step --no-live-pmc --warmup 1"
P='import json,sys; d=json.loads(sys.stdin.read()); print("%.1f Mpaths/s %.1f ms" % (d["value"], d["ms_per_step"]))'
for lib in libmer_nopf.so libmer.so libmer_pf5.so; do
  export MER_LIB=$PWD/mitsubaer_amd/$lib
  echo -n "$lib 256^3 256spp: "; $B --steps 3 2>/dev/null | python -c "$P"
  echo -n "$lib 256^3 256spp pipes=1: "; $B --steps 3 --options pipes=1 2>/dev/null | python -c "$P"
  echo -n "$lib 512^3 256spp: "; $B --res 512 --steps 2 2>/dev/null | python -c "$P"
  echo -n "$lib 256^3 32spp: "; $B --spp 32 --steps 4 2>/dev/null | python -c "$P"
done
