#!/bin/bash
B="python bench.py --no-cpu-baseline --no-target-512 --no-solo-step --no-live-pmc --warmup 1"
P='import json,sys; d=json.loads(sys.stdin.read()); print("%.1f Mpaths/s %.1f ms" % (d["value"], d["ms_per_step"]))'
for lib in libmer.so libmer_ab2.so libmer_ab8.so libmer.so; do
  export MER_LIB=$PWD/mitsubaer_amd/$lib
  echo -n "$lib 256^3 256spp: "; $B --steps 3 2>/dev/null | python -c "$P"
  echo -n "$lib 512^3 256spp: "; $B --res 512 --steps 2 2>/dev/null | python -c "$P"
done
