#!/bin/bash
B="python bench.py --no-cpu-baseline --no-target-512 --no-solo-step --no-live-pmc --warmup 1"
P='import json,sys; d=json.loads(sys.stdin.read()); print("%.1f Mpaths/s %.1f ms" % (d["value"], d["ms_per_step"]))'
for lib in libmer.so libmer_ev3.so libmer_ev4.so; do for opt in "pipes=4" "pipes=1"; do
  echo -n "$lib 256^3 256spp $opt: "; MER_LIB=$PWD/mitsubaer_amd/$lib $B --steps 3 --options $opt 2>/dev/null | python -c "$P"
done; echo -n "$lib 256^3 32spp: "; MER_LIB=$PWD/mitsubaer_amd/$lib $B --steps 3 --spp 32 2>/dev/null | python -c "$P"; done
