#!/bin/bash
B="python bench.py --no-cpu-baseline --no-target-512 --no-solo-step --no-live-pmc --warmup 1"
P='import json,sys; d=json.loads(sys.stdin.read()); print("%.1f Mpaths/s %.2f ms" % (d["value"], d["ms_per_step"]))'
for opt in "check_every=8" "check_every=4" "check_every=2" "check_every=16" "check_every=8" "check_every=4"; do
  echo -n "256^3 256spp $opt: "; $B --steps 3 --options $opt 2>/dev/null | python -c "$P"
  echo -n "256^3 32spp $opt: "; $B --spp 32 --steps 4 --options $opt 2>/dev/null | python -c "$P"
  echo -n "cfg2 64spp $opt: "; $B --workload cfg2 --spp 64 --steps 20 --options $opt 2>/dev/null | python -c "$P"
done
