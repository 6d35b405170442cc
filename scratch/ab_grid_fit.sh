#!/bin/bash
B="python bench.py --no-cpu-baseline --no-target-512 --no-solo-step --no-live-pmc --warmup 1"
P='import json,sys; d=json.loads(sys.stdin.read()); print("%.1f Mpaths/s %.1f ms" % (d["value"], d["ms_per_step"]))'
for opt in "grid_fit=0" "grid_fit=1" "grid_fit=0,pipes=1" "grid_fit=1,pipes=1"; do
  echo -n "256^3 256spp $opt: "; $B --steps 3 --options $opt 2>/dev/null | python -c "$P"
done
for opt in "grid_fit=0" "grid_fit=1"; do
  echo -n "256^3 32spp $opt: "; $B --spp 32 --steps 3 --options $opt 2>/dev/null | python -c "$P"
  echo -n "512^3 256spp $opt: "; $B --res 512 --steps 2 --options $opt 2>/dev/null | python -c "$P"
  echo -n "1024^3 8spp $opt: "; $B --workload cfg4 --res 1024 --size 1024 --spp 8 --steps 2 --options $opt 2>/dev/null | python -c "$P"
  echo -n "1024^3 128spp $opt: "; $B --workload cfg4 --res 1024 --size 1024 --spp 128 --steps 1 --options $opt 2>/dev/null | python -c "$P"
  echo -n "cfg2 64spp $opt: "; $B --workload cfg2 --spp 64 --steps 5 --options $opt 2>/dev/null | python -c "$P"
  echo -n "cfg5 128spp $opt: "; $B --workload cfg5 --spp 128 --steps 2 --options $opt 2>/dev/null | python -c "$P"
done
