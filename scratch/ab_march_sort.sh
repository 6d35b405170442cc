#!/bin/bash
# A/B of the spatial sort of the march list (option march_sort = bits per axis, march_sort_major): Mpaths/s, ms per step, K_march / K_event ms summed
B="python bench.py --no-cpu-baseline --no-target-512 --no-solo-step --no-live-pmc --steps 3 --warmup 1"
P='import json,sys; d=json.loads(sys.stdin.read()); c=d["counters_per_step"]; print("%.1f Mpaths/s %.1f ms  steps %.4g" % (d["value"], d["ms_per_step"], c["eikonal_steps"]))'
for opt in "march_sort=0" "march_sort=2" "march_sort=2,march_sort_major=1" "march_sort=1" "march_sort=3" "march_sort=3,march_sort_major=1" "march_sort=2,mq_sort=0" "march_sort=2,pipes=1" "march_sort=0,pipes=1"; do
  echo -n "256^3 256spp $opt: "; $B --options $opt 2>/dev/null | python -c "$P"
done
for opt in "march_sort=0" "march_sort=2" "march_sort=3" "march_sort=3,march_sort_major=1"; do
  echo -n "512^3 256spp $opt: "; $B --res 512 --steps 2 --options $opt 2>/dev/null | python -c "$P"
done
for opt in "march_sort=0" "march_sort=2"; do
  echo -n "256^3 32spp $opt: "; $B --spp 32 --options $opt 2>/dev/null | python -c "$P"
done
