#!/bin/bash
B="python bench.py --no-cpu-baseline --no-target-512 --no-solo-step --no-live-pmc --steps 3 --warmup 1"
P='import json,sys; d=json.loads(sys.stdin.read()); print("%.1f Mpaths/s %.1f ms" % (d["value"], d["ms_per_step"]))'
for opt in "ksteps=128" "ksteps=96" "ksteps=160" "ksteps=192" "ksteps=256" "ksteps=128,nslots=1048576" "ksteps=128,nslots=3145728" "ksteps=128,nslots=4194304" "ksteps=128,pipes=3" "ksteps=128,pipes=2" "ksteps=192,pipes=2" "ksteps=128,pipes=1" "ksteps=128,mq_sort=0"; do
  echo -n "256^3 256spp $opt: "; $B --options $opt 2>/dev/null | python -c "$P"
done
for opt in "ksteps=128" "ksteps=192" "ksteps=128,nslots=1048576" "ksteps=128,pipes=2"; do
  echo -n "256^3 32spp $opt: "; $B --spp 32 --options $opt 2>/dev/null | python -c "$P"
done
