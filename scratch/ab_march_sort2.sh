#!/bin/bash
B="python bench.py --no-cpu-baseline --no-target-512 --no-solo-step --no-live-pmc --warmup 1"
P='import json,sys; d=json.loads(sys.stdin.read()); c=d["counters_per_step"]; print("%.1f Mpaths/s %.1f ms" % (d["value"], d["ms_per_step"]))'
for opt in "march_sort=0" "march_sort=3" "march_sort=0,mq_sort=0" "march_sort=3,mq_sort=0" "march_sort=4,mq_sort=0" "march_sort=2"; do
  echo -n "512^3 256spp $opt: "; $B --res 512 --steps 2 --options $opt 2>/dev/null | python -c "$P"
done
for opt in "march_sort=0" "march_sort=3" "march_sort=4" "march_sort=3,mq_sort=1" "march_sort=2,mq_sort=1" "march_sort=0,mq_sort=1"; do
  echo -n "1024^3 128spp $opt: "; $B --workload cfg4 --res 1024 --size 1024 --spp 128 --steps 1 --options $opt 2>/dev/null | python -c "$P"
done
for opt in "march_sort=0" "march_sort=4" "march_sort=3,mq_sort=1"; do
  echo -n "1024^3 8spp $opt: "; $B --workload cfg4 --res 1024 --size 1024 --spp 8 --steps 2 --options $opt 2>/dev/null | python -c "$P"
done
