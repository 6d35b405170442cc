#!/bin/bash
B="python bench.py --no-cpu-baseline --no-target-512 --no-solo-step --no-live-pmc --warmup 1 --workload cfg4 --res 1024 --size 1024 --spp 128 --steps 1"
P='import json,sys; d=json.loads(sys.stdin.read()); print("%.1f Mpaths/s %.1f ms layout %s" % (d["value"], d["ms_per_step"], d["config"]["layout"]))'
for lay in auto cell8 brick27; do
  echo -n "1024^3 128spp --layout $lay: "; $B --layout $lay 2>/dev/null | python -c "$P"
done
echo -n "1024^3 128spp --layout brick27 mq_sort=1: "; $B --layout brick27 --options mq_sort=1 2>/dev/null | python -c "$P"
echo -n "1024^3 128spp --layout brick27 march_sort=4: "; $B --layout brick27 --options march_sort=4 2>/dev/null | python -c "$P"
