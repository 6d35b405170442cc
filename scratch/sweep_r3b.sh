#!/bin/bash
B="python bench.py --no-cpu-baseline --no-target-512 --no-solo-step --no-live-pmc --steps 3 --warmup 1"
P='import json,sys; d=json.loads(sys.stdin.read()); print("%.1f Mpaths/s %.1f ms" % (d["value"], d["ms_per_step"]))'
for opt in "pipes=4" "pipes=2" "pipes=3" "ksteps=96" "ksteps=160" "ksteps=192" "nslots=1048576" "nslots=3145728" "nslots=4194304"; do
  echo -n "256^3 256spp $opt: "; $B --options $opt 2>/dev/null | python -c "$P"
done
for opt in "pipes=4" "pipes=2" "ksteps=96" "nslots=262144" "small_render_slots=0"; do
  echo -n "256^3 32spp $opt: "; $B --spp 32 --options $opt 2>/dev/null | python -c "$P"
done
python bench.py --no-cpu-baseline --no-target-512 --no-solo-step --no-live-pmc --steps 1 --warmup 1 --spp 32 --options verbose=2 2> gpurun_out/tail_timeline_32spp_gridfit.txt > /dev/null
