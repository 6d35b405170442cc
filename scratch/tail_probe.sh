#!/bin/bash
# drain timeline (option verbose = 2) and slot / pipeline sweep of small renders: where does a render with few paths per slot spend its time?
set -o pipefail
B="python bench.py --no-cpu-baseline --no-target-512 --no-solo-step --steps 2 --warmup 1"
echo "== cfg3 256^3, 512^2 x 32 spp (one tile-shard's worth of an 8-GPU strong job), verbose=2 =="
$B --spp 32 --options verbose=2 2> gpurun_out/tail_cfg3_32spp_timeline.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
for opt in "pipes=4" "pipes=2" "pipes=1" "pipes=4,nslots=1048576" "pipes=4,nslots=524288" "pipes=2,nslots=524288" "pipes=1,nslots=524288" "pipes=1,nslots=262144" "pipes=4,ksteps=64" "pipes=4,ksteps=256" "pipes=4,adaptive_k=0"; do
  echo -n "cfg3 256^3 32 spp $opt: "; $B --spp 32 --options $opt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Mpaths/s %.1f ms' % (d['value'], d['ms_per_step']))"
done
echo "== cfg4 1024^3, 1024^2 x 8 spp =="
C="$B --workload cfg4 --res 1024 --size 1024 --spp 8"
$C --options verbose=2 2> gpurun_out/tail_cfg4_8spp_timeline.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
for opt in "pipes=4" "pipes=1" "pipes=4,nslots=1048576" "pipes=4,nslots=524288" "pipes=1,nslots=524288" "pipes=2,nslots=524288"; do
  echo -n "cfg4 1024^3 8 spp $opt: "; $C --options $opt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Mpaths/s %.1f ms' % (d['value'], d['ms_per_step']))"
done
