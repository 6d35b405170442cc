#!/bin/bash
# adaptive_k: 0 = fixed K, 1 = legacy (longer passes in the tail), 2 = shorter passes in the tail; on the headline job, an 8-GPU tile shard's worth of it, and 1024^3 at 8 spp
set -o pipefail
B="python bench.py --no-cpu-baseline --no-target-512 --no-solo-step --steps 3 --warmup 1"
P='import json,sys; d=json.loads(sys.stdin.read()); print("%.1f Mpaths/s %.1f ms" % (d["value"], d["ms_per_step"]))'
for opt in adaptive_k=1 adaptive_k=0 adaptive_k=2 "adaptive_k=0,ksteps=96" "adaptive_k=0,ksteps=64" "adaptive_k=2,ksteps=96"; do
  echo -n "cfg3 256^3 256 spp $opt: "; $B --spp 256 --options $opt 2>/dev/null | python -c "$P"
  echo -n "cfg3 256^3  32 spp $opt: "; $B --spp 32 --options $opt 2>/dev/null | python -c "$P"
done
for opt in adaptive_k=1 adaptive_k=0 adaptive_k=2; do
  echo -n "cfg3 512^3 256 spp $opt: "; $B --res 512 --spp 256 --steps 2 --options $opt 2>/dev/null | python -c "$P"
  echo -n "cfg4 1024^3 8 spp $opt: "; $B --workload cfg4 --res 1024 --size 1024 --spp 8 --steps 2 --options $opt 2>/dev/null | python -c "$P"
done
