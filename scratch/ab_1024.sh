#!/bin/bash
# 1024^3 (configs[3]) throughput against the sample size and the RIF layout
set -e
for spp in 8 32; do for lay in auto brick27; do
  echo "== spp $spp layout $lay"
  python bench.py --workload cfg4 --res 1024 --size 1024 --spp $spp --layout $lay --steps 1 --warmup 1 --no-cpu-baseline --no-target-512 --no-solo-step 2>&1 | tail -1
done; done
