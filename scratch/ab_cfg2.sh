#!/bin/bash
B="python bench.py --no-cpu-baseline --no-target-512 --no-solo-step --no-live-pmc --warmup 3 --workload cfg2 --spp 64 --steps 30"
P='import json,sys; d=json.loads(sys.stdin.read()); print("%.1f Mpaths/s %.2f ms" % (d["value"], d["ms_per_step"]))'
for lib in libmer.so libmer_t.so libmer.so libmer_t.so; do
  echo -n "cfg2 64spp $lib: "; MER_LIB=$PWD/mitsubaer_amd/$lib $B 2>/dev/null | python -c "$P"
done
