#!/bin/bash
# side-walk slots per kind: 3 (ships) vs 4; and pass length in the tail with spawned walks
P='import json,sys; d=json.loads(sys.stdin.read()); c=d["counters_per_step"]; print("%.1f Mpaths/s %.1f ms  in-lane %.1f %%" % (d["value"], d["ms_per_step"], 100*c["side_walks_in_the_paths_lane"]/max(c["side_walks_spawned"]+c["side_walks_in_the_paths_lane"],1)))'
B="python bench.py --no-cpu-baseline --no-target-512 --no-solo-step --no-live-pmc --steps 3 --warmup 1"
for lib in "" "mitsubaer_amd/libmer_sk4.so"; do
  for a in "--spp 256" "--spp 32" "--res 512 --steps 2" "--workload cfg4 --res 1024 --size 1024 --spp 8 --steps 2" "--workload cfg4 --res 1024 --size 1024 --spp 128 --steps 1"; do
    echo -n "slots/kind $( [ -z "$lib" ] && echo 3 || echo 4 ) $a: "; MER_LIB=${lib:+$PWD/$lib} $B $a 2>/dev/null | python -c "$P"
  done
done
for o in adaptive_k=2 "adaptive_k=0,ksteps=160"; do for a in "--spp 256" "--spp 32"; do echo -n "$o $a: "; $B $a --options $o 2>/dev/null | python -c "$P"; done; done
