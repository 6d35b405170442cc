#!/bin/bash
# headline / single pipeline / an eighth of the job / 512^3 / 1024^3 x 8 and x 128 spp / configs[1] / configs[4]: Mpaths/s and ms per step
B="python bench.py --no-cpu-baseline --no-target-512 --no-solo-step --no-live-pmc --warmup 1"
P='import json,sys; d=json.loads(sys.stdin.read()); print("%.1f Mpaths/s %.1f ms" % (d["value"], d["ms_per_step"]))'
O="$1"
echo -n "256^3 256spp: "; $B --steps 3 ${O:+--options $O} 2>/dev/null | python -c "$P"
echo -n "256^3 256spp pipes=1: "; $B --steps 3 --options pipes=1${O:+,$O} 2>/dev/null | python -c "$P"
echo -n "256^3 32spp: "; $B --spp 32 --steps 4 ${O:+--options $O} 2>/dev/null | python -c "$P"
echo -n "512^3 256spp: "; $B --res 512 --steps 2 ${O:+--options $O} 2>/dev/null | python -c "$P"
echo -n "1024^3 8spp: "; $B --workload cfg4 --res 1024 --size 1024 --spp 8 --steps 2 ${O:+--options $O} 2>/dev/null | python -c "$P"
echo -n "1024^3 128spp: "; $B --workload cfg4 --res 1024 --size 1024 --spp 128 --steps 1 ${O:+--options $O} 2>/dev/null | python -c "$P"
echo -n "cfg2 64spp: "; $B --workload cfg2 --spp 64 --steps 5 ${O:+--options $O} 2>/dev/null | python -c "$P"
echo -n "cfg5 128spp: "; $B --workload cfg5 --spp 128 --steps 2 ${O:+--options $O} 2>/dev/null | python -c "$P"
