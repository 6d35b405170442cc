#!/bin/bash
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-target-512 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{})
print('cfg3 value %.1f Mpaths/s  ms %.1f  solo march ms %.1f event ms %.1f' % (d['value'], d['ms_per_step'], r.get('kernel_ms_per_step',0), r.get('event_kernel_ms_per_step',0)))"
python bench.py --workload cfg5 --spp 128 --steps 2 --warmup 1 --no-cpu-baseline --no-target-512 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{})
print('cfg5 value %.1f Mpaths/s  ms %.1f' % (d['value'], d['ms_per_step']))"
