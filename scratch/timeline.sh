#!/bin/bash
python bench.py --no-cpu-baseline --no-target-512 --no-solo-step --no-live-pmc --steps 1 --warmup 1 --options verbose=2 2> gpurun_out/timeline_256.txt > /dev/null
grep "pipe 0" gpurun_out/timeline_256.txt | tail -9
python bench.py --no-cpu-baseline --no-target-512 --no-solo-step --no-live-pmc --steps 1 --warmup 1 --options verbose=2,pipes=1 2> gpurun_out/timeline_256_p1.txt > /dev/null
grep "pipe 0" gpurun_out/timeline_256_p1.txt | tail -10
