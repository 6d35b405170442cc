#!/bin/bash
# GPU tests, then the bench line with 4 pipelines and with 1
mkdir -p gpurun_out/ab
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/ab/tests.log 2>&1 || { tail -40 gpurun_out/ab/tests.log | cut -c1-300; exit 1; }
tail -2 gpurun_out/ab/tests.log
timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab/cfg3_p4.json 2> gpurun_out/ab/cfg3_p4.err || exit 1
MER_PIPES=1 timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab/cfg3_p1.json 2> gpurun_out/ab/cfg3_p1.err || exit 1
timeout -k 10 200 python bench.py --res 512 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ab/cfg3_512_p4.json 2> gpurun_out/ab/cfg3_512_p4.err || exit 1
