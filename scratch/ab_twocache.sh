#!/bin/bash
# two-cache RK4 march (mid + end cells gathered together) against the one-cache form (libmer_onecache.so: -DMER_ONE_CELL_CACHE)
set -e
python -m pytest tests/test_gpu_render.py -x -q -m gpu -k "bit_identical or oracle or auto_layout" 2>&1 | tail -3
for lib in libmer_onecache.so libmer.so; do
  MER_LIB=$PWD/mitsubaer_amd/$lib ./scratch/ab_quick.sh
done
