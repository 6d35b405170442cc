#!/bin/bash
# K_march with every lane's current BRICK27 record staged in LDS (lds_bricks=1) against re-gathering cells through L1 / L2 (lds_bricks=0)
set -e
python -m pytest tests/test_gpu_render.py -x -q -m gpu -k "bit_identical or oracle or auto_layout" 2>&1 | tail -3
for o in lds_bricks=0 lds_bricks=1; do
  ./scratch/ab_quick.sh --options $o --layout brick27
done
