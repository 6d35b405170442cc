#!/bin/bash
# builds the three variants of the reproducer (run them on an MI355X: ./pl_O3_branching; ./pl_O1_branching; ./pl_O3)
F="--offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-math-errno -fno-slp-vectorize -Wno-unused-function -Wno-unused-value"
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc $F -O3 -DMER_SDF_BRANCHING -o pl_O3_branching pl.hip &
/opt/rocm/bin/hipcc $F -O1 -DMER_SDF_BRANCHING -o pl_O1_branching pl.hip &
/opt/rocm/bin/hipcc $F -O3 -o pl_O3 pl.hip &
wait
