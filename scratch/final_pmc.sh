#!/bin/bash
# Round-2 counter passes of the final kernels (GPU box, repo root): 256^3, 512^3 and 1024^3 (32 spp), single pipeline.
export PMC_TIMEOUT=300
bash scratch/pmc_all.sh cfg3_256 --res 256 --no-target-512
echo "256 done"
bash scratch/pmc_all.sh cfg3_512 --res 512 --no-target-512
echo "512 done"
bash scratch/pmc_all.sh cfg4_1024 --workload cfg4 --res 1024 --size 1024 --spp 32 --no-target-512
echo "1024 done"
