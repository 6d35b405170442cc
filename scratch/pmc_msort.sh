#!/bin/bash
# L2 hit rate and fabric read requests of K_march with / without the spatial sort of the march list (single-pipeline step)
one() { tag=$1; opts=$2; shift 2
  B="bench.py $* --steps 1 --warmup 0 --no-cpu-baseline --no-solo-step --no-live-pmc --options pipes=1,$opts"
  bash scratch/pmc_pass.sh pmc_${tag}_rd "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" -- $B
  bash scratch/pmc_pass.sh pmc_${tag}_l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUBBLE_sum" -- $B
  echo "== $tag ($opts)"; grep -A6 "^march_kernel" gpurun_out/pmc_${tag}_rd/summary.txt | head -7; grep -A5 "^march_kernel" gpurun_out/pmc_${tag}_l2/summary.txt | head -6
}
export PMC_TIMEOUT=200
one ms0_256 march_sort=0 --res 256 --no-target-512
one ms2_256 march_sort=2 --res 256 --no-target-512
one ms0_512 march_sort=0 --res 512 --no-target-512
one ms3_512 march_sort=3 --res 512 --no-target-512
