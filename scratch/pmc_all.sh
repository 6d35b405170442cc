#!/bin/bash
# The counter passes of one workload (GPU box, repo root): scratch/pmc_all.sh <tag> <bench.py args...>
# Every pass is the same single-pipeline render (one bench step) under a different counter set; scratch/pmc_traffic.py merges them.
tag=$1; shift
B="bench.py $* --steps 1 --warmup 0 --no-cpu-baseline --no-solo-step --no-live-pmc --options pipes=1"
bash scratch/pmc_pass.sh pmc_${tag}_rd "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" -- $B
bash scratch/pmc_pass.sh pmc_${tag}_fetch "FETCH_SIZE" -- $B
bash scratch/pmc_pass.sh pmc_${tag}_write "WRITE_SIZE TCC_EA0_RDREQ_DRAM_sum" -- $B
bash scratch/pmc_pass.sh pmc_${tag}_l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUBBLE_sum" -- $B
bash scratch/pmc_pass.sh pmc_${tag}_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" -- $B
bash scratch/pmc_pass.sh pmc_${tag}_sq2 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE GRBM_COUNT" -- $B
