#!/bin/bash
# Which stage of the vector-memory path binds K_march?  TA / TCP counter passes (two counters of a block per pass: more than the block's
# counter slots makes rocprofv3 abort) of one single-pipeline bench step: scratch/pmc_ta.sh <tag> <options> <bench args>
tag=$1; opts=$2; shift; shift
export PMC_TIMEOUT=200
B="bench.py $* --steps 1 --warmup 0 --no-cpu-baseline --no-solo-step --no-target-512 --options pipes=1,$opts"
i=0
for c in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum"; do
  i=$((i+1))
  bash scratch/pmc_pass.sh pmc_${tag}_v$i "$c" -- $B
  grep -A3 "march_kernel" gpurun_out/pmc_${tag}_v$i/summary.txt | head -4
done
