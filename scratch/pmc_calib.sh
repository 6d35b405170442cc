#!/bin/bash
# counter passes of the calibration microbenchmark (known byte counts): scratch/pmc_calib.sh
bash scratch/pmc_pass.sh pmc_calib_rd "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" -- scratch/ubench/hbm_calib
bash scratch/pmc_pass.sh pmc_calib_fetch "FETCH_SIZE" -- scratch/ubench/hbm_calib
bash scratch/pmc_pass.sh pmc_calib_l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUBBLE_sum" -- scratch/ubench/hbm_calib
bash scratch/pmc_pass.sh pmc_calib_dram "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum" -- scratch/ubench/hbm_calib
