#!/bin/bash
# two more side-by-side campaigns with the product's sources on the MI355X (no source has changed since profiles/r05 was stamped)
mkdir -p gpurun_out/profiles_r05
PARSNP_FUZZ_CORE=hip timeout 500 python scripts/fuzz_campaign.py 6800 7000 8 > gpurun_out/profiles_r05/fuzz_hip_final_2.log 2>&1; tail -1 gpurun_out/profiles_r05/fuzz_hip_final_2.log
PM_FLAGGED_DIV=1 PARSNP_FUZZ_CORE=hip timeout 500 python scripts/fuzz_campaign.py 7000 7200 8 > gpurun_out/profiles_r05/fuzz_hip_final_div1_2.log 2>&1; tail -1 gpurun_out/profiles_r05/fuzz_hip_final_div1_2.log
