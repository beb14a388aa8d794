#!/bin/bash
# round 5, last call: the whole GPU suite, the profiler passes + default bench line + floor (scripts/round_profile.sh), the host's
# chatter of the other workloads (scripts/r5_why.sh), and a side-by-side campaign with every anchor list forced onto the
# resident route (rearranged small sets through the trimming kernels, the exact cluster test and the order check)
mkdir -p gpurun_out/final gpurun_out/profiles_r05
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/final/suite.log
bash scripts/round_profile.sh 2>&1 | tail -12
bash scripts/r5_why.sh 2>&1 | tail -6
PM_FLAGGED_DIV=1 PARSNP_FUZZ_CORE=hip timeout 600 python scripts/fuzz_campaign.py 6600 6760 6 > gpurun_out/profiles_r05/fuzz_hip_final_div1.log 2>&1; tail -1 gpurun_out/profiles_r05/fuzz_hip_final_div1.log
