#!/bin/bash
# round 6: longer side-by-side campaigns with the product's sources on the MI355X (default thresholds forced onto the small sets, and
# every anchor list let onto the resident route), then config 5 at full size five times (resident, same md5 every time)
mkdir -p gpurun_out/profiles_r06
PARSNP_FUZZ_CORE=hip timeout 1500 python scripts/fuzz_campaign.py 14000 14500 24 > gpurun_out/profiles_r06/fuzz_hip_long.log 2>&1; tail -2 gpurun_out/profiles_r06/fuzz_hip_long.log
PARSNP_FUZZ_CORE=hip PM_FLAGGED_DIV=1 timeout 1500 python scripts/fuzz_campaign.py 15000 15500 24 > gpurun_out/profiles_r06/fuzz_hip_long_flagged_div_1.log 2>&1; tail -2 gpurun_out/profiles_r06/fuzz_hip_long_flagged_div_1.log
for i in 1 2 3 4 5; do python scripts/why_route.py rearr500 2>&1 | grep "^rc"; done | cut -c1-200 | sort | uniq -c | tee gpurun_out/profiles_r06/rearr500_five_runs.txt
