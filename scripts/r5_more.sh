#!/bin/bash
# round 5, second evidence call: a fuzz campaign of the final kernels against the reference binary on the MI355X (resident route
# forced onto the small sets, every other seed with every anchor list on the route), the VALU calibration with the corrected launch
# shapes, and where a step of config 5 goes on the host route
mkdir -p gpurun_out/r5
O=gpurun_out/r5
timeout 120 parsnp_amd/bin/valu_calib > $O/valu_calib.json 2> $O/valu.err; python -c "
import json; d=json.load(open('$O/valu_calib.json')); print('valu ipc/simd', d['valu_int32_wave64_instructions_per_cycle_per_simd']); [print(r['chain'], r['waves_per_simd'], r['cycles_per_instruction_one_wave'], r['instructions_per_cycle_per_simd'], r['effective_ghz']) for r in d['runs']]"
PARSNP_FUZZ_CORE=hip timeout 900 python scripts/fuzz_campaign.py 6000 6250 12 > $O/fuzz_hip.log 2>&1; tail -3 $O/fuzz_hip.log
PARSNP_FUZZ_CORE=hip PM_FLAGGED_DIV=1 timeout 600 python scripts/fuzz_campaign.py 6300 6420 6 > $O/fuzz_hip_div1.log 2>&1; tail -3 $O/fuzz_hip_div1.log
PARSNP_BENCH_LOG=$O/laps_rearr500.log PARSNP_DEBUG_TIMERS=1 timeout 300 python bench.py --workload rearr500 --steps 1 --warmup 1 --cpu-sample 0 --other-configs off > $O/bench_rearr500_laps.json 2> /dev/null
grep -E "^\[(anchors|extend|filter_mums|lcb|chain|run_batch)" $O/laps_rearr500.log | tail -40
