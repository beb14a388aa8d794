#!/bin/bash
# round 5: the loop of a change on the GPU box -- bounded smoke run, the parity tests of the paths touched, 200 x 5 Mb against the
# reference's golden, a bench line with the host's laps, A/B runs of the new switches, the kernel table + the device's idle gaps
mkdir -p gpurun_out/r5
O=gpurun_out/r5
timeout 120 python scripts/smoke_core.py 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "master_ep or resident_route or batched or random_regions or test_events or small_regions or medium or properties" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_big.py -m gpu -x -q -k "bact200" 2>&1 | tail -4
B="timeout 300 python bench.py --steps 60 --warmup 5 --cpu-sample 0 --other-configs off"
PARSNP_BENCH_LOG=$O/laps.log PARSNP_DEBUG_TIMERS=1 timeout 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --other-configs off > /dev/null 2> $O/laps.err
$B > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | python scripts/benchline.py
$B --tune master_seg=0 > $O/bench_master0.json 2> /dev/null; tail -1 $O/bench_master0.json | python scripts/benchline.py | head -1
$B --tune timing=0 > $O/bench_timing0.json 2> /dev/null; tail -1 $O/bench_timing0.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('timing=0', d['value'], d['ms_per_step'])"
PARSNP_NO_DEVICE_CHAIN=1 PARSNP_SPLIT_SETTLE=1 PARSNP_ONE_STAGE=1 $B > $O/bench_r4forms.json 2> /dev/null; tail -1 $O/bench_r4forms.json | python scripts/benchline.py | head -1
bash scripts/profile_stats.sh > $O/stats.log 2>&1; head -30 gpurun_out/prof_stats/summary/kernel_stats.csv | cut -c1-160; cp gpurun_out/prof_stats/summary/idle_gaps.json $O/ 2>/dev/null
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r5/idle_gaps.json"))
    print("busy %.2f ms idle %.2f ms" % (d["kernel_busy_ms"], d["idle_ms_in_gaps_below_50ms"]))
    for g in d["gaps"][:25]: print("  %-70s n=%d total %.0f us max %.0f us" % (g["pair"][:70], g["count"], g["total_us"], g["max_us"]))
except Exception as e: print("no gaps", e)
PY
timeout 120 parsnp_amd/bin/valu_calib > $O/valu_calib.json 2> $O/valu.err; python -c "
import json; d=json.load(open('$O/valu_calib.json')); print('valu ipc/simd', d['valu_int32_wave64_instructions_per_cycle_per_simd']); [print(r) for r in d['runs'] if r['waves_per_simd'] in (1,4,8)]"
timeout 400 python bench.py --workload rearr500 --steps 3 --warmup 1 --cpu-sample 0 --other-configs off > $O/bench_rearr500.json 2> $O/bench_rearr500.err; tail -1 $O/bench_rearr500.json | python scripts/benchline.py | head -2
