#!/bin/bash
# round 4, first GPU contact of the resident route:  gpurun --timeout 1200 -- 'bash scripts/r4_check1.sh'
mkdir -p gpurun_out
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "resident or synthetic or replay_modes or device_rows" 2>&1 | tail -15 > $O/r4_t1.log; tail -3 $O/r4_t1.log
timeout 400 python -m pytest tests/test_fuzz_vs_reference.py -m gpu -x -q -k "resident_route_on_gpu" 2>&1 | tail -15 > $O/r4_t2.log; tail -3 $O/r4_t2.log
timeout 300 python bench.py --steps 40 --warmup 5 --cpu-sample 0 > $O/r4_bench40.json 2> $O/r4_bench40.err; tail -1 $O/r4_bench40.json | python scripts/benchline.py
PARSNP_BENCH_LOG=$O/r4_laps.log PARSNP_DEBUG_TIMERS=1 timeout 300 python bench.py --steps 3 --warmup 2 --cpu-sample 0 2>/dev/null | python scripts/benchline.py
grep -E "^\[(setup|anchors|resident|extend|lcb|filter|chain|run_batch)" $O/r4_laps.log | tail -40
timeout 500 python -m pytest tests/test_gpu_big.py -m gpu -x -q -k "baseline_size" 2>&1 | tail -15 > $O/r4_t3.log; tail -3 $O/r4_t3.log
