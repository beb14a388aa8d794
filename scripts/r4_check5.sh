#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
timeout 300 python bench.py --steps 60 --warmup 5 --cpu-sample 0 > $O/r4_bench60.json 2> $O/r4_bench60.err; tail -1 $O/r4_bench60.json | python scripts/benchline.py
timeout 300 python bench.py --steps 60 --warmup 5 --cpu-sample 0 --host-threads 4 2>/dev/null | python scripts/benchline.py | head -2
timeout 600 python -m pytest tests/test_fuzz_vs_reference.py tests/test_gpu_parity.py -m gpu -x -q -k "resident" 2>&1 | tail -3
PARSNP_BENCH_LOG=$O/r4_laps.log PARSNP_DEBUG_TIMERS=1 timeout 300 python bench.py --steps 3 --warmup 2 --cpu-sample 0 2>/dev/null > /dev/null
grep -E "^\[(anchors|resident|extend|lcb|filter|chain|run_batch)" $O/r4_laps.log | tail -26
bash scripts/profile_stats.sh > $O/r4_stats.log 2>&1; tail -3 $O/r4_stats.log | cut -c1-300
