#!/bin/bash
# GPU test suite + default bench in one gpurun call:  gpurun --timeout 900 -- 'bash scripts/gpu_check.sh'
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.json | python scripts/benchline.py
PARSNP_BENCH_LOG=gpurun_out/bench_laps.log PARSNP_DEBUG_TIMERS=1 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --other-configs off 2>/dev/null | python scripts/benchline.py
grep -E "^\[(setup|validate|generation|lcb)" gpurun_out/bench_laps.log | tail -24
