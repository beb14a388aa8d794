#!/bin/bash
# the whole GPU suite + a short bench line:  gpurun --timeout 1500 -- 'bash scripts/gpu_suite.sh'
mkdir -p gpurun_out/r5
timeout 120 python scripts/smoke_core.py 2>&1 | tail -2
timeout 1300 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 300 python bench.py --steps 60 --warmup 5 --cpu-sample 0 --other-configs off > gpurun_out/r5/bench_suite.json 2> gpurun_out/r5/bench_suite.err; tail -1 gpurun_out/r5/bench_suite.json | python scripts/benchline.py
