#!/bin/bash
# the whole GPU suite with its summary line kept, and the default bench line once more (another box: the spread between boxes)
mkdir -p gpurun_out/final gpurun_out/profiles_r05
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/final/suite_full.log 2>&1; grep -E "passed|failed|error" gpurun_out/final/suite_full.log | tail -3 | tee gpurun_out/profiles_r05/gpu_suite.txt
timeout 600 python bench.py > gpurun_out/profiles_r05/bench_default_rerun.json 2> gpurun_out/final/bench_rerun.err; tail -1 gpurun_out/profiles_r05/bench_default_rerun.json | python scripts/benchline.py | head -2
