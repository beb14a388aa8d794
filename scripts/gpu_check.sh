#!/bin/bash
# GPU test suite + default bench in one gpurun call:  gpurun --timeout 900 -- 'bash scripts/gpu_check.sh'
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.json
