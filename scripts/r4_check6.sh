#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
timeout 300 python bench.py --steps 60 --warmup 5 --cpu-sample 0 > $O/r4_bench60.json 2> $O/r4_bench60.err; tail -1 $O/r4_bench60.json | python scripts/benchline.py
timeout 600 python -m pytest tests/test_fuzz_vs_reference.py tests/test_gpu_parity.py -m gpu -x -q -k "resident or synthetic or test_events or random_regions" 2>&1 | tail -3
bash scripts/profile_stats.sh > $O/r4_stats.log 2>&1; tail -3 $O/r4_stats.log | cut -c1-400
