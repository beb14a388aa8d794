#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "small_regions or batched or random_regions or test_events" 2>&1 | tail -3
timeout 300 python bench.py --steps 60 --warmup 5 --cpu-sample 0 > $O/r4_bench_grp1.json 2> $O/r4_bench_grp1.err; tail -1 $O/r4_bench_grp1.json | python scripts/benchline.py
timeout 300 python bench.py --steps 60 --warmup 5 --cpu-sample 0 --tune group_small=0 > $O/r4_bench_grp0.json 2> $O/r4_bench_grp0.err; tail -1 $O/r4_bench_grp0.json | python scripts/benchline.py
