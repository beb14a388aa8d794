#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_vs_reference.py -m gpu -x -q -k "small_regions or batched or random_regions or test_events or mumi or medium or resident or synthetic or golden or properties" 2>&1 | tail -3
timeout 300 python bench.py --steps 60 --warmup 5 --cpu-sample 0 --other-configs off > $O/r4_bench_c10.json 2> $O/r4_bench_c10.err; tail -1 $O/r4_bench_c10.json | python scripts/benchline.py
bash scripts/profile_stats.sh > $O/r4_stats.log 2>&1; grep -E "FoldCandidates|WaveS|SeedWalk|MasterEP|Grouped|SmallPair|ClusterVal|SettleClean|JudgePairs" $O/prof_stats/summary/kernel_stats.csv | cut -c1-200
