#!/bin/bash
# the rocprofv3 passes on the final build, then the default bench line WITH the stamped traffic file in place
mkdir -p gpurun_out/final
O=gpurun_out/final
bash scripts/profile.sh > $O/profile.log 2>&1; ls gpurun_out/prof/summary
cp gpurun_out/prof/summary/traffic_seed_extend.json gpurun_out/prof/summary/calibration.json profiles/r04/
bash scripts/sqcounters.sh > $O/sq.log 2>&1; tail -4 $O/sq.log | cut -c1-200
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json | python scripts/benchline.py | head -2
timeout 300 python bench.py --steps 60 --warmup 5 --cpu-sample 0 --other-configs off --tune group_small=0 > $O/bench_group_small_0.json 2> /dev/null; tail -1 $O/bench_group_small_0.json | python scripts/benchline.py | head -1
