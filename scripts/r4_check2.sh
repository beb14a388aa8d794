#!/bin/bash
# round 4: bench + laps + kernel trace of the resident route:  gpurun --timeout 900 -- 'bash scripts/r4_check2.sh'
mkdir -p gpurun_out
O=gpurun_out
timeout 300 python bench.py --steps 40 --warmup 5 --cpu-sample 0 > $O/r4_bench40.json 2> $O/r4_bench40.err; tail -1 $O/r4_bench40.json | python scripts/benchline.py
PARSNP_BENCH_LOG=$O/r4_laps.log PARSNP_DEBUG_TIMERS=1 timeout 300 python bench.py --steps 3 --warmup 2 --cpu-sample 0 2>/dev/null | python scripts/benchline.py
grep -E "^\[(setup|anchors|resident|extend|lcb|filter|chain|run_batch)" $O/r4_laps.log | tail -24
export TMPDIR=/tmp; D=$PWD
rm -rf $O/r4_prof; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $D/$O/r4_prof -o r4 -- python $D/bench.py --steps 8 --warmup 2 --cpu-sample 0 > $D/$O/r4_prof_bench.json 2> $D/$O/r4_prof.err)
f=$(find $O/r4_prof -name "*kernel_stats.csv" | head -1)
cp "$f" $O/r4_kernel_stats.csv; head -45 $O/r4_kernel_stats.csv | cut -c1-160
rm -rf $O/r4_prof
