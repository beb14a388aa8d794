#!/bin/bash
# scripts/profile_stats.sh -- only pass 1 of scripts/profile.sh (kernel trace + stats of the bench command), for a quick
# refresh of profiles/<round>/kernel_stats.csv after a kernel change:  gpurun --timeout 300 -- 'bash scripts/profile_stats.sh'
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_stats
rm -rf $OUT; mkdir -p $OUT/summary
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o ks -- python $REPO/bench.py --steps 3 --warmup 1 --cpu-sample 0 --other-configs off > $OUT/summary/bench_under_rocprof.json 2> $OUT/stats.err
cd $REPO
python scripts/profile_summary.py $OUT > /dev/null 2>&1
rm -rf $OUT/stats
head -12 $OUT/summary/kernel_stats.csv; tail -1 $OUT/summary/bench_under_rocprof.json | python scripts/benchline.py
