#!/bin/bash
# scripts/profile.sh -- the rocprofv3 evidence behind bench.py's roofline, run on the GPU box from the repo root:
#   1. kernel trace + stats of the default bench command            -> gpurun_out/prof/stats
#   2. PMC pass FETCH_SIZE, 3. PMC pass WRITE_SIZE (separate passes) -> gpurun_out/prof/{fetch,write}
#   4. FETCH_SIZE / WRITE_SIZE calibration on known byte counts      -> gpurun_out/prof/calib   (scripts/hbm_calib.hip)
# then scripts/profile_summary.py condenses them into gpurun_out/prof/summary/ (copied by hand into profiles/rNN/).
# Every bench run is wrapped in `timeout`: a hung profiler run must not eat the GPU budget.
set -x
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT/summary
export TMPDIR=/tmp
cd /tmp
timeout 200 python $REPO/bench.py --steps 20 --warmup 3 --other-configs off > $OUT/summary/bench_plain.json 2> $OUT/bench_plain.err
timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/stats -o ks -- python $REPO/bench.py --steps 3 --warmup 1 --cpu-sample 0 --other-configs off > $OUT/summary/bench_under_rocprof.json 2> $OUT/stats.err
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o pf -- python $REPO/bench.py --steps 1 --warmup 0 --cpu-sample 0 --other-configs off > $OUT/fetch.out 2> $OUT/fetch.err
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o pw -- python $REPO/bench.py --steps 1 --warmup 0 --cpu-sample 0 --other-configs off > $OUT/write.out 2> $OUT/write.err
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/calib -o cf -- $REPO/parsnp_amd/bin/hbm_calib > $OUT/summary/calib_bytes.json 2> $OUT/calib.err
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/calibw -o cw -- $REPO/parsnp_amd/bin/hbm_calib > /dev/null 2> $OUT/calibw.err
timeout 120 $REPO/parsnp_amd/bin/valu_calib > $OUT/summary/valu_calib.json 2> $OUT/valu.err
cd $REPO
find $OUT -name "*.csv" | head -40
python scripts/profile_summary.py $OUT
ls -la $OUT/summary
