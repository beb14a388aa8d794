#!/bin/bash
# rounds 5-6 (ROUND=r06 by default): the profiler passes on the build at hand (kernel trace + stats, FETCH_SIZE / WRITE_SIZE passes, the HBM and VALU
# calibrations, SQ counters), the summaries copied into profiles/$R/ so that the default bench line that follows can quote the
# traffic and the issue fraction OF THIS BUILD; then the per-step floor (bench.py --genomes 25 ... 200).
#   gpurun --timeout 1500 -- 'bash scripts/round_profile.sh'
R=${ROUND:-r06}
mkdir -p gpurun_out/final profiles/$R
O=gpurun_out/final
bash scripts/profile.sh > $O/profile.log 2>&1; ls gpurun_out/prof/summary
for f in traffic_seed_extend.json traffic_phases.json calibration.json kernel_stats.csv pmc_per_kernel.json idle_gaps.json calib_bytes.json valu_calib.json bench_plain.json bench_under_rocprof.json; do cp gpurun_out/prof/summary/$f profiles/$R/ 2>/dev/null; done
bash scripts/sqcounters.sh > $O/sq.log 2>&1; tail -4 $O/sq.log | cut -c1-200
cp gpurun_out/sq/summary.json profiles/$R/sq_seed_extend.json
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json | python scripts/benchline.py | head -2
cp $O/bench_default.json profiles/$R/bench_default.json
timeout 300 python bench.py --steps 60 --warmup 5 --cpu-sample 0 --other-configs off --tune master_seg=0 > $O/bench_master_seg_0.json 2> /dev/null; tail -1 $O/bench_master_seg_0.json | python scripts/benchline.py | head -1
cp $O/bench_master_seg_0.json profiles/$R/
bash scripts/floor.sh; cp gpurun_out/floor.json profiles/$R/floor.json
mkdir -p gpurun_out/profiles_$R; cp profiles/$R/* gpurun_out/profiles_$R/
# the bucket order against the radix sort (tune bucket_sort = 0), the device's timeline of one step, config 3 with inversions and config 5 with the host's laps
timeout 300 python bench.py --steps 60 --warmup 5 --cpu-sample 0 --other-configs off --tune bucket_sort=0 > $O/bench_bucket_sort_0.json 2> /dev/null; tail -1 $O/bench_bucket_sort_0.json | python scripts/benchline.py | head -1
cp $O/bench_bucket_sort_0.json profiles/$R/
P=$GRAFT_REPO_ROOT/$O/tl; rm -rf $P; mkdir -p $P
( cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace -d $P -o ks -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --cpu-sample 0 --other-configs off > $P/bench.json 2> $P/err.log )
python scripts/step_timeline.py $P profiles/$R/timeline_bact200.txt | head -12; rm -rf $P
bash scripts/r6_step.sh inv rearr prof
cp gpurun_out/r6/inv_laps.txt gpurun_out/r6/rearr_laps.txt gpurun_out/r6/kernels_bact200inv.txt gpurun_out/r6/kernels_rearr500.txt gpurun_out/r6/bench_inv.json gpurun_out/r6/bench_rearr500.json profiles/$R/ 2>/dev/null
# the product's sources with the hooks compiled in, side by side with the reference binary, final kernels
PARSNP_FUZZ_CORE=hip timeout 900 python scripts/fuzz_campaign.py 10000 10120 6 > gpurun_out/profiles_$R/fuzz_hip_final.log 2>&1; tail -2 gpurun_out/profiles_$R/fuzz_hip_final.log
PARSNP_FUZZ_CORE=hip PM_FLAGGED_DIV=1 timeout 900 python scripts/fuzz_campaign.py 11000 11100 4 > gpurun_out/profiles_$R/fuzz_hip_flagged_div_1.log 2>&1; tail -2 gpurun_out/profiles_$R/fuzz_hip_flagged_div_1.log
cp gpurun_out/profiles_$R/fuzz_hip_final.log gpurun_out/profiles_$R/fuzz_hip_flagged_div_1.log profiles/$R/
mkdir -p gpurun_out/profiles_$R; cp -r profiles/$R/* gpurun_out/profiles_$R/
