#!/bin/bash
# round 5: the profiler passes on the build at hand (kernel trace + stats, FETCH_SIZE / WRITE_SIZE passes, the HBM and VALU
# calibrations, SQ counters), the summaries copied into profiles/r05/ so that the default bench line that follows can quote the
# traffic and the issue fraction OF THIS BUILD; then the per-step floor (bench.py --genomes 25 ... 200).
#   gpurun --timeout 1500 -- 'bash scripts/round_profile.sh'
mkdir -p gpurun_out/final profiles/r05
O=gpurun_out/final
bash scripts/profile.sh > $O/profile.log 2>&1; ls gpurun_out/prof/summary
for f in traffic_seed_extend.json calibration.json kernel_stats.csv pmc_per_kernel.json idle_gaps.json calib_bytes.json valu_calib.json bench_plain.json bench_under_rocprof.json; do cp gpurun_out/prof/summary/$f profiles/r05/ 2>/dev/null; done
bash scripts/sqcounters.sh > $O/sq.log 2>&1; tail -4 $O/sq.log | cut -c1-200
cp gpurun_out/sq/summary.json profiles/r05/sq_seed_extend.json
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json | python scripts/benchline.py | head -2
cp $O/bench_default.json profiles/r05/bench_default.json
timeout 300 python bench.py --steps 60 --warmup 5 --cpu-sample 0 --other-configs off --tune master_seg=0 > $O/bench_master_seg_0.json 2> /dev/null; tail -1 $O/bench_master_seg_0.json | python scripts/benchline.py | head -1
cp $O/bench_master_seg_0.json profiles/r05/
bash scripts/floor.sh; cp gpurun_out/floor.json profiles/r05/floor.json
mkdir -p gpurun_out/profiles_r05; cp profiles/r05/* gpurun_out/profiles_r05/
# config 5 on the host route, with the host's laps (DESIGN.md section 6 quotes them)
PARSNP_DEBUG_TIMERS=1 timeout 400 python bench.py --workload rearr500 --steps 2 --warmup 1 --cpu-sample 0 --other-configs off > $O/rearr500_laps.json 2> $O/rearr500_laps.err
grep -E "^\[(anchors|extend|lcb|replay|sweep|generation 1\]|filter)" $O/rearr500_laps.err | tail -40 > gpurun_out/profiles_r05/rearr500_laps.txt; cp gpurun_out/profiles_r05/rearr500_laps.txt profiles/r05/
# the product's sources with the hooks compiled in, side by side with the reference binary, final kernels
PARSNP_FUZZ_CORE=hip timeout 600 python scripts/fuzz_campaign.py 6500 6580 4 > gpurun_out/profiles_r05/fuzz_hip_final.log 2>&1; tail -1 gpurun_out/profiles_r05/fuzz_hip_final.log
PARSNP_BENCH_LOG=$O/order.log timeout 200 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --other-configs off --tune order_debug=1 > /dev/null 2> $O/order.err; grep "order check" $O/order.log | tail -1 | tee gpurun_out/profiles_r05/order_check_counts.txt
