for v in 0 1; do
if [ $v = 1 ]; then export PARSNP_MARK_FIRST=1; fi
PARSNP_BENCH_LOG=gpurun_out/bench_laps_m$v.log PARSNP_DEBUG_TIMERS=1 timeout 200 python bench.py --steps 10 --warmup 2 --cpu-sample 0 2>/dev/null | python scripts/benchline.py
grep -E "^\[(validate_parallel|anchors|generation)" gpurun_out/bench_laps_m$v.log | tail -22
done
