run() { echo "== $1"; env $2 PARSNP_BENCH_LOG=gpurun_out/bench_laps_$1.log PARSNP_DEBUG_TIMERS=1 timeout 200 python bench.py --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null | python scripts/benchline.py; grep -E "^\[(validate_parallel\] mark|generation 0\] sort)" gpurun_out/bench_laps_$1.log | tail -2; }
run markfirst PARSNP_MARK_FIRST=1
run nice10 X=1
run nice0 PARSNP_MARK_NICE=0
run nice19 PARSNP_MARK_NICE=19
run nice10b X=1
