timeout 240 python -m pytest tests/test_gpu_gapalign.py -x -q 2>&1 | tail -3
timeout 300 python -m pytest tests/test_gpu_big.py -x -q -k "bact200" 2>&1 | tail -3
PM_GAP_DEBUG=3 PARSNP_BENCH_LOG=gpurun_out/bench_laps.log PARSNP_DEBUG_TIMERS=1 timeout 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null > gpurun_out/bench_out.json
grep -E "^\[gap" gpurun_out/bench_laps.log | tail -22
