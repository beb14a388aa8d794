timeout 240 python -m pytest tests/test_gpu_gapalign.py -x -q 2>&1 | tail -2
PARSNP_BENCH_LOG=gpurun_out/bench_laps0.log PARSNP_DEBUG_TIMERS=1 timeout 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null > gpurun_out/bench_out0.json
grep -E "^\[gap batch\] kernel" gpurun_out/bench_laps0.log
PM_GAP_DEBUG=3 PARSNP_BENCH_LOG=gpurun_out/bench_laps.log PARSNP_DEBUG_TIMERS=1 timeout 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null > gpurun_out/bench_out.json
grep -E "^\[(output|gap)" gpurun_out/bench_laps.log | tail -40
python -c "
import json; d=json.loads(open('gpurun_out/bench_out0.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['split_s']['output'], d['cold'])"
