for t in 4 8 12; do
PM_STAGE_THREADS=$t PARSNP_BENCH_LOG=gpurun_out/bench_laps_$t.log PARSNP_DEBUG_TIMERS=1 timeout 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null > gpurun_out/bench_out_$t.json
grep -E "^\[upload\] stage" gpurun_out/bench_laps_$t.log
python -c "
import json; d=json.loads(open('gpurun_out/bench_out_$t.json').read().strip().splitlines()[-1]); print($t, d['ms_per_step'], 'ingest %.3f upload %.3f output %.3f' % (d['split_s']['ingest'], d['split_s']['upload'], d['split_s']['output']), d['cold'])"
done
