for t in 8 16 32 4 8 16; do timeout 300 python bench.py --steps 100 --warmup 5 --cpu-sample 0 --other-configs off --tune filter_factor=$t 2>/dev/null | tail -1 | python -c "
import json,sys,statistics
d=json.loads(sys.stdin.read()); e=d['engine_ms']; print('filter_factor=$t', d['ms_per_step'], 'median', statistics.median(d['step_ms']), 'index', e['index'], 'repeat', e['repeat'], 'seed', e['seed_extend'], 'rest', e['rest_samples'])"; done
