for v in "parsnp_amd/lib/exp/libparsnp_hip_lead2.so" "parsnp_amd/lib/exp/libparsnp_hip_lead4.so" "" "parsnp_amd/lib/exp/libparsnp_hip_lead2.so" "parsnp_amd/lib/exp/libparsnp_hip_lead4.so" ""; do LD_PRELOAD=$v timeout 300 python bench.py --steps 150 --warmup 5 --cpu-sample 0 --other-configs off 2>/dev/null | tail -1 | python -c "
import json,sys,statistics
d=json.loads(sys.stdin.read()); e=d['engine_ms']; print('lib [$v]', d['ms_per_step'], 'median', statistics.median(d['step_ms']), 'seed', e['seed_extend'], 'rest_samples', e['rest_samples'], 'events', e['events'])"; done
