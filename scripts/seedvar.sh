#!/bin/bash
# The event search with other compile-time choices (here: the longest seed, PM_KMAX): step time, kernel time per step and of the
# anchor launch, and the result counts.  Measurement helper: the variant libraries (parsnp_amd/lib/exp/libparsnp_hip_<v>.so, built
# by hand with -DPM_KMAX=<k>) are substituted with LD_PRELOAD, the shipped binary is not touched.
for v in "" k14 k12 "" k14; do
  pre=""; [ -n "$v" ] && pre="$(pwd)/parsnp_amd/lib/exp/libparsnp_hip_$v.so"
  LD_PRELOAD=$pre python bench.py --steps 30 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
e=d['engine_ms']
print('variant=${v:-shipped}', 'ms_per_step', d['ms_per_step'], 'seed_extend/step', e['seed_extend'], 'anchor launch', d['anchor_launch_ms']['seed_extend'], 'index', e['index'], 'repeat', e['repeat'], 'sort', e['sort'], 'events', e['events'], 'rest', e['rest_samples'], 'anchors', d['anchors'], 'mums', d['mums'], 'lcbs', d['lcbs'])"
done
