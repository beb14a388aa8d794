#!/bin/bash
# SeedExtend variants (PM_DEBUG_SEED bits 8 and 2048 keep the results): kernel time per step and of the anchor launch
for d in "$@"; do
  PM_DEBUG_SEED=$d python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant $d', d['value'], 'seed_extend/step', d['engine_ms']['seed_extend'], 'anchor', d['anchor_launch_ms']['seed_extend'], 'mums', d['mums'], 'lcbs', d['lcbs'])"
done
