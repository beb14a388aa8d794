#!/bin/bash
# SeedExtend with other compile-time choices (make -C parsnp_amd/csrc exp: samples per lane, wavefronts per SIMD, leader spacing): kernel time per step and of the anchor launch.
# Measurement helper: the variant libraries are substituted with LD_PRELOAD, the shipped binary is not touched.
for v in "" per1 per2w7 per2w8 lead16; do
  pre=""; [ -n "$v" ] && pre="$(pwd)/parsnp_amd/lib/exp/libparsnp_hip_$v.so"
  LD_PRELOAD=$pre python bench.py --steps 10 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('variant=${v:-shipped}', 'ms_per_step', d['ms_per_step'], 'seed_extend/step', d['engine_ms']['seed_extend'], 'anchor launch', d['anchor_launch_ms']['seed_extend'], 'anchors', d['anchors'], 'mums', d['mums'])"
done
