#!/bin/bash
# gprof of the host side on bact200 (profiling build parsnp_amd/bin/parsnp_core_pg, see DESIGN notes)
python - <<'PY'
import os, sys, subprocess, tempfile
sys.path.insert(0, os.getcwd())
from parsnp_amd import synth, driver
base = tempfile.mkdtemp(dir='/dev/shm')
r, gs = synth.make('bact200')
rp, qs = synth.write_set(os.path.join(base, 'in'), r, gs)
out = os.path.join(base, 'out')
rc, ini = driver.run_core(os.path.abspath('parsnp_amd/bin/parsnp_core_pg'), rp, qs, out, threads=24)
print('rc', rc)
p = subprocess.run(['gprof', '-b', '-p', os.path.abspath('parsnp_amd/bin/parsnp_core_pg'), os.path.join(out, 'gmon.out')], capture_output=True, text=True)
print(p.stdout[:6000]); print(p.stderr[:500])
PY
