#!/usr/bin/env python3
"""Host-side phase timers of one parsnp_core run on a CONFIGS workload: python scripts/host_timers.py [workload] [threads]"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parsnp_amd import driver, synth  # noqa: E402
from parsnp_amd.paths import CORE_BIN  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "bact200"
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
base = tempfile.mkdtemp(prefix="timers_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
r, gs = synth.make(name)
rp, qs = synth.write_set(os.path.join(base, "in"), r, gs)
out = os.path.join(base, "out")
env = dict(os.environ, PARSNP_DEBUG_TIMERS="1")
t = time.time()
rc, _ = driver.run_core(os.path.abspath(CORE_BIN), rp, qs, out, timing=os.path.join(base, "timing.json"), env=env, threads=threads)
print("rc", rc, "wall %.2fs" % (time.time() - t))
for line in open(os.path.join(out, "parsnp-aligner.err")):
    if line.startswith("["):
        print(line.rstrip())
print(json.dumps(json.load(open(os.path.join(base, "timing.json")))))
