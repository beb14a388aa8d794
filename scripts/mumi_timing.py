#!/usr/bin/env python3
"""Time parsnp_core in calcmumi mode on a CONFIGS workload: python scripts/mumi_timing.py [workload]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parsnp_amd import driver, synth  # noqa: E402
from parsnp_amd.paths import CORE_BIN  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "bact200"
base = tempfile.mkdtemp(prefix="mumi_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
r, gs = synth.make(name)
rp, qs = synth.write_set(os.path.join(base, "in"), r, gs)
out = os.path.join(base, "out")
t = time.time()
rc, _ = driver.run_core(os.path.abspath(CORE_BIN), rp, qs, out, calcmumi=1, threads=24, env=dict(os.environ, PARSNP_DEBUG_TIMERS="1"))
print("rc", rc, "wall %.2fs" % (time.time() - t))
print(open(os.path.join(out, "parsnp-aligner.err")).read()[-800:])
print(open(os.path.join(out, "all.mumi")).read()[:120])
