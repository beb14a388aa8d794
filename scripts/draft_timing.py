#!/usr/bin/env python3
"""Time parsnp_core on a synthetic draft-assembly set (contigs joined by N runs): python scripts/draft_timing.py n genomes contigs"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parsnp_amd import driver, synth  # noqa: E402
from parsnp_amd.paths import CORE_BIN  # noqa: E402

n, ng, c = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
base = tempfile.mkdtemp(prefix="draft_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
rp, qs = synth.draft_set(os.path.join(base, "in"), n=n, n_genomes=ng, contigs=c)
out = os.path.join(base, "out")
t = time.time()
rc, _ = driver.run_core(os.path.abspath(CORE_BIN), rp, qs, out, timing=os.path.join(base, "timing.json"), threads=16)
print("rc", rc, "wall %.2fs" % (time.time() - t))
if rc == 0:
    tj = json.load(open(os.path.join(base, "timing.json")))
    print(json.dumps({k: tj[k] for k in tj if k.endswith("_s") or k in ("finder_calls", "regions_processed", "anchors", "mums", "lcbs")}))
else:
    print(open(os.path.join(out, "parsnp-aligner.err")).read()[-1500:])
