#!/usr/bin/env python3
"""Run parsnp_core once on a synth.CONFIGS workload and say whether it stayed on the resident route and, if not, why:
python scripts/why_route.py rearr50 [threads]"""
import json, os, shutil, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parsnp_amd import driver, synth
from parsnp_amd.paths import CORE_BIN
name = sys.argv[1]
base = "/dev/shm" if os.path.isdir("/dev/shm") else None
d = tempfile.mkdtemp(prefix="why_", dir=base)
try:
    ref, gs = synth.make(name)
    rp, qs = synth.write_set(os.path.join(d, "in"), ref, gs)
    env = dict(os.environ, OMP_WAIT_POLICY="passive", PARSNP_DEBUG_TIMERS="1")
    rc, _ = driver.run_core(CORE_BIN, rp, qs, os.path.join(d, "out"), env=env, threads=int(sys.argv[2]) if len(sys.argv) > 2 else 24, timing=os.path.join(d, "t.json"))
    tj = json.load(open(os.path.join(d, "t.json"))) if os.path.exists(os.path.join(d, "t.json")) else {}
    print("rc", rc, {k: tj.get(k) for k in ("resident", "resident_why", "resident_retry", "device_chain", "outside_writes", "anchors", "mums", "lcbs", "total_s", "extend_s")})
    err = open(os.path.join(d, "out", "parsnp-aligner.err")).read().splitlines()
    for l in err:
        if l.startswith("[resident") or "route" in l:
            print(l[:300])
finally:
    shutil.rmtree(d, ignore_errors=True)
