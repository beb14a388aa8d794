#!/usr/bin/env python3
"""Same XMFA from the generation-parallel and the in-order replay, run to run: python scripts/replay_consistency.py [workload] [genomes] [threads]"""
import hashlib
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parsnp_amd import driver, synth  # noqa: E402
from parsnp_amd.paths import CORE_BIN  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "bact200"
model, kw = synth.CONFIGS[name]
kw = dict(kw)
if len(sys.argv) > 2:
    kw["n_genomes"] = int(sys.argv[2])
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 24
base = tempfile.mkdtemp(prefix="replay_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
r, gs = {"population": synth.population, "musclefree": synth.musclefree, "rearranged": synth.rearranged, "pop_rearranged": synth.pop_rearranged}[model](**kw)
rp, qs = synth.write_set(os.path.join(base, "in"), r, gs)
sums = []
for tag, extra in (("generations", {}), ("generations", {}), ("generations", {}), ("in_order", {"PARSNP_SEQUENTIAL_REPLAY": "1"})):
    out = os.path.join(base, "out")
    env = dict(os.environ, PARSNP_DEBUG_TIMERS="1", **extra)
    rc, _ = driver.run_core(os.path.abspath(CORE_BIN) + ("_hooks" if extra else ""), rp, qs, out, env=env, threads=threads)
    h = hashlib.md5()
    with open(os.path.join(out, "parsnpAligner.xmfa"), "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    note = [l.strip() for l in open(os.path.join(out, "parsnp-aligner.err")) if l.startswith("[extend]")]
    sums.append(h.hexdigest())
    print(tag, rc, h.hexdigest(), note)
print("identical" if len(set(sums)) == 1 else "DIFFERENT")
