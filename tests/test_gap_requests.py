"""Requests derived on the device (include/parsnp_mum.h): the recursion's seed regions are worked out by the engine itself from
its resident anchor table and searched beside the host's validation of the anchors (pm_multi_mum_batch_spec, a helper thread),
or -- with that switched off -- sent as references into the table, 16 bytes per region instead of 16 bytes per region and
genome (pm_multi_mum_batch_gaps).  On the CPU the engine's functors run sequentially (tests/emu); the runs must use the
paths (spec_hits / gap_requests > 0), give the bytes of the run in which every row travels after the validation, and all
must be the reference binary's golden."""
import json
import os
import subprocess
import sys

import pytest

import xmfa_util
from parsnp_amd import driver, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E2E = json.load(open(os.path.join(ROOT, "tests", "golden", "e2e.json")))

_CHILD = """
import sys, json
sys.path.insert(0, %r)
from parsnp_amd.core_api import CoreRun
r = CoreRun(sys.argv[1], sys.argv[2])
rep = r.step(); r.write(); r.close()
print(json.dumps({k: rep[k] for k in ("gap_requests", "spec_regions", "spec_hits", "finder_calls", "finder_regions", "anchors", "mums", "lcbs")}))
""" % ROOT


@pytest.mark.parametrize("name", ["pop6x200k", "rearr6x300k"])
def test_gap_requests_same_bytes(emu, tmp_path, name):
    core_lib = os.path.join(os.path.dirname(emu[0]), "libparsnp_core_emu.so")
    ref, gs = synth.make(name)
    rp, qs = synth.write_set(str(tmp_path / "in"), ref, gs)
    got = {}
    for tag, env in (("spec", {}), ("gaps", {"PARSNP_NO_SPECULATIVE_SEEDS": "1"}), ("rows", {"PARSNP_NO_SPECULATIVE_SEEDS": "1", "PARSNP_NO_GAP_REQUESTS": "1"})):
        out = str(tmp_path / tag)
        os.makedirs(out)
        ini = os.path.join(out, "run.ini")
        open(ini, "w").write(driver.ini_text(rp, qs, out, threads=4))
        # (small sets: the thresholds of the long-list routes are lowered so that the anchor list takes them)
        e = dict(os.environ, PM_DIRTY_MIN="16", PARSNP_PARALLEL_MIN="16", **env)
        p = subprocess.run([sys.executable, "-c", _CHILD, ini, core_lib], capture_output=True, text=True, env=e, cwd=out, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        got[tag] = (json.loads(p.stdout.strip().splitlines()[-1]), xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")),
                    xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")))
    assert got["rows"][0]["gap_requests"] == 0 and got["rows"][0]["spec_regions"] == 0 and got["gaps"][0]["spec_regions"] == 0
    if name == "pop6x200k":      # collinear: the seeds lie between anchors that follow each other in every genome
        assert got["gaps"][0]["gap_requests"] > 100
        assert got["spec"][0]["spec_hits"] > 100 and got["spec"][0]["spec_hits"] == got["spec"][0]["spec_regions"]
    else:                        # rearranged: the host walks its bitmaps; what the engine computed ahead is not asked for
        assert got["spec"][0]["spec_regions"] > 0
    assert got["spec"][1] == got["gaps"][1] == got["rows"][1] == E2E[name]["xmfa_md5"]
    assert got["spec"][2] == got["gaps"][2] == got["rows"][2] == E2E[name]["log"]
