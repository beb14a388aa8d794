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
print(json.dumps({k: rep[k] for k in ("gap_requests", "layout_images", "spec_regions", "spec_hits", "finder_calls", "finder_regions", "anchors", "mums", "lcbs")}))
""" % ROOT

# the same with the step run three times before the output is written: the runs after the first start on the bitmaps the
# previous one left all zero (or, without an image, on bitmaps cleared again)
_CHILD_REPEAT = _CHILD.replace("rep = r.step(); r.write()", "r.step(); r.step(); rep = r.step(); r.write()")


@pytest.mark.parametrize("name", ["pop6x200k", "rearr6x300k", "pop12x400k"])
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
        e = dict(os.environ, PM_DIRTY_MIN="16", PARSNP_PARALLEL_MIN="16", PARSNP_NO_RESIDENT="1", **env)      # (routes of the HOST route)
        p = subprocess.run([sys.executable, "-c", _CHILD, ini, core_lib], capture_output=True, text=True, env=e, cwd=out, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        got[tag] = (json.loads(p.stdout.strip().splitlines()[-1]), xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")),
                    xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")))
    assert got["rows"][0]["gap_requests"] == 0 and got["rows"][0]["spec_regions"] == 0 and got["gaps"][0]["spec_regions"] == 0
    if name != "rearr6x300k":    # collinear: the seeds lie between anchors that follow each other in every genome
        assert got["gaps"][0]["gap_requests"] > 100
        # (the engine also guesses the regions next to rows the host may still refuse or trim: a few of those are never asked for)
        assert got["spec"][0]["spec_hits"] > 100 and 0 <= got["spec"][0]["spec_regions"] - got["spec"][0]["spec_hits"] <= 32
    else:                        # rearranged: the engine's order flags say so, and nothing is computed ahead (the host walks its bitmaps)
        assert got["spec"][0]["spec_regions"] == 0
    assert got["spec"][1] == got["gaps"][1] == got["rows"][1] == E2E[name]["xmfa_md5"]
    assert got["spec"][2] == got["gaps"][2] == got["rows"][2] == E2E[name]["log"]


# The layout after the anchor call as an image built on the device (include/parsnp_mum.h: pm_layout_image) against the host
# marking its own bitmaps (PARSNP_HOST_MARKS=1): same bytes, also when the step is repeated in one process (the bitmaps of
# the previous run are reused: all zero after an image run, cleared after a host run).
@pytest.mark.parametrize("name", ["pop6x200k", "rearr6x300k", "pop12x400k"])      # pop12x400k: 166 flagged candidates, 6 of them tangled
def test_layout_image_same_bytes(emu, tmp_path, name):
    core_lib = os.path.join(os.path.dirname(emu[0]), "libparsnp_core_emu.so")
    ref, gs = synth.make(name)
    rp, qs = synth.write_set(str(tmp_path / "in"), ref, gs)
    got = {}
    # image: asked for before the validation, the engine choosing the rows, the host putting right what it decides otherwise;
    # late: asked for after it (PARSNP_LATE_IMAGE=1: the route of a list whose overlap flags the host works out itself)
    for tag, env, child in (("image", {}, _CHILD), ("late", {"PARSNP_LATE_IMAGE": "1"}, _CHILD), ("host", {"PARSNP_HOST_MARKS": "1"}, _CHILD),
                            ("image3", {}, _CHILD_REPEAT), ("late3", {"PARSNP_LATE_IMAGE": "1"}, _CHILD_REPEAT), ("host3", {"PARSNP_HOST_MARKS": "1"}, _CHILD_REPEAT)):
        out = str(tmp_path / tag)
        os.makedirs(out)
        ini = os.path.join(out, "run.ini")
        open(ini, "w").write(driver.ini_text(rp, qs, out, threads=4))
        e = dict(os.environ, PM_DIRTY_MIN="16", PARSNP_PARALLEL_MIN="16", PARSNP_CHECK_ZERO="1", PARSNP_NO_RESIDENT="1", **env)
        p = subprocess.run([sys.executable, "-c", child, ini, core_lib], capture_output=True, text=True, env=e, cwd=out, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        got[tag] = (json.loads(p.stdout.strip().splitlines()[-1]), xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")),
                    xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")))
    assert got["host"][0]["layout_images"] == 0 and got["host3"][0]["layout_images"] == 0
    if name != "rearr6x300k":    # collinear: the accepted anchors lie in list order, the marks are put off -- and come as an image
        assert got["image"][0]["layout_images"] == 1 and got["image3"][0]["layout_images"] == 1
        assert got["late"][0]["layout_images"] == 1 and got["late3"][0]["layout_images"] == 1
    for tag in got:
        assert got[tag][1] == E2E[name]["xmfa_md5"], tag
        assert got[tag][2] == E2E[name]["log"], tag
