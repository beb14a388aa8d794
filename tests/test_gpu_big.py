"""BASELINE-size parity on a real MI355X: parsnp_core (24 host threads, both replay modes) against the REFERENCE binary's
XMFA md5, MUM/LCB signature and log counters at the sizes BASELINE.json quotes -- config 3 (200 x 5 Mb, --no-partition),
one 250-genome partition of config 4 (2000 x 5 Mb, Random(42) order) and config 5 cut to its first 50 genomes
(5 % segregating sites, 10 % of every genome rearranged).  The goldens (tests/golden/e2e_big.json) were produced in the
build container by tests/golden/make_golden_big.py from oracle/_ref/parsnp_core_ref; the inputs are regenerated here
from the same seeds (parsnp_amd.synth)."""
import json
import os
import shutil
import tempfile

import pytest

import xmfa_util
from parsnp_amd import driver, synth
from parsnp_amd.paths import CORE_BIN

pytestmark = pytest.mark.gpu

BIG_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "e2e_big.json")
BIG = json.load(open(BIG_PATH)) if os.path.exists(BIG_PATH) else {}


@pytest.fixture(scope="module")
def scratch():
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (6 << 30) else None
    d = tempfile.mkdtemp(prefix="parsnp_big_", dir=base)
    yield d
    shutil.rmtree(d, ignore_errors=True)


def inputs(name, base):
    d = os.path.join(base, name, "in")
    if name == "bact2000_p0":
        ref, gs, ids = synth.make_partition(0)
        return synth.write_set(d, ref, gs, ids)
    ref, gs = synth.make(name)
    return synth.write_set(d, ref, gs)


@pytest.mark.parametrize("name", ["bact200", "bact2000_p0", "rearr50"])
def test_baseline_size_against_reference(scratch, name):
    if name not in BIG:
        pytest.skip("no reference golden for %s in tests/golden/e2e_big.json" % name)
    want = BIG[name]
    rp, qs = inputs(name, scratch)
    assert len(qs) == want["n_queries"]
    for mode in ("generations", "in_order"):
        env = dict(os.environ, OMP_WAIT_POLICY="passive")
        if mode == "in_order":
            env["PARSNP_SEQUENTIAL_REPLAY"] = "1"
        out = os.path.join(scratch, name, "out_" + mode)
        rc, _ = driver.run_core(CORE_BIN, rp, qs, out, env=env, threads=24)
        assert rc == 0, open(os.path.join(out, "parsnp-aligner.err")).read()[-2000:]
        x = os.path.join(out, "parsnpAligner.xmfa")
        assert xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")) == want["log"], mode
        if xmfa_util.md5(x) != want["xmfa_md5"]:      # every byte; the MUM/LCB signature (a Python pass over a 1 GB file) only to say what differs
            assert xmfa_util.mum_lcb_signature(x) == want["signature"], mode + ": MUM / LCB coordinates differ"
            assert False, mode + ": same MUMs and LCBs, the gap columns differ"
        assert "NOTE" not in open(os.path.join(out, "parsnpAligner.log")).read()
        shutil.rmtree(out, ignore_errors=True)
    shutil.rmtree(os.path.join(scratch, name), ignore_errors=True)


def _big_scratch(need_gb):
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (need_gb << 30) else None
    if base is None and shutil.disk_usage(tempfile.gettempdir()).free < (need_gb << 30):
        pytest.skip("needs %d GB of scratch space" % need_gb)
    return tempfile.mkdtemp(prefix="parsnp_full_", dir=base)


def test_config4_all_partitions_and_merge():
    """BASELINE config 4 in full on the one GPU there is: 2000 x 5 Mb in the reference driver's Random(42) order, cut into
    8 partitions of 250 (parsnp:1553-1564), every partition through parsnp_core (24 host threads) one after the other,
    then the native merge (include/parsnp_merge.h).  Checked: partition 0 against the REFERENCE binary's golden (whole-XMFA
    md5 + log counters); every partition's XMFA self-consistent (251 rows per block, MUM columns, every record spells its
    genome interval); every trimmed partition holds the same reference intervals; the merged parsnp.xmfa holds all 2001
    sequences in every block, every record spells its genome interval, and its reference bases are the intersection's."""
    import time
    from parsnp_amd import partition_run
    d = _big_scratch(60)
    try:
        t0 = time.time()
        model, kw = synth.CONFIGS["bact2000"]
        ref, gs = synth.population(**kw)
        rp, qs = synth.write_set(os.path.join(d, "in"), ref, gs)
        del gs
        t1 = time.time()
        res = partition_run.run_partitioned(CORE_BIN, rp, driver.driver_order(qs), os.path.join(d, "out"), 250, keep_trimmed=True, threads=24)
        t2 = time.time()
        parts = res["partitions"]
        assert len(parts) == 8 and all(p["ok"] and p["queries"] == 250 for p in parts), [(p["index"], p["rc"]) for p in parts]
        if "bact2000_p0" in BIG:      # the driver's first chunk = the golden's partition 0
            x0 = os.path.join(parts[0]["dir"], "parsnpAligner.xmfa")
            assert xmfa_util.log_counters(os.path.join(parts[0]["dir"], "parsnpAligner.log")) == BIG["bact2000_p0"]["log"]
            assert xmfa_util.md5(x0) == BIG["bact2000_p0"]["xmfa_md5"]
        pieces = None
        for p in parts:
            x = os.path.join(p["dir"], "parsnpAligner.xmfa")
            st = xmfa_util.native_consistency(x, os.path.join(d, "in"), threads=16)
            assert st["bad_length"] == 0 and st["bad_mum_column"] == 0 and st["bad_sequence"] == 0 and st["missing_genomes"] == 0, (p["index"], st)
            assert st["min_rows"] == 251 and st["max_rows"] == 251 and st["lcbs"] > 500, st
            iv = x + ".trimmed.iv"
            tr = xmfa_util.native_consistency(x + ".trimmed", os.path.join(d, "in"), threads=16, intervals=iv)
            assert tr["bad_length"] == 0 and tr["bad_sequence"] == 0 and tr["shifted"] == 0, (p["index"], tr)
            mine = open(iv).read()
            assert pieces is None or mine == pieces, "partition %d: trimmed reference intervals differ" % p["index"]
            pieces = mine
            os.remove(x + ".trimmed")
        t3 = time.time()
        m = res["merged"]
        assert m["sequences"] == 2001 and m["clusters"] == len(pieces.splitlines()) > 500
        ms = xmfa_util.native_consistency(m["xmfa"], os.path.join(d, "in"), merged=True, threads=16)
        assert ms["lcbs"] == m["clusters"] and ms["min_rows"] == 2001 and ms["max_rows"] == 2001 and ms["sequences"] == 2001, ms
        assert ms["bad_length"] == 0 and ms["bad_mum_column"] == 0 and ms["bad_sequence"] == 0 and ms["shifted"] == 0 and ms["missing_genomes"] == 0, ms
        assert ms["ref_bases"] == m["ref_bases"] == tr["ref_bases"]
        assert m["ref_bases"] > 0.8 * len(ref)
        print("config 4: generate %.1f s, 8 partitions + merge %.1f s, partition checks %.1f s, merged check %.1f s; %d clusters, %d reference bases, merged XMFA %.1f GB"
              % (t1 - t0, t2 - t1, t3 - t2, time.time() - t3, m["clusters"], m["ref_bases"], os.path.getsize(m["xmfa"]) / 1e9))
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_config5_full_500_genomes():
    """BASELINE config 5 in full: 500 x 5 Mb, 5 % segregating sites, 10 % of every genome rearranged, --no-partition on one
    GPU (24 host threads): XMFA self-consistency (501 rows per block, MUM columns, every record spells its genome interval,
    reverse-strand records present), run-to-run determinism, and the same bytes from the sharded form of the run -- 4 ranks,
    each with its block of 125 query genomes resident, here sharing the one GPU and exchanging over gloo."""
    import subprocess, sys, time
    d = _big_scratch(30)
    try:
        t0 = time.time()
        ref, gs = synth.make("rearr500")
        rp, qs = synth.write_set(os.path.join(d, "in"), ref, gs)
        del gs
        t1 = time.time()
        sums, walls = [], []
        for rep in range(2):
            out = os.path.join(d, "out%d" % rep)
            t = time.time()
            rc, _ = driver.run_core(CORE_BIN, rp, qs, out, threads=24, env=dict(os.environ, OMP_WAIT_POLICY="passive"))
            walls.append(time.time() - t)
            assert rc == 0, open(os.path.join(out, "parsnp-aligner.err")).read()[-2000:]
            sums.append(xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")))
            assert "NOTE" not in open(os.path.join(out, "parsnpAligner.log")).read()
        assert sums[0] == sums[1]
        t2 = time.time()
        st = xmfa_util.native_consistency(os.path.join(d, "out0", "parsnpAligner.xmfa"), os.path.join(d, "in"), threads=16)
        assert st["lcbs"] > 5000 and st["min_rows"] == 501 and st["max_rows"] == 501, st
        assert st["bad_length"] == 0 and st["bad_mum_column"] == 0 and st["bad_sequence"] == 0 and st["missing_genomes"] == 0, st
        assert st["reverse"] > 1000, st
        t3 = time.time()
        shutil.rmtree(os.path.join(d, "out1"), ignore_errors=True)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        out = os.path.join(d, "sharded")
        os.makedirs(out)
        ini = os.path.join(out, "run.ini")
        open(ini, "w").write(driver.ini_text(rp, qs, out, threads=6))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1", "--master-port", "29591",
               "-m", "parsnp_amd.sharded", ini]
        p = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=root, OMP_WAIT_POLICY="passive"), cwd=out, timeout=1500)
        assert p.returncode == 0, p.stderr[-3000:]
        assert xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")) == sums[0]
        assert xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")) == xmfa_util.log_counters(os.path.join(d, "out0", "parsnpAligner.log"))
        print("config 5: generate %.1f s, whole process %.1f / %.1f s, check %.1f s, sharded x4 on one GPU %.1f s; %s" % (t1 - t0, walls[0], walls[1], t3 - t2, time.time() - t3, st))
    finally:
        shutil.rmtree(d, ignore_errors=True)
