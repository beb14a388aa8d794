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
from parsnp_amd.paths import CORE_BIN, CORE_HOOKS_BIN

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
    # the shipped binary (the resident route), then the host route in both of
    # its replay modes (test hooks: forced through the product's sources built with csrc/host/hooks.h's switches)
    for mode in ("shipped", "generations", "in_order"):
        env = dict(os.environ, OMP_WAIT_POLICY="passive")
        if mode != "shipped":
            env["PARSNP_NO_RESIDENT"] = "1"
        if mode == "in_order":
            env["PARSNP_SEQUENTIAL_REPLAY"] = "1"
        out = os.path.join(scratch, name, "out_" + mode)
        timing = os.path.join(scratch, name, "timing_" + mode + ".json")
        rc, _ = driver.run_core(CORE_BIN if mode == "shipped" else CORE_HOOKS_BIN, rp, qs, out, env=env, threads=24, timing=timing)
        if rc == 0:
            tj = json.load(open(timing))
            # every configuration stays on the resident route -- since round 6 the rearranged set too: its anchor list is settled on
            # the device (tangled rows in rounds) and clusters of waiting regions that meet in some genome wait for one another
            assert tj["resident"] == (1 if mode == "shipped" else 0), (mode, tj)
            if tj["resident"]:
                assert tj["d2h_bytes"] < 20e6 and tj["resident_retry"] == 0, tj      # rows stay on the device until the writer asks
                if name != "rearr50":
                    assert tj["device_chain"] == 1, tj      # ... and phases C-D came from the device in one call (pm_store_chain_*)
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
    """BASELINE config 4 in full on the one GPU there is (flows.config4_flow: 2000 x 5 Mb in the reference driver's
    Random(42) order, 8 partitions of 250 through parsnp_core one after the other, the native merge); partition 0 against the
    REFERENCE binary's golden (whole-XMFA md5 + log counters)."""
    import flows
    d = _big_scratch(60)
    try:
        model, kw = synth.CONFIGS["bact2000"]
        flows.config4_flow(CORE_BIN, d, kw, 250, 8, threads=24, golden=BIG.get("bact2000_p0"), min_lcbs=500)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_config5_full_500_genomes():
    """BASELINE config 5 in full (flows.config5_flow: 500 x 5 Mb, 5 % segregating sites, 10 % of every genome rearranged,
    --no-partition on one GPU: since round 6 against the REFERENCE binary's golden at this size -- whole-XMFA md5 (2.47 GB) + log
    counters, from the device-resident route --, then the same run sharded over 4 ranks that share the GPU and exchange over gloo)."""
    import flows
    d = _big_scratch(30)
    try:
        flows.config5_flow(CORE_BIN, d, "rearr500", {}, threads=24, ranks=4, min_lcbs=5000, min_reverse=1000, golden=BIG.get("rearr500"))
    finally:
        shutil.rmtree(d, ignore_errors=True)
