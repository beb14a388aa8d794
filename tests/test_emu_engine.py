"""The engine's kernel functors (parsnp_amd/csrc/engine/kernels.h) and orchestration (engine_core.h), executed
sequentially on the host by tests/emu, against the CPU restatement.  Checks the LOGIC of what the GPU runs; the
GPU execution itself is checked by the -m gpu tests."""
import os

import numpy as np
import pytest

import oracles
from parsnp_amd.binding import Lib, Session
from seqgen import adversarial_case, mutate, random_seq
from parsnp_amd import driver, synth
import test_host_logic
import xmfa_util


@pytest.fixture(scope="module")
def libs(emu, cpu_checkers):
    return Lib(emu[0]), oracles.load_restatement()


def same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a[:4], b[:4]))


def test_random_regions(libs):
    E, O = libs
    rng = np.random.default_rng(5)
    total = 0
    for it in range(400):
        ref, qs = adversarial_case(rng, 10, 90, int(rng.integers(1, 5)))
        minsize = int(rng.integers(1, 12))
        a = oracles.restatement_multi_mum(O, [ref] + qs, minsize, 1)
        with Session(E, [ref] + qs) as s:
            b = s.whole(minsize)
        assert same(a, b), (it, ref, qs, minsize)
        total += len(a[0])
    assert total > 500


def test_master_ep_both_kernels(libs):
    """Master.EP from the genomes' segments (MasterEPSeg, the shipped kernel: every other test here runs it) and by the round-4
    kernel (tune master_seg = 0): the same candidates, against the restatement; regions longer than one 256-position chunk and more
    genomes than a wavefront has lanes"""
    E, O = libs
    rng = np.random.default_rng(55)
    total = 0
    for it in range(60):
        if it % 10 == 0:
            ref = random_seq(rng, 1500)
            qs = [mutate(rng, ref, sub=0.03, indel=0.004) for _ in range(70 if it % 20 == 0 else 5)]
        else:
            ref, qs = adversarial_case(rng, 10, 700, int(rng.integers(1, 5)))
        minsize = int(rng.integers(4, 14))
        want = oracles.restatement_multi_mum(O, [ref] + qs, minsize, 1)
        for seg in (1, 0):
            with Session(E, [ref] + qs) as s:
                s.tune("master_seg", seg)
                got = s.whole(minsize)
            assert same(want, got), (it, seg, minsize)
        total += len(want[0])
    assert total > 200


def test_event_order_both_ways(libs):
    """the events put in order by buckets (EventBucketCount ... CoarseFromBuckets, shipped: every other test here runs it) and by
    the gather + radix sort of rounds 1-5 (tune bucket_sort = 0): the same candidates as the restatement -- whole genomes with
    repeats (long buckets: Shell's gaps), regions of several 256-position blocks, batches with grouped small regions beside them"""
    E, O = libs
    rng = np.random.default_rng(77)
    total = 0
    for it in range(40):
        if it % 8 == 0:
            unit = random_seq(rng, 40)
            ref = random_seq(rng, 600) + unit * 12 + random_seq(rng, 500)      # a tandem repeat: many events of one block
            qs = [mutate(rng, ref, sub=0.02, indel=0.003) for _ in range(6)]
        else:
            ref, qs = adversarial_case(rng, 10, 900, int(rng.integers(1, 5)))
        minsize = int(rng.integers(4, 14))
        want = oracles.restatement_multi_mum(O, [ref] + qs, minsize, 1)
        for how in (1, 0):
            with Session(E, [ref] + qs) as s:
                s.tune("bucket_sort", how)
                got = s.whole(minsize)
            assert same(want, got), (it, how, minsize)
        total += len(want[0])
    assert total > 200
    for how in (1, 0):      # batches: regions above 128 bases (sorted part) beside small ones (grouped part)
        rng2 = np.random.default_rng(19)
        assert sum(batch_case(rng2, E, O, tune=("bucket_sort", how)) for _ in range(4)) > 30


def batch_case(rng, E, O, n_regions=40, glen=3000, nq=4, big_minsize=False, tune=None):
    ref = random_seq(rng, glen)
    qs = []
    for g in range(nq):
        q = mutate(rng, ref, sub=0.04, indel=0.004)
        if g % 2:
            a = glen // 4; b = a + glen // 3
            q = q[:a] + oracles.revcomp(q[a:b]) + q[b:]
        qs.append(q)
    seqs = [ref] + qs
    starts = np.zeros((n_regions, nq + 1), np.int64); lens = np.zeros_like(starts); mins = np.zeros(n_regions, np.int32)
    for r in range(n_regions):
        aligned = big_minsize or rng.random() < 0.5   # the same window in every genome -> multi-MUMs exist
        f0 = rng.random(); f1 = rng.random()
        for g, s in enumerate(seqs):
            if aligned:
                ln = min(len(s), int(200 + f1 * (len(s) // 2))); st = int(f0 * (len(s) - ln))
            else:
                ln = int(rng.integers(0, 400)) if rng.random() < 0.8 else int(rng.integers(400, len(s)))
                st = int(rng.integers(0, len(s) - ln + 1))
            starts[r, g] = st; lens[r, g] = ln
        mins[r] = int(rng.integers(14, 30)) if big_minsize else int(rng.integers(3, 14))
    with Session(E, seqs) as s:
        if tune:
            s.tune(*tune)
        got = s.multi_mum_batch(starts, lens, mins)
    n = 0
    for r in range(n_regions):
        sub = [seqs[g][starts[r, g]:starts[r, g] + lens[r, g]] for g in range(nq + 1)]
        want = oracles.restatement_multi_mum(O, sub, int(mins[r]), 1)
        assert same(want, got[r]), (r, starts[r], lens[r], mins[r])
        n += len(want[0])
    return n


def test_batched_regions(libs):
    E, O = libs
    rng = np.random.default_rng(6)
    assert sum(batch_case(rng, E, O) for _ in range(5)) > 50
    assert batch_case(rng, E, O, n_regions=12, glen=20000, nq=3, big_minsize=True) > 5


def small_region_batch(rng, E, O, nq, n_regions, distinct, minsize_hi=10, low_complexity=False):
    """a recursion-shaped batch: every region at most 128 bases in every genome, the genomes a population that carries
    `distinct` versions of each stretch (+ a few private ones)"""
    glen = 4000
    ref = (random_seq(rng, 7) * (glen // 7 + 1))[:glen] if low_complexity else random_seq(rng, glen)
    versions = [ref] + [mutate(rng, ref, sub=0.03, indel=0.0) for _ in range(distinct - 1)]      # (no indels: one window fits all)
    qs = []
    for g in range(nq):
        q = bytearray(versions[int(rng.integers(0, distinct))])
        if rng.random() < 0.1:
            at = int(rng.integers(0, glen)); q[at:at + 1] = b"ACGT"[int(rng.integers(0, 4)):][:1]
        if g % 5 == 4:
            a = glen // 4; b = a + glen // 3
            q = bytearray(bytes(q[:a]) + oracles.revcomp(bytes(q[a:b])) + bytes(q[b:]))
        qs.append(bytes(q))
    seqs = [ref] + qs
    starts = np.zeros((n_regions, nq + 1), np.int64); lens = np.zeros_like(starts); mins = np.zeros(n_regions, np.int32)
    for r in range(n_regions):
        ln = int(rng.integers(0, 129)); st = int(rng.integers(0, glen - 140))
        for g in range(nq + 1):
            jitter = int(rng.integers(-3, 4)) if rng.random() < 0.2 else 0
            lens[r, g] = max(0, min(128, ln + jitter)); starts[r, g] = st + (int(rng.integers(0, 4)) if rng.random() < 0.1 else 0)
        mins[r] = int(rng.integers(2, minsize_hi))
    got = {}
    for grouped in (1, 0):
        with Session(E, seqs) as s:
            s.tune("group_small", grouped)
            got[grouped] = s.multi_mum_batch(starts, lens, mins)
            counts = dict(s.last_timing())
        if not grouped or distinct <= 40:      # (more distinct pieces than the table holds: most regions are handed back, short ones may stay)
            assert (counts.get("n_grouped", 0) > 0) == bool(grouped), counts
    n = 0
    for r in range(n_regions):
        assert same(got[0][r], got[1][r]), (r, starts[r], lens[r], mins[r])
        if r % 4 == 0:
            sub = [seqs[g][starts[r, g]:starts[r, g] + lens[r, g]] for g in range(nq + 1)]
            want = oracles.restatement_multi_mum(O, sub, int(mins[r]), 1)
            assert same(want, got[1][r]), (r, starts[r], lens[r], mins[r])
            n += len(want[0])
    return n


@pytest.mark.parametrize("nq,distinct,low", [(5, 2, False), (63, 4, False), (64, 3, False), (70, 6, False), (150, 5, False), (200, 80, False), (200, 28, False), (40, 3, True)])
def test_small_regions_once_per_distinct_piece(libs, nq, distinct, low):
    """GroupedPairEvents (store_kernels.h): the events of a batch's small regions computed once per distinct query piece and
    laid out per pair without the sort -- the same multi-MUMs as pair by pair (group_small = 0) and as the restatement; with
    more genomes than lanes, with more distinct pieces than the table holds (80 versions) and with more events per piece than
    the lists hold (a 7-base tandem repeat), where a region is handed back to SmallPairEvents"""
    E, O = libs
    rng = np.random.default_rng(1000 + nq)
    assert small_region_batch(rng, E, O, nq, 48 if nq < 100 else 24, distinct, low_complexity=low) >= (0 if low or distinct > 20 else 5)


def test_long_minimum_lengths(libs):
    """minsize > 47 makes the sampling stride exceed one 32-base window (left arm continues from memory), and matches
    longer than the 64 prefetched bases continue from memory on the right"""
    E, O = libs
    rng = np.random.default_rng(11)
    for minsize in (48, 60, 90, 130):
        ref = random_seq(rng, 6000)
        qs = [mutate(rng, ref, sub=0.004, indel=0.0005), oracles.revcomp(mutate(rng, ref, sub=0.004))]
        a = oracles.restatement_multi_mum(O, [ref] + qs, minsize, 1)
        with Session(E, [ref] + qs) as s:
            b = s.whole(minsize)
        assert same(a, b) and len(a[0]) > 3, minsize


def test_events(libs):
    E, O = libs
    rng = np.random.default_rng(7)
    for it in range(300):
        ref, (q,) = adversarial_case(rng, 10, 120)
        min_len = int(rng.integers(1, 20))
        K = min(min_len, 16)
        for strand in (0, 1):
            qq = oracles.revcomp(q) if strand else q
            j0, l0, n0, r0 = oracles.restatement_events(O, ref, qq, min_len)
            j1, l1, n1, r1 = E.find_events(ref, q, min_len, strand)
            a = sorted(zip(l0.tolist(), j0.tolist(), n0.tolist(), [x if x >= K else 0 for x in r0.tolist()]))
            b = sorted(zip(l1.tolist(), j1.tolist(), n1.tolist(), r1.tolist()))
            assert a == b, (it, ref, q, min_len, strand)


def test_end_to_end_with_emulated_engine(emu, tmp_path):
    r, gs = synth.make("viral50")
    rp, qs = synth.write_set(str(tmp_path / "in"), r, gs)
    test_host_logic.check(emu[1], "viral50", rp, qs, str(tmp_path / "out"))


@pytest.mark.parametrize("name", ["rearr6x300k", "pop6x200k"])
@pytest.mark.parametrize("variant", ["device_rows_and_flags", "host_overlap", "host_rows"])
def test_device_rows_and_overlap_flags(emu, tmp_path, name, variant):
    """the HOST route (what a step falls back to when the resident route does not apply): MUM rows (start, strand, flags) and the
    cheap overlap flags built by the engine (CompactCandidates, Dirty* kernels) feed the threaded anchor validation; thresholds
    lowered so that the small sets take that route.  The variants switch the overlap test / the row construction back to the
    host: same bytes either way."""
    r, gs = synth.make(name)
    rp, qs = synth.write_set(str(tmp_path / "in"), r, gs)
    env = dict(os.environ, PARSNP_PARALLEL_MIN="8", PARSNP_FREE_MIN="2", PM_DIRTY_MIN="8", PARSNP_NO_RESIDENT="1")      # (the host route: the resident route has its own test below)
    if variant == "host_overlap":
        env["PARSNP_HOST_OVERLAP"] = "1"
    if variant == "host_rows":
        env["PARSNP_NO_DEVICE_ROWS"] = "1"
    out = str(tmp_path / "out")
    rc, _ = driver.run_core(emu[1], rp, qs, out, env=env, threads=4)
    err = open(os.path.join(out, "parsnp-aligner.err")).read()
    assert rc == 0, err[-2000:]
    want = test_host_logic.E2E[name]
    assert xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")) == want["xmfa_md5"]
    assert xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")) == want["log"]


@pytest.mark.parametrize("name,flagged_div,expect", [("viral50", 8, "resident"), ("pop6x200k", 8, "resident"), ("pop12x400k", 8, "resident"), ("pop12x400k", 1, "resident"), ("popinv12x400k", 8, "resident"),
                                                      ("rearr6x300k", 8, "resident"), ("rearr6x300k", 1, "resident"), ("messy", 8, "resident"), ("pchunk", 8, "host")])
def test_resident_route(emu, tmp_path, name, flagged_div, expect):
    """The resident route (csrc/host/resident.cpp over include/parsnp_mum.h's pm_store_*: candidates validated and trimmed, regions
    walked, generations validated, chaining verdicts and inter-LCB fillers computed on rows that stay with the engine) in the
    kernel emulation: the reference's bytes where it is taken (collinear sets; pop12x400k has 166 flagged anchor candidates, 6 of
    them tangled), where the engine declines the anchor list (rearranged: more than one row in eight overlaps an earlier one)
    and where the route is left because the reference's processing order would show (the step is repeated on the host route).
    flagged_div = 1 lets every anchor list onto the route, so that the trimming kernels see the rearranged set as well."""
    rp, qs, kw = test_host_logic.harsh_inputs(name, str(tmp_path))
    log = str(tmp_path / "route.log")
    env = dict(os.environ, PARSNP_PARALLEL_MIN="8", PARSNP_FREE_MIN="2", PM_DIRTY_MIN="8", PM_FLAGGED_DIV=str(flagged_div), PARSNP_RESIDENT_LOG=log, PARSNP_CHECK_ZERO="1")
    out = str(tmp_path / "out")
    rc, _ = driver.run_core(emu[1], rp, qs, out, env=env, threads=4, **kw)
    assert rc == 0, open(os.path.join(out, "parsnp-aligner.err")).read()[-2000:]
    want = test_host_logic.E2E[name]
    assert xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")) == want["xmfa_md5"]
    assert xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")) == want["log"]
    route = open(log).read()
    assert ("resident=1" in route) == (expect == "resident"), route
    assert ("retry=1" in route) == (expect == "left"), route


@pytest.mark.parametrize("name", ["pop12x400k", "viral50"])
@pytest.mark.parametrize("variant", ["host_list_logic", "reported_tie", "split_settle", "one_stage", "gate_closed", "clusters_unsure", "exact_tail", "serial_tangle"])
def test_resident_route_variants(emu, tmp_path, name, variant):
    """the entry points the shipped route no longer calls, and its fall-backs: phases C-D by the host's list logic over
    pm_store_judge / _unmark / _fill (PARSNP_NO_DEVICE_CHAIN: pm_store_chain_* switched off; PM_CHAIN_TIE: the device reports two
    MUMs with one reference start and the caller takes over), pm_store_settle + pm_store_seeds as two calls, every generation its own pm_store_validate call -- the reference's bytes
    each time, as from the one-call forms (test_resident_route)"""
    rp, qs, kw = test_host_logic.harsh_inputs(name, str(tmp_path))
    log = str(tmp_path / "route.log")
    env = dict(os.environ, PARSNP_PARALLEL_MIN="8", PARSNP_FREE_MIN="2", PM_DIRTY_MIN="8", PARSNP_RESIDENT_LOG=log, PARSNP_CHECK_ZERO="1")
    env.update({"host_list_logic": {"PARSNP_NO_DEVICE_CHAIN": "1"}, "reported_tie": {"PM_CHAIN_TIE": "1"}, "split_settle": {"PARSNP_SPLIT_SETTLE": "1"}, "one_stage": {"PARSNP_ONE_STAGE": "1"}, "gate_closed": {"PM_STAGE_GATE": "1"}, "clusters_unsure": {"PM_CLUSTER_UNSURE": "1"}, "exact_tail": {"PM_FAST_TAIL": "0"}, "serial_tangle": {"PM_TANGLE_ROUNDS": "0"}}[variant])
    out = str(tmp_path / "out")
    rc, _ = driver.run_core(emu[1], rp, qs, out, env=env, threads=4, **kw)
    assert rc == 0, open(os.path.join(out, "parsnp-aligner.err")).read()[-2000:]
    want = test_host_logic.E2E[name]
    assert xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")) == want["xmfa_md5"]
    assert xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")) == want["log"]
    route = open(log).read()
    assert "resident=1" in route and ("chain=1" in route) == (variant in ("split_settle", "one_stage", "gate_closed", "clusters_unsure", "exact_tail", "serial_tangle")), route
    if variant == "clusters_unsure" and name != "viral50":      # the collinear test of the clusters reported failure: ClustersCollide found them disjoint, the generations ran
        assert "exact=0" not in route, route


def test_unaligned_twice_on_the_resident_route(emu, tmp_path):
    """step, write, step, write in ONE process with unaligned=1 on the resident route: the writer attaches the host's layout
    bitmaps to the image it fetched, and the next step must not inherit them (they pointed into freed memory once)"""
    import json
    from parsnp_amd.core_api import CoreRun
    rp, qs, kw = test_host_logic.harsh_inputs("pop6x200k", str(tmp_path))
    out = str(tmp_path / "out")
    os.makedirs(out)
    ini = os.path.join(out, "parsnpAligner.ini")
    open(ini, "w").write(driver.ini_text(rp, qs, out, unaligned=1, threads=2))
    code = (
        "import sys, hashlib, os; sys.path.insert(0, %r)\n"
        "from parsnp_amd.core_api import CoreRun\n"
        "r = CoreRun(%r, lib_path=%r)\n"
        "res = []\n"
        "for k in range(3):\n"
        "    s = r.step(); assert s['resident'] == 1, s\n"
        "    assert r.write() == 0\n"
        "    res.append((s['mums'], s['lcbs'], hashlib.md5(open(os.path.join(%r, 'parsnpAligner.xmfa'), 'rb').read()).hexdigest(), hashlib.md5(open(os.path.join(%r, 'parsnp.unalign'), 'rb').read()).hexdigest()))\n"
        "assert res[0] == res[1] == res[2], res\n"
        "print(res[0][2])\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ini, os.path.join(os.path.dirname(emu[0]), "libparsnp_core_emu.so"), out, out))
    env = dict(os.environ, PARSNP_PARALLEL_MIN="8", PARSNP_FREE_MIN="2", PM_DIRTY_MIN="8")
    import subprocess, sys
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=out)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-3000:]
    assert p.stdout.strip().splitlines()[-1] == test_host_logic.E2E["pop6x200k"]["xmfa_md5"]


def test_work_budget_retry(libs, monkeypatch):
    """a region whose repeat structure exhausts the per-thread work budget is run again with a larger one instead of
    failing the run (engine_core.h: run); a budget of 48 steps (pm_session_tune) makes a 40-copy tandem repeat enough to trigger it"""
    E, O = libs
    rng = np.random.default_rng(77)
    unit = b"ACGTTGCA"
    ref = random_seq(rng, 300) + unit * 40 + random_seq(rng, 300)
    qs = [mutate(rng, ref, sub=0.01), random_seq(rng, 50) + unit * 37 + random_seq(rng, 200)]
    want = oracles.restatement_multi_mum(O, [ref] + qs, 9, 1)
    with Session(E, [ref] + qs) as s:
        s.tune("work_budget", 48)
        got = s.whole(9)
        retried = dict(s.last_timing()).get("budget_retries", 0)
    assert same(want, got)
    assert retried >= 1


def mumi_cases(rng, count):
    for it in range(count):
        ref, qs = adversarial_case(rng, 20, int(rng.choice([60, 250])), int(rng.integers(1, 4)))
        if it % 3 == 0:
            qs[0] = mutate(rng, ref, sub=0.03)
        if it % 40 == 0:
            qs[0] = ref
        yield ref, qs


def check_mumi(E, O, count, seed):
    import ctypes as C
    O.oracle_mumi_coverage.restype = C.c_int64
    rng = np.random.default_rng(seed)
    tot = 0
    for ref, qs in mumi_cases(rng, count):
        want = [O.oracle_mumi_coverage(ref, C.c_int64(len(ref)), q, C.c_int64(len(q)), 1) for q in qs]
        with Session(E, [ref] + qs) as s:
            got = s.mumi_coverage()
        assert got == want, (ref, qs)
        tot += sum(want)
    assert tot > 1000


def test_mumi_coverage(libs):
    check_mumi(libs[0], libs[1], 300, 8)


def test_order_check_pieces(tmp_path):
    """tests/emu/order_check.cpp against the product's store_kernels.h: trimming on 64-bit masks (what the order check decides a noted
    candidate with) leaves the (shift, length) that settle_row leaves on the image the masks were read from, and both equal a
    genome-by-genome restatement of Aligner::trim; marks that nest give intervals that nest (the bounds of ForeignBound); the mask
    readers against base-by-base loops; order_key orders by (reference start, generation)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "order_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-w", os.path.join(root, "tests", "emu", "order_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr


def test_scan_operator_and_xcd_numbering(tmp_path):
    """tests/emu/scan_check.cpp against the product's kernels.h: the join of the wavefront scan is associative and a 64-lane
    segmented scan with the kernel's update rule (rounds, carry of the last lane) gives every event the state of the sequential
    rule (Test_UM + Intersect_UM's carry: furthest end, its first event in (l, j) order, second furthest end); xcd_item maps a
    launch of xcd_grid(n) workgroups onto the n items exactly once each"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "scan_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-w", os.path.join(root, "tests", "emu", "scan_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr
