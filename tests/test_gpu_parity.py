"""Parity of the HIP engine (parsnp_amd/lib/libparsnp_hip.so, through the C ABI) on a real MI355X:
against the CPU restatement on seeded inputs, against the reference's own csgmum code where oracle/_ref ships it,
against the committed goldens, and end to end through parsnp_core against the reference binary's goldens."""
import json
import os

import numpy as np
import pytest

import oracles
import test_emu_engine as T
import test_host_logic
import xmfa_util
from parsnp_amd import driver, synth
from parsnp_amd.binding import Lib, Session
from parsnp_amd.paths import CORE_BIN, CORE_HOOKS_BIN, HIP_LIB
from seqgen import adversarial_case, mutate, random_seq
from test_golden import G, mers, read_fasta

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def libs(cpu_checkers):
    H = Lib(HIP_LIB)          # raises if the HIP library is missing: no fallback
    assert H.provider == "hip"
    return H, oracles.load_restatement()


def test_random_regions(libs):
    H, O = libs
    rng = np.random.default_rng(15)
    total = 0
    for it in range(300):
        ref, qs = adversarial_case(rng, 10, 90, int(rng.integers(1, 5)))
        minsize = int(rng.integers(1, 12))
        a = oracles.restatement_multi_mum(O, [ref] + qs, minsize, 1)
        with Session(H, [ref] + qs) as s:
            b = s.whole(minsize)
        assert T.same(a, b), (it, ref, qs, minsize)
        total += len(a[0])
    assert total > 400


@pytest.mark.skipif(not oracles.have_reference(), reason="oracle/_ref/libcsgmum_ref.so not shipped")
def test_random_regions_vs_reference_code(libs):
    H, _ = libs
    R = oracles.load_reference()
    rng = np.random.default_rng(16)
    n = 0
    for it in range(300):
        ref, qs = adversarial_case(rng, 10, 90, int(rng.integers(1, 5)))
        if any(not any(c in ref for c in q) for q in qs) or any(not any(c in ref for c in oracles.revcomp(q)) for q in qs):
            continue
        minsize = int(rng.integers(3, 11))
        a = oracles.reference_multi_mum(R, [ref] + qs, minsize)
        with Session(H, [ref] + qs) as s:
            b = s.whole(minsize)
        assert T.same(a, b), (it, ref, qs, minsize)
        n += len(a[0])
    assert n > 300


def test_master_ep_both_kernels(libs):
    """Master.EP from the genomes' segments (MasterEPSeg, shipped) and by the round-4 kernel (tune master_seg = 0) on the device: the
    same candidates as the restatement, chunks of 256 positions crossed, 70 genomes (more than a wavefront has lanes), and both
    kernels on a 200 kb population where the candidates number thousands"""
    H, O = libs
    T.test_master_ep_both_kernels((H, O))
    ref, gs = synth.make("pop6x200k")
    got = []
    for seg in (1, 0):
        with Session(H, [ref] + gs) as s:
            s.tune("master_seg", seg)
            got.append(s.whole(19))
    assert len(got[0][0]) > 1000 and T.same(got[0], got[1])


def test_event_order_both_ways(libs):
    """the events put in order by (pair, block) buckets (shipped) and by the gather + radix sort of rounds 1-5 (tune bucket_sort = 0)
    on the device: the emulation's cases, and both on a 200 kb population (thousands of candidates, 780 blocks per pair)"""
    H, O = libs
    T.test_event_order_both_ways((H, O))
    ref, gs = synth.make("pop6x200k")
    got = []
    for how in (1, 0):
        with Session(H, [ref] + gs) as s:
            s.tune("bucket_sort", how)
            got.append(s.whole(19))
    assert len(got[0][0]) > 1000 and T.same(got[0], got[1])


def test_batched_regions(libs):
    H, O = libs
    rng = np.random.default_rng(17)
    assert sum(T.batch_case(rng, H, O) for _ in range(6)) > 50
    assert T.batch_case(rng, H, O, n_regions=24, glen=40000, nq=4, big_minsize=True) > 10


@pytest.mark.parametrize("nq,distinct,low", [(5, 2, False), (64, 3, False), (70, 6, False), (150, 5, False), (200, 80, False), (200, 28, False), (300, 9, False), (40, 3, True)])
def test_small_regions_once_per_distinct_piece(libs, nq, distinct, low):
    """GroupedPairEvents on the device (one wavefront per small region: pieces numbered through LDS, one diagonal scan per
    distinct piece and strand, per-pair event blocks without the sort) against pair by pair and the restatement"""
    H, O = libs
    rng = np.random.default_rng(2000 + nq)
    assert T.small_region_batch(rng, H, O, nq, 64 if nq < 100 else 32, distinct, low_complexity=low) >= (0 if low or distinct > 20 else 5)


def test_empty_and_ragged(libs):
    H, O = libs
    seqs = [b"ACGTACGTTTGACCA", b"", b"ACGTACGTTTGACCA", b"N" * 40, b"TGGTCAAACGTACGT"]
    with Session(H, seqs) as s:
        k, lon, sp, fw = s.whole(4)
        assert len(k) == 0                       # an empty genome has no match: no multi-MUM
        starts = np.zeros((3, 5), np.int64); lens = np.array([[15, 0, 15, 40, 15], [0, 0, 0, 0, 0], [15, 0, 15, 0, 15]], np.int64)
        out = s.multi_mum_batch(starts, lens, [4, 4, 4])
        assert all(len(o[0]) == 0 for o in out)
    seqs = [b"ACGTACGTTTGACCA", b"ACGTACGTTTGACCA", b"TGGTCAAACGTACGT"]
    a = oracles.restatement_multi_mum(O, seqs, 4, 1)
    with Session(H, seqs) as s:
        b = s.whole(4)
    assert T.same(a, b) and len(a[0]) == 1 and a[3][0].tolist() == [1, 0]


def test_events(libs):
    H, O = libs
    rng = np.random.default_rng(18)
    for it in range(150):
        ref, (q,) = adversarial_case(rng, 10, 120)
        min_len = int(rng.integers(1, 20)); K = min(min_len, 16)
        for strand in (0, 1):
            qq = oracles.revcomp(q) if strand else q
            j0, l0, n0, r0 = oracles.restatement_events(O, ref, qq, min_len)
            j1, l1, n1, r1 = H.find_events(ref, q, min_len, strand)
            a = sorted(zip(l0.tolist(), j0.tolist(), n0.tolist(), [x if x >= K else 0 for x in r0.tolist()]))
            b = sorted(zip(l1.tolist(), j1.tolist(), n1.tolist(), r1.tolist()))
            assert a == b, (it, ref, q, min_len, strand)


def test_mers_anchor_golden(libs, tmp_path):
    H, _ = libs
    ref, qs = mers(base=str(tmp_path))
    seqs = [read_fasta(ref)] + [read_fasta(p) for p in qs]
    g2 = np.load(os.path.join(G, "mers_anchor.npz"))
    with Session(H, seqs) as s:
        k, lon, sp, fw = s.whole(17)
    assert np.array_equal(k, g2["k"]) and np.array_equal(lon, g2["lon"]) and np.array_equal(sp, g2["sp"]) and np.array_equal(fw, g2["fwd"])


def test_medium_vs_restatement(libs):
    H, O = libs
    rng = np.random.default_rng(19)
    ref = random_seq(rng, 120000)
    qs = []
    for g in range(5):
        q = mutate(rng, ref, sub=0.02, indel=0.002)
        if g == 2:
            q = q[:30000] + oracles.revcomp(q[30000:70000]) + q[70000:]
        if g == 3:   # contig padding like ingest: 310 N
            q = q[:50000] + b"N" * 310 + q[50000:]
        qs.append(q)
    a = oracles.restatement_multi_mum(O, [ref] + qs, 19, 19)
    with Session(H, [ref] + qs) as s:
        b = s.whole(19)
    assert len(a[0]) > 300 and T.same(a, b)


def test_properties_at_scale(libs):
    """5 Mb genomes (BASELINE config-3 size, 4 queries): size-independent properties of the candidate list --
    every candidate is an exact match in every genome on the reported strand, maximal on at least one side in some
    genome, candidates are sorted and their reference intervals strictly advance; a planted inversion comes back on
    the reverse strand; determinism across two runs."""
    H, _ = libs
    ref, gs = synth.population(seed=21, n=5_000_000, n_genomes=4, div=0.02, indel_frac=0.05)
    g1 = gs[1]; gs[1] = g1[:1_000_000] + oracles.revcomp(g1[1_000_000:1_200_000]) + g1[1_200_000:]
    seqs = [ref] + gs
    with Session(H, seqs) as s:
        k, lon, sp, fw = s.whole(25)
        k2, lon2, sp2, fw2 = s.whole(25)
        timing = s.last_timing()
    assert np.array_equal(k, k2) and np.array_equal(lon, lon2) and np.array_equal(sp, sp2) and np.array_equal(fw, fw2)
    assert len(k) > 20000 and (lon >= 25).all()
    assert (np.diff(k) > 0).all() and (np.diff(k + lon) > 0).all()
    assert (fw[:, 1] == 0).sum() > 500 and (fw[:, 0] == 1).all()
    R = np.frombuffer(ref, np.uint8)
    rcs = [np.frombuffer(oracles.revcomp(g), np.uint8) for g in gs]
    fws = [np.frombuffer(g, np.uint8) for g in gs]
    idx = np.random.default_rng(0).choice(len(k), 3000, replace=False)
    for c in idx:
        want = R[k[c]:k[c] + lon[c]]
        for g in range(4):
            src = fws[g] if fw[c, g] else rcs[g]
            assert np.array_equal(src[sp[c, g]:sp[c, g] + lon[c]], want), (c, g)
    print("timing", timing)


@pytest.mark.parametrize("name,genomes", [("bact200", 40), ("rearr500", 24), ("bact200inv", 40)])
def test_xmfa_consistency_at_scale(libs, tmp_path, name, genomes):
    """BASELINE-size genomes (5 Mb; config 3 shape and config 5 shape with fewer genomes), whole run with 8 host threads:
    size-independent properties of the XMFA -- rows of an LCB have one length, MUM columns are gap-free and identical in
    every row, every record spells the genome interval its header names (reverse-complemented for '-' records) -- plus
    run-to-run determinism."""
    model, kw = synth.CONFIGS[name]
    kw = dict(kw, n_genomes=genomes)
    if name == "bact200inv":
        kw["inv_every"] = 8      # 5 of the 40 genomes carry their inversion
    ref, gs = {"population": synth.population, "pop_rearranged": synth.pop_rearranged, "pop_inverted": synth.pop_inverted}[model](**kw)
    rp, qs = synth.write_set(str(tmp_path / "in"), ref, gs)
    sums = []
    for rep in range(2):
        out = str(tmp_path / ("out%d" % rep))
        rc, _ = driver.run_core(CORE_BIN, rp, qs, out, threads=8)
        assert rc == 0, open(os.path.join(out, "parsnp-aligner.err")).read()[-2000:]
        sums.append(xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")))
    assert sums[0] == sums[1]
    st = xmfa_util.consistency(str(tmp_path / "out0" / "parsnpAligner.xmfa"), [ref] + gs)
    assert st["lcbs"] > 100 and st["records"] == st["lcbs"] * (genomes + 1)
    assert st["bad_length"] == 0 and st["bad_mum_column"] == 0 and st["bad_sequence"] == 0, st
    if name == "rearr500":
        assert st["reverse"] > 50, st
    print(name, st)


def test_mumi_coverage(libs):
    T.check_mumi(libs[0], libs[1], 200, 28)


E2E = json.load(open(os.path.join(G, "e2e.json")))


def test_parsnp_core_mers(libs, tmp_path):
    ref, qs = mers(base=str(tmp_path))
    test_host_logic.check(CORE_BIN, "mers", ref, qs, str(tmp_path / "out"))


@pytest.mark.parametrize("name", ["viral50", "pop6x200k", "rearr6x300k", "pop20x1m", "bact8", "popinv12x400k"])
def test_parsnp_core_synthetic(libs, tmp_path, name):
    r, gs = synth.make(name)
    rp, qs = synth.write_set(str(tmp_path / "in"), r, gs)
    test_host_logic.check(CORE_BIN, name, rp, qs, str(tmp_path / "out"))


@pytest.mark.parametrize("name,mode", [("pop20x1m", "resident"), ("pop20x1m", "generations"), ("pop20x1m", "in_order"), ("draft20x1m", "resident"), ("draft20x1m", "generations"),
                                       ("poprearr10x400k", "generations")])
def test_parsnp_core_replay_modes_threaded(libs, tmp_path, name, mode):
    """8 host threads: the resident route (the shipped binary's default for a long anchor list: generations validated on the
    device), the host route's generation-parallel replay of the recursion (clusters of regions validated concurrently, atomic
    marks) and its forced in-order replay give the reference's bytes"""
    if name == "pop20x1m":
        r, gs = synth.make(name)
        rp, qs = synth.write_set(str(tmp_path / "in"), r, gs); kw = {}
    else:
        rp, qs, kw = test_host_logic.harsh_inputs(name, str(tmp_path))
    env = dict(os.environ)
    if mode != "resident":
        env["PARSNP_NO_RESIDENT"] = "1"
    if mode == "in_order":
        env["PARSNP_SEQUENTIAL_REPLAY"] = "1"
    out = str(tmp_path / "out")
    rc, _ = driver.run_core(CORE_BIN if mode == "resident" else CORE_HOOKS_BIN, rp, qs, out, env=env, threads=8, **kw)
    assert rc == 0, open(os.path.join(out, "parsnp-aligner.err")).read()[-2000:]
    assert xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")) == E2E[name]["xmfa_md5"]
    assert xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")) == E2E[name]["log"]


@pytest.mark.parametrize("name", ["rearr6x300k", "poprearr10x400k", "pop20x1m"])
@pytest.mark.parametrize("variant", ["device_rows_and_flags", "host_overlap", "host_rows"])
def test_parsnp_core_device_rows_and_overlap_flags(libs, tmp_path, name, variant):
    """the HOST route (what a step falls back to when the resident route does not apply): the MUM rows, the cheap overlap flags
    and the list-order bits come from the device (CompactCandidates, DirtyExtent/Prefix/Mark) and feed the threaded anchor
    validation in place; switching either back to the host must not change a byte.  (parsnp_core_hooks = the product's sources
    with the test hooks of csrc/host/hooks.h compiled in; the shipped binary ignores these switches.)"""
    if name == "poprearr10x400k":
        rp, qs, kw = test_host_logic.harsh_inputs(name, str(tmp_path))
    else:
        r, gs = synth.make(name)
        rp, qs = synth.write_set(str(tmp_path / "in"), r, gs); kw = {}
    env = dict(os.environ, PARSNP_PARALLEL_MIN="8", PARSNP_FREE_MIN="2", PM_DIRTY_MIN="8", PARSNP_NO_RESIDENT="1")      # (routes of the HOST route)
    if variant == "host_overlap":
        env["PARSNP_HOST_OVERLAP"] = "1"
    if variant == "host_rows":
        env["PARSNP_NO_DEVICE_ROWS"] = "1"
    env["PARSNP_DEBUG_TIMERS"] = "1"
    out = str(tmp_path / "out")
    rc, _ = driver.run_core(CORE_HOOKS_BIN, rp, qs, out, env=env, threads=8, **kw)
    err = open(os.path.join(out, "parsnp-aligner.err")).read()
    assert rc == 0, err[-2000:]
    assert xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")) == E2E[name]["xmfa_md5"]
    assert xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")) == E2E[name]["log"]


@pytest.mark.parametrize("name,flagged_div,expect", [("viral50", 8, "resident"), ("pop6x200k", 8, "resident"), ("pop12x400k", 8, "resident"), ("pop12x400k", 1, "resident"), ("popinv12x400k", 8, "resident"), ("pop20x1m", 8, "resident"),
                                                      ("bact8", 8, "resident"), ("rearr6x300k", 8, "resident"), ("rearr6x300k", 1, "resident"), ("poprearr10x400k", 1, "resident"), ("messy", 8, "resident"), ("pchunk", 8, "host")])
def test_parsnp_core_resident_route(libs, tmp_path, name, flagged_div, expect):
    """The resident route on the device (store_kernels.h through pm_store_*): the reference's bytes where it is taken, where the
    engine declines the anchor list and where the route is left and the step repeated on the host route; thresholds lowered so
    that the small sets take it (pop12x400k: 166 flagged anchor candidates, 6 tangled; flagged_div = 1 sends rearranged lists
    through the trimming kernels too).  Three steps in one process: the stores start over with every anchor call."""
    rp, qs, kw = test_host_logic.harsh_inputs(name, str(tmp_path))
    log = str(tmp_path / "route.log")
    env = dict(os.environ, PARSNP_PARALLEL_MIN="8", PARSNP_FREE_MIN="2", PM_DIRTY_MIN="8", PM_FLAGGED_DIV=str(flagged_div), PARSNP_RESIDENT_LOG=log, PARSNP_CHECK_ZERO="1")
    out = str(tmp_path / "out")
    rc, _ = driver.run_core(CORE_HOOKS_BIN, rp, qs, out, env=env, threads=8, **kw)
    assert rc == 0, open(os.path.join(out, "parsnp-aligner.err")).read()[-2000:]
    assert xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")) == E2E[name]["xmfa_md5"]
    assert xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")) == E2E[name]["log"]
    route = open(log).read()
    assert ("resident=1" in route) == (expect == "resident"), route
    assert ("retry=1" in route) == (expect == "left"), route


@pytest.mark.parametrize("name", ["pop12x400k", "pop20x1m"])
@pytest.mark.parametrize("variant", ["one_call_forms", "host_list_logic", "reported_tie", "split_settle", "one_stage", "gate_closed", "clusters_unsure", "exact_tail", "serial_tangle"])
def test_parsnp_core_resident_route_variants(libs, tmp_path, name, variant):
    """phases C-D from the device in one call (pm_store_chain_*: sort, chaining with the ratio test in the reference's float /
    double mix, LCB filter, second pass, fillers) and validation + seed regions in one call (pm_store_settle_seeds), against the
    forms they replaced -- the host's list logic over pm_store_judge / _unmark / _fill, the device reporting a tie and handing
    over, pm_store_settle + pm_store_seeds as two calls: the reference's bytes each time"""
    rp, qs, kw = test_host_logic.harsh_inputs(name, str(tmp_path))
    log = str(tmp_path / "route.log")
    env = dict(os.environ, PARSNP_PARALLEL_MIN="8", PARSNP_FREE_MIN="2", PM_DIRTY_MIN="8", PARSNP_RESIDENT_LOG=log, PARSNP_CHECK_ZERO="1")
    env.update({"one_call_forms": {}, "host_list_logic": {"PARSNP_NO_DEVICE_CHAIN": "1"}, "reported_tie": {"PM_CHAIN_TIE": "1"}, "split_settle": {"PARSNP_SPLIT_SETTLE": "1"}, "one_stage": {"PARSNP_ONE_STAGE": "1"}, "gate_closed": {"PM_STAGE_GATE": "1"}, "clusters_unsure": {"PM_CLUSTER_UNSURE": "1"}, "exact_tail": {"PM_FAST_TAIL": "0"}, "serial_tangle": {"PM_TANGLE_ROUNDS": "0"}}[variant])
    out = str(tmp_path / "out")
    rc, _ = driver.run_core(CORE_HOOKS_BIN, rp, qs, out, env=env, threads=8, **kw)
    assert rc == 0, open(os.path.join(out, "parsnp-aligner.err")).read()[-2000:]
    assert xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")) == E2E[name]["xmfa_md5"]
    assert xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")) == E2E[name]["log"]
    route = open(log).read()
    assert "resident=1" in route and ("chain=1" in route) == (variant in ("one_call_forms", "split_settle", "one_stage", "gate_closed", "clusters_unsure", "exact_tail", "serial_tangle")), route
    if variant == "clusters_unsure":      # the collinear test of the clusters reported failure: ClustersCollide found them disjoint, the generations ran
        assert "exact=0" not in route, route


@pytest.mark.parametrize("name", ["mers", "messy", "pop6x200k_p"])
def test_parsnp_core_calcmumi(libs, tmp_path, name):
    test_host_logic.check_mumi(CORE_BIN, name, str(tmp_path))


@pytest.mark.parametrize("name", ["poprearr10x400k", "messy", "pchunk", "draft8x300k", "draft20x1m"])
def test_parsnp_core_harsh_inputs(libs, tmp_path, name):
    rp, qs, kw = test_host_logic.harsh_inputs(name, str(tmp_path))
    out = str(tmp_path / "out")
    rc, _ = driver.run_core(CORE_BIN, rp, qs, out, **kw)
    assert rc == 0, open(os.path.join(out, "parsnp-aligner.err")).read()[-2000:]
    want = E2E[name]
    assert xmfa_util.mum_lcb_signature(os.path.join(out, "parsnpAligner.xmfa")) == want["signature"]
    assert xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")) == want["xmfa_md5"]
    assert xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")) == want["log"]


@pytest.mark.skipif(not os.path.exists(os.path.join(oracles.REFDIR, "parsnp_core_ref")), reason="reference binary not shipped")
def test_parsnp_core_vs_reference_binary_fresh_input(libs, tmp_path):
    """an input that has no committed golden: run the shipped reference binary and the product side by side"""
    r, gs = synth.population(seed=77, n=400_000, n_genomes=9, div=0.02, indel_frac=0.05)
    g = gs[4]; gs[4] = g[:100_000] + oracles.revcomp(g[100_000:140_000]) + g[140_000:]
    rp, qs = synth.write_set(str(tmp_path / "in"), r, gs)
    ref_bin = os.path.join(oracles.REFDIR, "parsnp_core_ref")
    rc1, _ = driver.run_core(ref_bin, rp, qs, str(tmp_path / "ref"))
    rc2, _ = driver.run_core(CORE_BIN, rp, qs, str(tmp_path / "hip"))
    assert rc1 == 0 and rc2 == 0
    a, b = str(tmp_path / "ref" / "parsnpAligner.xmfa"), str(tmp_path / "hip" / "parsnpAligner.xmfa")
    assert xmfa_util.mum_lcb_signature(a) == xmfa_util.mum_lcb_signature(b)
    assert xmfa_util.md5(a) == xmfa_util.md5(b)
    assert xmfa_util.log_counters(str(tmp_path / "ref" / "parsnpAligner.log")) == xmfa_util.log_counters(str(tmp_path / "hip" / "parsnpAligner.log"))


def test_bench_two_ranks_smoke(libs, tmp_path):
    """bench.py's multi-rank path (barrier, max-over-ranks, aggregate value) with 2 ranks; on a 1-GPU box the ranks share
    the GPU and rendezvous over gloo, on a multi-GPU node the same command uses RCCL with one GPU per rank"""
    import subprocess, sys
    from conftest import ROOT
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29577",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "pop6x200k", "--cpu-sample", "0"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["genomes_per_gpu"] == 6
    assert abs(d["value"] - 2 * 6 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 0.02
    assert d["roofline"]["kernel"] and d["mums"] > 2000
    assert 0 < d["core_bp_in_every_partition"] <= d["core_bp_aligned"] // 2 + 1000


def test_bench_sharded_mode_one_rank(libs):
    """bench.py --mode sharded (one alignment over the GPUs, engine-owned RCCL) with the one rank a single-GPU box allows"""
    import subprocess, sys
    from conftest import ROOT
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "sharded", "--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "pop6x200k", "--cpu-sample", "0"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and "sharded" in d["config"]["parallelism"]
    assert d["mums"] > 2000 and "exchange_ep" in d["engine_ms"] and "exchange_states" in d["engine_ms"]


def test_bench_both_scalings_child_process(libs):
    """bench.py's default mode for N > 1 reports the sharded (strong-scaling) measurement of the same workload beside the
    partition-per-GPU headline, measured by a child process per rank on rank 0's genome files.  With one GPU the plumbing is
    exercised with one rank (PARSNP_BENCH_CHILD_TEST): the child's line comes back under `sharded_strong`, where the engine's RCCL communicator counts 1 rank"""
    import subprocess, sys
    from conftest import ROOT
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "pop6x200k", "--cpu-sample", "0"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, PARSNP_BENCH_CHILD_TEST="1"))
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["scaling"] == "weak" and d["n_gpus"] == 1 and d["n_ranks_seen_by_rccl"] is None and len(d["per_rank"]) == 1      # (no communicator exists at N = 1: nothing to report)
    ss = d["sharded_strong"]
    assert ss and "error" not in ss and "skipped" not in ss, ss
    assert ss["scaling"] == "strong" and ss["n_ranks_seen_by_rccl"] == 1 and ss["mums"] == d["mums"] and ss["lcbs"] == d["lcbs"]
    assert "exchange_ep" in ss["engine_ms"] and "sharded" in ss["config"]


@pytest.mark.parametrize("name,world", [("poprearr10x400k", 2), ("bact8", 4)])
def test_sharded_run_on_gpu(libs, tmp_path, name, world):
    """SURVEY 8e-2 with the HIP engine: `world` ranks, each with its block of the query genomes resident (on this 1-GPU
    box they share the GPU and exchange over gloo; with one GPU per rank the same launcher uses RCCL)"""
    import subprocess, sys
    from conftest import ROOT
    if name == "bact8":
        r, gs = synth.make(name)
        rp, qs = synth.write_set(str(tmp_path / "in"), r, gs); kw = {}
    else:
        rp, qs, kw = test_host_logic.harsh_inputs(name, str(tmp_path))
    out = str(tmp_path / "out")
    os.makedirs(out)
    ini = os.path.join(out, "run.ini")
    open(ini, "w").write(driver.ini_text(rp, qs, out, threads=4, **kw))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
           "--master-port", "29581", "-m", "parsnp_amd.sharded", ini]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=out, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert xmfa_util.mum_lcb_signature(os.path.join(out, "parsnpAligner.xmfa")) == E2E[name]["signature"]
    assert xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")) == E2E[name]["xmfa_md5"]
    assert xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")) == E2E[name]["log"]


def test_rccl_session_one_rank(libs):
    """the device-collective path (the engine's own RCCL communicator: all-reduce(min) of Master.EP in place, pack /
    all-gather / unpack of the candidate columns, all on the engine's stream) on a ONE-rank communicator -- what a box with
    a single GPU can run of it: same candidates as the plain session, for whole genomes, batches and calcmumi"""
    H, O = libs
    ident = H.rccl_unique_id()
    assert len(ident) == 128 and any(ident)
    rng = np.random.default_rng(41)
    for it in range(12):
        ref, qs = adversarial_case(rng, 10, int(rng.choice([90, 4000])), int(rng.integers(1, 6)))
        minsize = int(rng.integers(2, 12))
        with Session(H, [ref] + qs) as plain, Session(H, [ref] + qs, rccl=(0, 1, H.rccl_unique_id())) as sharded:
            assert T.same(plain.whole(minsize), sharded.whole(minsize)), (it, minsize)
            assert plain.mumi_coverage() == sharded.mumi_coverage()
            names = [n for n, _ in sharded.last_timing()]
    ref = random_seq(rng, 60000)
    qs = [mutate(rng, ref, sub=0.02, indel=0.002) for _ in range(5)]
    with Session(H, [ref] + qs) as plain, Session(H, [ref] + qs, rccl=(0, 1, ident)) as sharded:
        a, b = plain.whole(17), sharded.whole(17)
        assert len(a[0]) > 100 and T.same(a, b)
        assert "exchange_ep" in [n for n, _ in sharded.last_timing()] and "exchange_states" in [n for n, _ in sharded.last_timing()]


@pytest.mark.parametrize("name", ["pop6x200k", "rearr6x300k"])
def test_parsnp_core_sharded_binary_one_rank(libs, tmp_path, name):
    """parsnp_core started as rank 0 of a 1-rank sharded run (PARSNP_SHARD_WORLD / PARSNP_RCCL_ID_FILE): the id file
    hand-over, the RCCL session and the exchanges inside the drop-in binary; bytes as the reference's"""
    r, gs = synth.make(name)
    rp, qs = synth.write_set(str(tmp_path / "in"), r, gs)
    open(str(tmp_path / "rccl.id"), "wb").write(b"\x07" * 128 + b"a previous launch")      # a stale id file must not survive rank 0's start
    env = dict(os.environ, PARSNP_SHARD_WORLD="1", PARSNP_SHARD_RANK="0", PARSNP_RCCL_ID_FILE=str(tmp_path / "rccl.id"), PARSNP_SHARD_NONCE="launch-%s" % name)
    out = str(tmp_path / "out")
    rc, _ = driver.run_core(CORE_BIN, rp, qs, out, env=env, threads=4)
    assert rc == 0, open(os.path.join(out, "parsnp-aligner.err")).read()[-2000:]
    assert not os.path.exists(str(tmp_path / "rccl.id"))      # rank 0 removes its id file at exit
    assert xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")) == E2E[name]["xmfa_md5"]
    assert xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")) == E2E[name]["log"]


def test_sharded_run_rccl_two_gpus(libs, tmp_path):
    """two ranks, one GPU each, exchanges over the engine's RCCL communicator (xGMI): runs wherever two GPUs are visible.  Fails
    fast and loudly: the communicator must come up within 60 s (PARSNP_RCCL_TIMEOUT), the run within 5 minutes."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the build pool hands out single-GPU boxes; the driver's 8-GPU node runs it)")
    import subprocess, sys
    for name in ("poprearr10x400k", "pop20x1m"):      # the host route (rearranged) and the resident route (12 801 anchors)
        rp, qs, kw = test_host_logic.harsh_inputs(name, str(tmp_path / name))
        out = str(tmp_path / name / "out")
        os.makedirs(out)
        ini = os.path.join(out, "parsnpAligner.ini")
        open(ini, "w").write(driver.ini_text(rp, qs, out, threads=4, **kw))
        env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), PARSNP_RCCL_TIMEOUT="60")
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", "29533", "-m", "parsnp_amd.sharded", ini], cwd=out, env=env, capture_output=True, text=True, timeout=300)
        print("two-GPU sharded run of %s: exit code %d\n%s" % (name, p.returncode, p.stderr[-1500:]))
        assert p.returncode == 0, p.stderr[-3000:]
        assert xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")) == E2E[name]["xmfa_md5"]


def test_work_budget_retry(libs, monkeypatch):
    T.test_work_budget_retry(libs, monkeypatch)


def test_bench_two_gpus_both_scalings(libs):
    """two ranks, one GPU each: bench.py's one line carries the partition-per-GPU (weak) figure with torch's RCCL seeing both
    ranks AND the sharded (strong) figure of the same workload measured over the engine's own RCCL communicator by the child
    processes; runs wherever two GPUs are visible (the build pool hands out single-GPU boxes).  Loud: prints what RCCL counted,
    what the two exchanges cost on xGMI and every rank's step time; a silently serialised exchange shows as a red test, not as
    a flat curve -- two ranks must beat one by 10 % on 200 x 5 Mb (the searches of a step, 36 % of it, are what is sharded:
    1.22x is the ceiling at two ranks)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the build pool hands out single-GPU boxes; the driver's 8-GPU node runs it)")
    import subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PARSNP_RCCL_TIMEOUT="60")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--cpu-sample", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert one.returncode == 0, one.stderr[-3000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--cpu-sample", "0"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    ss = d["sharded_strong"]
    print("one GPU: %.0f genomes/s, %.2f ms per step" % (d1["value"], d1["ms_per_step"]))
    print("two GPUs, a partition each (weak): %.0f genomes/s; RCCL (torch) counts %s ranks; per rank: %s" % (d["value"], d["n_ranks_seen_by_rccl"], d["per_rank"]))
    print("two GPUs, one alignment sharded (strong): %s" % json.dumps(ss))
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["n_ranks_seen_by_rccl"] == 2 and len(d["per_rank"]) == 2
    assert ss and "error" not in ss and "skipped" not in ss, ss
    assert ss["scaling"] == "strong" and ss["n_gpus"] == 2 and ss["n_ranks_seen_by_rccl"] == 2
    em = ss.get("engine_ms") or {}
    print("exchanges on xGMI per step: all-reduce(min) of Master.EP %.3f ms, all-gather of the candidate columns %.3f ms" % (em.get("exchange_ep", -1), em.get("exchange_states", -1)))
    assert d["value"] >= 1.6 * d1["value"], "a partition per GPU does not scale: %.0f vs %.0f genomes/s" % (d["value"], d1["value"])
    assert ss["value"] >= 1.1 * d1["value"], "the sharded alignment is not faster on two GPUs than on one: %.0f vs %.0f genomes/s" % (ss["value"], d1["value"])
