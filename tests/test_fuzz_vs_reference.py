"""Seeded random small genome sets -- random sizes, divergence, indels, inversions, translocations, duplicated blocks, N runs,
multi-contig files, random LCB parameters and thread counts -- through the REFERENCE binary and through our parsnp_core
side by side: same exit code, same XMFA bytes, same log counters.  CPU run: host logic over the CPU checker; GPU run: the
product."""
import os

import numpy as np
import pytest

import oracles
import xmfa_util
from parsnp_amd import driver, synth
from parsnp_amd.paths import CORE_BIN

CORE_HOOKS_BIN = os.path.join(os.path.dirname(CORE_BIN), "parsnp_core_hooks")      # the product's sources with the test hooks compiled in

REFBIN = os.path.join(oracles.REFDIR, "parsnp_core_ref")
pytestmark = pytest.mark.skipif(not os.path.exists(REFBIN), reason="reference binary not built/shipped")


def random_case(seed, big=False):
    rng = np.random.default_rng((5000 if big else 1000) + seed)
    n = int(rng.integers(200_000, 600_000)) if big else int(rng.integers(3_000, 60_000))
    ng = int(rng.integers(6, 15)) if big else int(rng.integers(2, 9))
    div = float(rng.choice([0.002, 0.01, 0.03, 0.06]))
    ref, gs = synth.population(seed=int(rng.integers(1, 1 << 30)), n=n, n_genomes=ng, div=div, indel_frac=float(rng.choice([0.0, 0.05, 0.3])))
    gs = [bytearray(g) for g in gs]
    for g in gs:                                            # per-genome structural edits
        for _ in range(int(rng.integers(0, 12 if big else 4))):
            L = len(g)
            a = int(rng.integers(0, max(1, L - 2000))); b = a + int(rng.integers(200, 20000 if big else 2000))
            kind = int(rng.integers(0, 5))
            if kind == 0:   g[a:b] = oracles.revcomp(bytes(g[a:b]))                       # inversion
            elif kind == 1: blk = g[a:b]; del g[a:b]; p = int(rng.integers(0, len(g))); g[p:p] = blk   # translocation
            elif kind == 2: g[b:b] = g[a:b]                                             # tandem duplication
            elif kind == 3: g[a:a] = b"N" * int(rng.integers(1, 400))                   # N run
            else:           del g[a:b]                                                  # deletion
    gs = [bytes(g) for g in gs]
    if rng.random() < 0.2:
        gs[0] = oracles.revcomp(gs[0])
    kw = {}
    if rng.random() < 0.3: kw["mincluster"] = int(rng.choice([10, 21, 60]))
    if rng.random() < 0.3: kw["clusterd"] = int(rng.choice([30, 100, 300, 1000]))
    if rng.random() < 0.2: kw["diagdiff"] = float(rng.choice([0.05, 0.12, 0.4, 20]))
    if rng.random() < 0.2: kw["anchors"] = str(int(rng.integers(12, 30))); kw["mums"] = str(int(rng.integers(8, 20)))
    if rng.random() < 0.15: kw["partpos"] = int(rng.integers(2000, max(2001, n // 2)))
    kw["threads"] = int(rng.choice([4, 8, 12])) if big else int(rng.choice([1, 1, 3, 6]))
    contigs = int(rng.choice([1, 1, 1, 3, 7]))
    return ref, gs, kw, contigs


def write(base, ref, gs, contigs, seed):
    if contigs == 1:
        return synth.write_set(base, ref, gs)
    rng = np.random.default_rng(seed)
    os.makedirs(base, exist_ok=True)

    def cut(g):
        if len(g) < 400 * contigs:
            return [g]
        edges = [0] + sorted(int(x) for x in rng.choice(np.arange(200, len(g) - 200), contigs - 1, replace=False)) + [len(g)]
        return [g[edges[i]:edges[i + 1]] for i in range(contigs)]
    rp = os.path.join(base, "ref.fna")
    synth.write_contigs(rp, "ref", cut(ref))
    qs = []
    for i, g in enumerate(gs):
        p = os.path.join(base, "g%04d.fna" % i)
        synth.write_contigs(p, "g%04d" % i, cut(g))
        qs.append(p)
    return rp, qs


def run(core, rp, qs, out, kw):
    rc, _ = driver.run_core(core, rp, qs, out, timeout=600, **kw)
    x = os.path.join(out, "parsnpAligner.xmfa")
    lg = os.path.join(out, "parsnpAligner.log")
    return (rc, xmfa_util.md5(x) if os.path.exists(x) else None, xmfa_util.log_counters(lg) if os.path.exists(x) else open(lg).read())


def side_by_side(core, seed, tmp_path, big=False):
    ref, gs, kw, contigs = random_case(seed, big)
    rp, qs = write(str(tmp_path / "in"), ref, gs, contigs, seed)
    a = run(REFBIN, rp, qs, str(tmp_path / "ref"), kw)
    b = run(core, rp, qs, str(tmp_path / "mine"), kw)
    assert a == b, (seed, kw, contigs)


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_host_logic(cpu_checkers, tmp_path, seed):
    side_by_side(cpu_checkers, seed, tmp_path)


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_resident_route(emu, tmp_path, monkeypatch, seed):
    """side by side with the reference binary for the RESIDENT route (resident.cpp / pm_store_*) in the kernel emulation: the
    thresholds lowered so that these small sets take it, and -- every other seed -- every anchor list let onto it however many
    of its rows overlap earlier ones (PM_FLAGGED_DIV=1: the trimming and the in-order settling of tangled rows at work on
    rearranged genomes).  Where the reference's order would show the route is left and the step repeated on the host route:
    the bytes must be the reference's either way."""
    for k, v in dict(PM_DIRTY_MIN="2", PARSNP_PARALLEL_MIN="2", PARSNP_FREE_MIN="1", PARSNP_CHECK_ZERO="1", PM_FLAGGED_DIV="1" if seed % 2 else "8",
                     PM_ATOMIC_MARKS="1" if seed % 4 == 0 else "0").items():      # (every fourth seed: the marks of an in-order list by atomic ORs as well)
        monkeypatch.setenv(k, v)
    ref, gs, kw, contigs = random_case(seed)
    if kw.get("threads", 1) < 2:
        kw["threads"] = 3
    rp, qs = write(str(tmp_path / "in"), ref, gs, contigs, seed)
    a = run(REFBIN, rp, qs, str(tmp_path / "ref"), kw)
    b = run(emu[1], rp, qs, str(tmp_path / "mine"), kw)
    assert a == b, (seed, kw, contigs)


def inversion_case():
    """a population with three clean inversions (20 kb in genome 2, 30 kb in genome 4, 5 kb in genome 5): the cheap running-extent
    test flags every anchor candidate inside an inverted block (a third of the list), hardly any of them overlaps anything"""
    ref, gs = synth.population(seed=77, n=150_000, n_genomes=6, div=0.01, indel_frac=0.0)
    gs = [bytearray(g) for g in gs]
    for k, (a, b) in ((1, (40000, 60000)), (3, (90000, 120000)), (4, (20000, 25000))):
        gs[k][a:b] = oracles.revcomp(bytes(gs[k][a:b]))
    return ref, [bytes(g) for g in gs]


def inversions_on_the_resident_route(core, tmp_path, monkeypatch, variant):
    for k, v in dict(PM_DIRTY_MIN="2", PARSNP_PARALLEL_MIN="2", PARSNP_FREE_MIN="1", PARSNP_CHECK_ZERO="1", PARSNP_RESIDENT_LOG=str(tmp_path / "route.log")).items():
        monkeypatch.setenv(k, v)
    if variant == "tangled_limit":
        monkeypatch.setenv("PM_TANGLED_MAX", "0")
    ref, gs = inversion_case()
    rp, qs = synth.write_set(str(tmp_path / "in"), ref, gs)
    a = run(REFBIN, rp, qs, str(tmp_path / "ref"), dict(threads=3))
    b = run(core, rp, qs, str(tmp_path / "mine"), dict(threads=3))
    assert a == b
    x = open(str(tmp_path / "mine" / "parsnpAligner.xmfa")).read()
    assert sum(1 for l in x.splitlines() if l.startswith(">") and " - " in l) >= 3      # reverse-strand LCB records: the inversions are aligned
    route = open(str(tmp_path / "route.log")).read()
    if variant == "taken":      # more than one row in eight flagged, a handful tangled: taken; reverse pairs judged by ChainJudge's ordered loop
        assert "resident=1" in route and "chain=1" in route and "retry=0" in route, route
    else:                       # the tangled rows counted after the collision test, over the limit: declined there, the host route from the same result
        assert "resident=0" in route and "overlap" in route, route


@pytest.mark.parametrize("variant", ["taken", "tangled_limit"])
def test_inversions_on_the_resident_route(emu, tmp_path, monkeypatch, variant):
    """the anchor list of a population with a few inversions stays on the device (round 5: the list is declined by its TANGLED rows,
    counted on the device, not by the rows the cheap test flags), reverse-strand members and all: the reference binary's bytes"""
    inversions_on_the_resident_route(emu[1], tmp_path, monkeypatch, variant)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["taken", "tangled_limit"])
def test_inversions_on_the_resident_route_on_gpu(tmp_path, monkeypatch, variant):
    inversions_on_the_resident_route(CORE_HOOKS_BIN, tmp_path, monkeypatch, variant)


def clusters_in_another_order(core, tmp_path, monkeypatch):
    """a population at 3 % divergence with two long inversions (30 kb of 150 kb in genome 2, 37 kb in genome 4): the recursion's
    seed regions inside an inverted block form several clusters, and the inverted genome holds them in the OPPOSITE order.  The
    collinear test of a generation's clusters (ClustersDisjoint) fails there; round 5's exact test (ClustersCollide: every cluster's
    extent ORed into a scratch image, a bit found set = two clusters meet) finds them disjoint and the generation runs on the
    device: the reference binary's bytes, the route kept"""
    for k, v in dict(PM_DIRTY_MIN="2", PARSNP_PARALLEL_MIN="2", PARSNP_FREE_MIN="1", PARSNP_CHECK_ZERO="1", PARSNP_RESIDENT_LOG=str(tmp_path / "route.log")).items():
        monkeypatch.setenv(k, v)
    n = 150_000
    ref, gs = synth.population(seed=78, n=n, n_genomes=6, div=0.03, indel_frac=0.0)
    gs = [bytearray(g) for g in gs]
    for k, (a, b) in ((1, (n // 4, n // 4 + n // 5)), (3, (n // 2, n // 2 + n // 4))):
        gs[k][a:b] = oracles.revcomp(bytes(gs[k][a:b]))
    rp, qs = synth.write_set(str(tmp_path / "in"), ref, [bytes(g) for g in gs])
    a = run(REFBIN, rp, qs, str(tmp_path / "ref"), dict(threads=3))
    b = run(core, rp, qs, str(tmp_path / "mine"), dict(threads=3))
    assert a == b
    route = open(str(tmp_path / "route.log")).read()
    assert "resident=1" in route and "retry=0" in route and "exact=0" not in route, route


def test_clusters_in_another_order(emu, tmp_path, monkeypatch):
    clusters_in_another_order(emu[1], tmp_path, monkeypatch)


@pytest.mark.gpu
def test_clusters_in_another_order_on_gpu(tmp_path, monkeypatch):
    clusters_in_another_order(CORE_HOOKS_BIN, tmp_path, monkeypatch)


def waiting_clusters(core, tmp_path, monkeypatch):
    """a population with a 60 kb inversion in every third genome: the seed region before an inverted block and the one behind it are
    neighbours in the inverted genome, so the two clusters they belong to on the reference MEET there, and the reference (which always
    pops the region with the smallest reference start) finishes the first and everything it leads to before the second.  Until round
    6 the route was left there; now the later cluster WAITS a generation (pm_store_validate's done[]: ClusterExtents / ClusterInvolved
    / ClusterDefer) -- the reference binary's bytes, the route kept, a region deferred"""
    for k, v in dict(PM_DIRTY_MIN="2", PARSNP_PARALLEL_MIN="2", PARSNP_FREE_MIN="1", PARSNP_CHECK_ZERO="1", PARSNP_RESIDENT_LOG=str(tmp_path / "route.log")).items():
        monkeypatch.setenv(k, v)
    ref, gs = synth.pop_inverted(seed=43, n=400_000, n_genomes=12, div=0.02, indel_frac=0.05, inv_every=3, inv_len=60_000)
    rp, qs = synth.write_set(str(tmp_path / "in"), ref, gs)
    a = run(REFBIN, rp, qs, str(tmp_path / "ref"), dict(threads=3))
    b = run(core, rp, qs, str(tmp_path / "mine"), dict(threads=3))
    assert a == b
    route = open(str(tmp_path / "route.log")).read()
    assert "resident=1" in route and "retry=0" in route and "deferred=0" not in route, route


def test_waiting_clusters(emu, tmp_path, monkeypatch):
    waiting_clusters(emu[1], tmp_path, monkeypatch)


@pytest.mark.gpu
def test_waiting_clusters_on_gpu(tmp_path, monkeypatch):
    waiting_clusters(CORE_HOOKS_BIN, tmp_path, monkeypatch)


def tied_mums(core, tmp_path, monkeypatch):
    """seed 8953 of round 6's campaign: two ACCEPTED MUMs with one reference start.  Aligner::trim walks the genomes once, in order: a
    later genome shifts the second candidate by a base onto the first one's reference start after the reference's own turn has passed
    (:1399-1477).  sort( mums ) (:338) is unstable, so what it does with the pair depends on the list it is handed -- the reference's is in
    doWork's processing order, the resident route's in generation order.  The device's chain reports the tie; the host then puts the
    recursion's MUMs into the reference's order (region by region by reference start and generation, row by row) before its own list
    logic sorts them: the reference binary's bytes (23 MUMs filtered, not 24), the route kept"""
    for k, v in dict(PM_DIRTY_MIN="2", PARSNP_PARALLEL_MIN="2", PARSNP_FREE_MIN="1", PARSNP_CHECK_ZERO="1", PARSNP_RESIDENT_LOG=str(tmp_path / "route.log")).items():
        monkeypatch.setenv(k, v)
    ref, gs, kw, contigs = random_case(8953, False)
    kw["threads"] = 3
    rp, qs = write(str(tmp_path / "in"), ref, gs, contigs, 8953)
    a = run(REFBIN, rp, qs, str(tmp_path / "ref"), kw)
    b = run(core, rp, qs, str(tmp_path / "mine"), kw)
    assert a == b
    route = open(str(tmp_path / "route.log")).read()
    assert "resident=1" in route and "retry=0" in route and "chain=0" in route, route      # (chain=0: the tie was reported, the host's list logic ran)


def test_tied_mums_on_the_resident_route(emu, tmp_path, monkeypatch):
    tied_mums(emu[1], tmp_path, monkeypatch)


@pytest.mark.gpu
def test_tied_mums_on_the_resident_route_on_gpu(tmp_path, monkeypatch):
    tied_mums(CORE_HOOKS_BIN, tmp_path, monkeypatch)


def order_case(core, tmp_path, monkeypatch, route):
    """seed 7059 of round 5's campaign: a recursion candidate (7 bases, one reverse-strand member) whose flipped member lies 20 kb
    outside its region, where another region's MUM gets marked.  The reference processes that other region LATER, trims the
    candidate to 2 bases against the anchors alone and accepts it (201 MUMs found); a generation scheme that happens to have
    marked the other MUM first trims it away (200).  Both routes now note such candidates and decide them again in the
    reference's order (ForeignBound and the kernels around it / the end of extend_generations): the resident route is left, the host's generations hand
    over to the in-order replay, and the log counts what the reference counts."""
    for k, v in dict(PM_DIRTY_MIN="2", PARSNP_PARALLEL_MIN="2", PARSNP_FREE_MIN="1", PARSNP_CHECK_ZERO="1", PARSNP_RESIDENT_LOG=str(tmp_path / "route.log")).items():
        monkeypatch.setenv(k, v)
    if route == "host":
        monkeypatch.setenv("PARSNP_NO_RESIDENT", "1")
    ref, gs, kw, contigs = random_case(7059, False)
    kw["threads"] = 3
    rp, qs = write(str(tmp_path / "in"), ref, gs, contigs, 7059)
    a = run(REFBIN, rp, qs, str(tmp_path / "ref"), kw)
    b = run(core, rp, qs, str(tmp_path / "mine"), kw)
    assert a == b
    assert any(l.startswith("Number of MUMs found") and l.split()[-1] == "201" for l in a[2]), a[2]
    log = open(str(tmp_path / "route.log")).read()
    if route == "resident":
        assert "retry=1" in log and "decided differently by the reference's order" in log, log


@pytest.mark.parametrize("route", ["resident", "host"])
def test_order_of_reads_outside_a_region(emu, tmp_path, monkeypatch, route):
    order_case(emu[1], tmp_path, monkeypatch, route)


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["resident", "host"])
def test_order_of_reads_outside_a_region_on_gpu(tmp_path, monkeypatch, route):
    order_case(CORE_HOOKS_BIN, tmp_path, monkeypatch, route)


def outside_write_case(core, tmp_path, monkeypatch, seed, stays, strict_route=True):
    """round 6: an ACCEPTED reverse-strand member outside its region (TMum.cpp:33-35 flips it against the whole genome) no longer
    ends the resident route by itself: its marks are checked against the regions of the store (OutsideWriteCheck) and the route is
    left only where a region on the wrong side of the reference's order covers them.  Seed 7030 of the round's emulation campaign
    holds one that is harmless -- the run stays and gives the reference's bytes --, seed 7174 one that is not: the step is repeated
    on the host route, the reference's bytes again."""
    for k, v in dict(PM_DIRTY_MIN="2", PARSNP_PARALLEL_MIN="2", PARSNP_FREE_MIN="1", PARSNP_CHECK_ZERO="1", PM_FLAGGED_DIV="1", PARSNP_CHECK_NEIGHBOURS="1",
                     PARSNP_RESIDENT_LOG=str(tmp_path / "route.log")).items():
        monkeypatch.setenv(k, v)
    ref, gs, kw, contigs = random_case(seed, False)
    kw["threads"] = 3
    rp, qs = write(str(tmp_path / "in"), ref, gs, contigs, seed)
    a = run(REFBIN, rp, qs, str(tmp_path / "ref"), kw)
    b = run(core, rp, qs, str(tmp_path / "mine"), kw)
    assert a == b
    log = open(str(tmp_path / "route.log")).read()
    if stays and not strict_route and "retry=1" in log:
        # (on the device, with the clusters side by side under the collinear test, the candidate's read outside its region may see
        # another wavefront's marks first: the order check then leaves the route -- the bytes above are the reference's either way)
        assert "decided differently by the reference's order" in log or "accepted outside its region" in log, log
    elif stays:
        assert "resident=1" in log and "retry=0" in log and "outside=1" in log, log
    else:
        # (the emulation, one thread at a time, always gets as far as the write; on the device the candidate's READ outside its region
        # may come first in time and fail the order check instead: either way the route is left and the host route answers)
        assert "retry=1" in log and ("accepted outside its region" in log or "decided differently by the reference's order" in log), log


@pytest.mark.parametrize("seed,stays", [(7030, True), (7174, False)])
def test_writes_outside_a_region(emu, tmp_path, monkeypatch, seed, stays):
    outside_write_case(emu[1], tmp_path, monkeypatch, seed, stays)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,stays", [(7030, True), (7174, False)])
def test_writes_outside_a_region_on_gpu(tmp_path, monkeypatch, seed, stays):
    outside_write_case(CORE_HOOKS_BIN, tmp_path, monkeypatch, seed, stays, strict_route=False)


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_host_route_over_device_rows(emu, tmp_path, monkeypatch, seed):
    """the same side by side for the HOST route over the kernel emulation (PARSNP_NO_RESIDENT: what a step falls back to),
    threaded, with the thresholds of its long-list paths lowered so that these small sets take them: MUM rows and overlap flags
    from the engine, threaded validation, generation-parallel replay -- scripts/fuzz_campaign.py with PARSNP_FUZZ_CORE=emu is
    the long form"""
    for k, v in dict(PM_DIRTY_MIN="2", PARSNP_PARALLEL_MIN="2", PARSNP_FREE_MIN="1", PARSNP_CHECK_ZERO="1", PARSNP_NO_RESIDENT="1").items():
        monkeypatch.setenv(k, v)
    ref, gs, kw, contigs = random_case(seed)
    if kw.get("threads", 1) < 2:
        kw["threads"] = 3
    rp, qs = write(str(tmp_path / "in"), ref, gs, contigs, seed)
    a = run(REFBIN, rp, qs, str(tmp_path / "ref"), kw)
    b = run(emu[1], rp, qs, str(tmp_path / "mine"), kw)
    assert a == b, (seed, kw, contigs)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(40))
def test_fuzz_resident_route_on_gpu(tmp_path, monkeypatch, seed):
    """the resident route's KERNELS (store_kernels.h: wave64 code the CPU suite only runs through its one-thread emulation) side
    by side with the reference binary: the product's sources with the test hooks compiled in, thresholds lowered so that the
    small sets take the route, every other seed with every anchor list let onto it (PM_FLAGGED_DIV=1)"""
    for k, v in dict(PM_DIRTY_MIN="2", PARSNP_PARALLEL_MIN="2", PARSNP_FREE_MIN="1", PM_FLAGGED_DIV="1" if seed % 2 else "8", PM_ATOMIC_MARKS="1" if seed % 4 == 0 else "0").items():
        monkeypatch.setenv(k, v)
    side_by_side(CORE_HOOKS_BIN, seed, tmp_path, big=seed >= 32)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(60))
def test_fuzz_on_gpu(tmp_path, seed):
    side_by_side(CORE_BIN, seed, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(10))
def test_fuzz_bigger_sets_on_gpu(tmp_path, seed):
    """0.2-0.6 Mb, 6-14 genomes, up to 11 structural edits per genome, 4-12 host threads: the threaded paths (parallel
    validation, generation-parallel replay with its hand-over and restart) against the reference's single-threaded result"""
    side_by_side(CORE_BIN, seed, tmp_path, big=True)
