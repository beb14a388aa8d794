"""Host side of the parsnp_core replacement (ini, ingest, validation, recursive extension, LCB chaining, XMFA/log)
checked end to end against the reference binary's committed goldens, with the CPU checker standing in for the GPU
engine behind the C ABI (oracle/_ref/parsnp_core_oracle -- test build only)."""
import json
import os

import pytest

import xmfa_util
from parsnp_amd import driver, synth
from test_golden import mers

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
E2E = json.load(open(os.path.join(G, "e2e.json")))


def check(core, name, rp, qs, out, env=None):
    rc, _ = driver.run_core(core, rp, qs, out, env=env)
    assert rc == 0, open(os.path.join(out, "parsnp-aligner.err")).read()[-2000:]
    x = os.path.join(out, "parsnpAligner.xmfa")
    want = E2E[name]
    assert xmfa_util.mum_lcb_signature(x) == want["signature"]
    assert xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")) == want["log"]
    assert xmfa_util.md5(x) == want["xmfa_md5"]   # every byte, inter-MUM gap columns (the reference's MUSCLE call) included
    assert "NOTE" not in open(os.path.join(out, "parsnpAligner.log")).read()
    assert os.path.exists(os.path.join(out, "allmums.out"))


def test_mers(cpu_checkers, tmp_path):
    ref, qs = mers(base=str(tmp_path))
    check(cpu_checkers, "mers", ref, qs, str(tmp_path / "out"))


@pytest.mark.parametrize("name", ["viral50", "pop6x200k", "rearr6x300k"])
def test_synthetic(cpu_checkers, tmp_path, name):
    r, gs = synth.make(name)
    rp, qs = synth.write_set(str(tmp_path / "in"), r, gs)
    check(cpu_checkers, name, rp, qs, str(tmp_path / "out"))


@pytest.mark.parametrize("name", ["pop6x200k", "rearr6x300k", "poprearr10x400k", "popinv12x400k"])
def test_xmfa_self_consistency(cpu_checkers, tmp_path, name):
    """the size-independent XMFA properties used at full size on the GPU (tests/xmfa_util.consistency), here on sets whose
    bytes are also pinned by goldens: equal row lengths, clean MUM columns, every record spells its genome interval"""
    r, gs = synth.make(name)
    rp, qs = synth.write_set(str(tmp_path / "in"), r, gs)
    rc, _ = driver.run_core(cpu_checkers, rp, qs, str(tmp_path / "out"))
    assert rc == 0
    st = xmfa_util.consistency(str(tmp_path / "out" / "parsnpAligner.xmfa"), [r] + gs)
    assert st["lcbs"] > 10 and st["bad_length"] == 0 and st["bad_mum_column"] == 0 and st["bad_sequence"] == 0, st


def harsh_inputs(name, base):
    if name in ("pop6x200k", "rearr6x300k"):
        r, gs = synth.make(name)
        return synth.write_set(os.path.join(base, "in"), r, gs) + ({},)
    if name == "messy":
        return synth.messy_set(os.path.join(base, "in")) + ({},)
    if name == "draft8x300k":
        return synth.draft_set(os.path.join(base, "in"), n=300_000, n_genomes=8, contigs=60) + ({},)
    if name == "draft20x1m":
        return synth.draft_set(os.path.join(base, "in"), n=1_000_000, n_genomes=20, contigs=300) + ({},)
    if name == "pchunk":
        r, gs = synth.make("pop6x200k")
        return synth.write_set(os.path.join(base, "in"), r, gs) + (dict(partpos=66660),)
    r, gs = synth.make(name)
    return synth.write_set(os.path.join(base, "in"), r, gs) + ({},)


@pytest.mark.parametrize("name", ["poprearr10x400k", "messy", "pchunk", "draft8x300k"])
def test_harsh_inputs(cpu_checkers, tmp_path, name):
    """rearranged 5 %-divergent population (asymmetric regions, reverse LCBs); multi-contig / IUPAC / CRLF / lower-case
    FASTA; reference longer than the chunk size p (3 chunks + the <50 bp tail rule); draft assemblies (60 shuffled,
    half reverse-complemented contigs per genome, joined by N runs)"""
    rp, qs, kw = harsh_inputs(name, str(tmp_path))
    out = str(tmp_path / "out")
    rc, _ = driver.run_core(cpu_checkers, rp, qs, out, **kw)
    assert rc == 0
    want = E2E[name]
    assert xmfa_util.mum_lcb_signature(os.path.join(out, "parsnpAligner.xmfa")) == want["signature"]
    assert xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")) == want["log"]
    assert xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")) == want["xmfa_md5"]


def test_no_speculation_same_result(cpu_checkers, tmp_path):
    """the batched speculative sweep must not change the result of the in-order replay"""
    r, gs = synth.make("pop6x200k")
    rp, qs = synth.write_set(str(tmp_path / "in"), r, gs)
    env = dict(os.environ, PARSNP_NO_SPECULATION="1")
    check(cpu_checkers, "pop6x200k", rp, qs, str(tmp_path / "out"), env=env)


@pytest.mark.parametrize("name,threads", [("pop6x200k", 1), ("pop6x200k", 6), ("messy", 3), ("draft8x300k", 4), ("pchunk", 2)])
def test_in_order_replay_same_result(cpu_checkers, tmp_path, name, threads):
    """the recursion is normally replayed generation by generation (clusters of overlapping regions in parallel) and falls
    back to the reference's in-order replay when a generation is not disjoint; PARSNP_SEQUENTIAL_REPLAY=1 forces the
    in-order replay from the start.  Both must give the reference's bytes, with one thread and with several."""
    rp, qs, kw = harsh_inputs(name, str(tmp_path))
    for tag, extra in (("gen", {}), ("seq", {"PARSNP_SEQUENTIAL_REPLAY": "1"})):
        out = str(tmp_path / tag)
        rc, _ = driver.run_core(cpu_checkers, rp, qs, out, env=dict(os.environ, **extra), threads=threads, **kw)
        assert rc == 0
        assert xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")) == E2E[name]["xmfa_md5"], tag
        assert xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")) == E2E[name]["log"], tag


@pytest.mark.parametrize("switch", ["PARSNP_ORDERED_FLAGGED", "PARSNP_CHAIN_TWICE", "PARSNP_EXACT_OVERLAP"])
@pytest.mark.parametrize("name", ["poprearr10x400k", "draft8x300k"])
def test_plain_variants_of_the_host_shortcuts(cpu_checkers, tmp_path, name, switch):
    """every host shortcut has a switch that takes the plain route instead -- flagged candidates all in candidate order,
    the second chaining pass always run, the overlap flags from scratch bitmaps -- and the bytes must not change (the default
    route is pinned by the goldens in the other tests)"""
    rp, qs, kw = harsh_inputs(name, str(tmp_path))
    out = str(tmp_path / "out")
    env = dict(os.environ, PARSNP_PARALLEL_MIN="8", PARSNP_FREE_MIN="2")   # the shortcuts at work on small sets
    env[switch] = "1"
    rc, _ = driver.run_core(cpu_checkers, rp, qs, out, env=env, threads=4, **kw)
    assert rc == 0
    assert xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")) == E2E[name]["xmfa_md5"]
    assert xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")) == E2E[name]["log"]


@pytest.mark.parametrize("route", ["walks", "rows"])
@pytest.mark.parametrize("name", ["pop6x200k", "poprearr10x400k", "messy", "draft8x300k"])
def test_derived_left_neighbours_equal_the_walk(cpu_checkers, tmp_path, name, route):
    """the seed regions left of an anchor are derived from the walk right of the previous anchor where that walk ended
    at this anchor in every genome ("walks"); where the threaded validation found the accepted anchors in list order in
    every genome, both regions of every anchor are derived from the rows of its list neighbours without reading a
    bitmap ("rows": PARSNP_PARALLEL_MIN sends the small sets through that validation).  PARSNP_CHECK_NEIGHBOURS=1 repeats
    every derived region with the bitmap walk of determineRegion (parsnp.cpp:1199-1290) and aborts on a difference"""
    rp, qs, kw = harsh_inputs(name, str(tmp_path))
    out = str(tmp_path / "out")
    env = dict(os.environ, PARSNP_CHECK_NEIGHBOURS="1")
    if route == "rows":
        env.update(PARSNP_PARALLEL_MIN="8", PARSNP_FREE_MIN="2", PARSNP_DEBUG_TIMERS="1")
    rc, _ = driver.run_core(cpu_checkers, rp, qs, out, env=env, threads=3, **kw)
    assert rc == 0, open(os.path.join(out, "parsnp-aligner.err")).read()[-2000:]
    assert xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")) == E2E[name]["xmfa_md5"]
    if route == "rows" and name == "pop6x200k":     # a collinear set: the row route must actually have been taken
        assert "seeds from rows" in open(os.path.join(out, "parsnp-aligner.err")).read()


@pytest.mark.parametrize("name", ["pop6x200k", "poprearr10x400k"])
def test_literal_worklist_same_result(cpu_checkers, tmp_path, name):
    """the map-based work list (unique keys) and the reference's literal vector + std::sort + adjacent-dedup agree"""
    rp, qs, kw = harsh_inputs(name, str(tmp_path))
    out = str(tmp_path / "out")
    env = dict(os.environ, PARSNP_FORCE_LITERAL_WORKLIST="1")
    rc, _ = driver.run_core(cpu_checkers, rp, qs, out, env=env, **kw)
    assert rc == 0
    assert xmfa_util.mum_lcb_signature(os.path.join(out, "parsnpAligner.xmfa")) == E2E[name]["signature"]
    assert xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")) == E2E[name]["xmfa_md5"]
    assert xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")) == E2E[name]["log"]


@pytest.mark.parametrize("name,par_min", [("poprearr10x400k", None), ("rearr6x300k", None), ("pop6x200k", "8"), ("messy", "2"), ("pchunk", "8")])
def test_threaded_validation_same_result(cpu_checkers, tmp_path, name, par_min):
    """cores > 1: long candidate lists are validated with the clean/dirty parallel scheme; the result must not change.
    PARSNP_PARALLEL_MIN lowers the list-length threshold so that small sets (and the recursion's short lists) use it too."""
    rp, qs, kw = harsh_inputs(name, str(tmp_path))
    out = str(tmp_path / "out")
    env = dict(os.environ, PARSNP_FREE_MIN="2")     # flagged candidates that meet no other flagged one are settled in parallel
    if par_min:
        env["PARSNP_PARALLEL_MIN"] = par_min
    rc, _ = driver.run_core(cpu_checkers, rp, qs, out, env=env, threads=4, timing=str(tmp_path / "t.json"), **kw)
    assert rc == 0
    assert xmfa_util.mum_lcb_signature(os.path.join(out, "parsnpAligner.xmfa")) == E2E[name]["signature"]
    assert xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")) == E2E[name]["xmfa_md5"]
    assert xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")) == E2E[name]["log"]


@pytest.mark.parametrize("plain", [False, True, "mix", "pwrite", "groups", "groups_pwrite", "groups_grow"])
@pytest.mark.parametrize("name", ["mers", "rearr6x300k", "poprearr10x400k", "messy", "draft8x300k"])
def test_streamed_and_plain_records_agree(cpu_checkers, tmp_path, name, plain):
    """the XMFA records are streamed into their places in the file from the MUM table and the gap alignments (sizes
    first); an LCB that has to be trimmed against its predecessor, and everything under PARSNP_PLAIN_OUTPUT=1, is built
    as strings the way the reference does it (parsnp.cpp:663-1071).  Both routes, several threads: the golden bytes."""
    if name == "mers":
        rp, qs = mers(base=str(tmp_path)); kw = {}
    else:
        rp, qs, kw = harsh_inputs(name, str(tmp_path))
    out = str(tmp_path / "out")
    env = dict(os.environ, PARSNP_DEBUG_TIMERS="1")
    if plain == "mix":
        env["PARSNP_OUTPUT_MIX"] = "3"     # every third LCB is turned into strings late, as one that needs the trim is
    elif plain == "pwrite":
        env["PARSNP_OUTPUT_PWRITE"] = "1"  # positioned writes instead of the shared mapping (a file system that cannot reserve)
    elif plain in ("groups", "groups_pwrite", "groups_grow"):
        env["PARSNP_GAP_GROUPS"] = "3"     # the LCBs in three groups, each laid out and written as soon as its gaps are aligned
        if plain == "groups_pwrite":
            env["PARSNP_OUTPUT_PWRITE"] = "1"
        if plain == "groups_grow":
            env["PARSNP_RESERVE_TINY"] = "1"   # too small a reservation: the mapping grows group by group
    elif plain:
        env["PARSNP_PLAIN_OUTPUT"] = "1"
    rc, _ = driver.run_core(cpu_checkers, rp, qs, out, env=env, threads=3, **kw)
    assert rc == 0, open(os.path.join(out, "parsnp-aligner.err")).read()[-2000:]
    assert xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")) == E2E[name]["xmfa_md5"]
    import re
    m = re.search(r"\[output\] (\d+) LCBs printed: (\d+) streamed, (\d+) as strings", open(os.path.join(out, "parsnp-aligner.err")).read())
    assert m and int(m.group(1)) > 0
    assert (int(m.group(2)) == 0) if plain is True else (int(m.group(2)) > 0)
    if plain == "mix":
        assert int(m.group(3)) > 0
    err = open(os.path.join(out, "parsnp-aligner.err")).read()
    assert ("through positioned writes" in err) if plain in (True, "pwrite", "groups_pwrite") else ("through a shared mapping" in err)
    if plain in ("groups", "groups_pwrite", "groups_grow"):
        assert "in 3 group(s)" in err


@pytest.mark.parametrize("name", ["pop6x200k", "poprearr10x400k", "messy", "pchunk", "draft8x300k"])
def test_threaded_validation_of_short_lists(cpu_checkers, tmp_path, name):
    """the threaded validation of a candidate list (clean candidates settled and marked by genome stripes, flagged ones against
    those marks -- side by side where they meet no other flagged candidate, in list order where they do) on lists far shorter
    than its production threshold: the reference's bytes"""
    rp, qs, kw = harsh_inputs(name, str(tmp_path))
    out = str(tmp_path / "out")
    env = dict(os.environ, PARSNP_PARALLEL_MIN="8", PARSNP_FREE_MIN="2")
    rc, _ = driver.run_core(cpu_checkers, rp, qs, out, env=env, threads=4, **kw)
    assert rc == 0, open(os.path.join(out, "parsnp-aligner.err")).read()[-2000:]
    assert xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")) == E2E[name]["xmfa_md5"]
    assert xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")) == E2E[name]["log"]


MUMI = json.load(open(os.path.join(G, "mumi.json")))


def mumi_inputs(name, base):
    if name == "mers":
        return mers(base=base) + ({},)
    if name == "messy":
        return synth.messy_set(os.path.join(base, "in")) + ({},)
    if name == "draft8x300k":
        return synth.draft_set(os.path.join(base, "in"), n=300_000, n_genomes=8, contigs=60) + ({},)
    if name == "draft20x1m":
        return synth.draft_set(os.path.join(base, "in"), n=1_000_000, n_genomes=20, contigs=300) + ({},)
    r, gs = synth.make("pop6x200k")
    return synth.write_set(os.path.join(base, "in"), r, gs) + (dict(partpos=40000),)


def check_mumi(core, name, base):
    """calcmumi=1: <outdir>/all.mumi "idx:dist" lines identical to the reference binary's (order is unspecified there)"""
    rp, qs, kw = mumi_inputs(name, base)
    out = os.path.join(base, "out")
    rc, _ = driver.run_core(core, rp, qs, out, calcmumi=1, **kw)
    assert rc == 0
    lines = sorted(open(os.path.join(out, "all.mumi")).read().split(), key=lambda x: int(x.split(":")[0]))
    assert lines == MUMI[name]
    assert not os.path.exists(os.path.join(out, "parsnpAligner.xmfa"))


@pytest.mark.parametrize("name", ["mers", "messy", "pop6x200k_p"])
def test_calcmumi(cpu_checkers, tmp_path, name):
    check_mumi(cpu_checkers, name, str(tmp_path))


def test_cli_surface(cpu_checkers, tmp_path):
    import subprocess
    assert subprocess.run([cpu_checkers, "-v"], capture_output=True, text=True).stdout.strip() == "Parsnp v1.0.1"
    assert "parameter file" in subprocess.run([cpu_checkers, "-h"], capture_output=True, text=True).stdout
    assert subprocess.run([cpu_checkers], capture_output=True).returncode == 1
    # missing reference file -> exit(1) with the reference's message
    ini = tmp_path / "x.ini"
    ini.write_text(driver.ini_text("/nonexistent/ref.fna", [], str(tmp_path)))
    p = subprocess.run([cpu_checkers, str(ini)], capture_output=True, text=True)
    assert p.returncode == 1 and "Cannot open reference file" in p.stdout


def test_no_mums_found(cpu_checkers, tmp_path):
    """unrelated genomes: rc 0, 'NO MUMS FOUND' in the log, no XMFA (src/parsnp.cpp:3223-3229)"""
    import numpy as np
    rng = np.random.default_rng(1)
    ref = synth.random_genome(rng, 5000).tobytes(); q = synth.random_genome(rng, 5000).tobytes()
    rp, qs = synth.write_set(str(tmp_path / "in"), ref, [q])
    out = str(tmp_path / "out")
    rc, _ = driver.run_core(cpu_checkers, rp, qs, out)
    assert rc == 0
    assert not os.path.exists(os.path.join(out, "parsnpAligner.xmfa"))
    assert open(os.path.join(out, "parsnpAligner.log")).read().strip() == "NO MUMS FOUND"
