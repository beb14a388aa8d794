"""The inter-MUM gap aligner ON THE DEVICE (pm_gap_align_batch in parsnp_amd/lib/libparsnp_hip.so, one wavefront per gap)
against the reference's MUSCLE call: the committed vectors produced by the reference (tests/golden/gapalign.json), the
host restatement (parsnp_amd/csrc/host/gapalign.cpp, itself pinned against libMUSCLE) on fresh seeded sets including
200-sequence gaps, and -- where it ships -- the reference's own MuscleInterface (oracle/_ref/muscle_ref).
The bar is identical rows; jobs the device declines (cols = -1) must be exactly the ones outside its documented limits."""
import ctypes as C
import json
import os
import random

import numpy as np
import pytest

import gapgen
from parsnp_amd.paths import HIP_LIB
from test_gapalign import aligner  # noqa: F401  (fixture: the host restatement)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEVICE_COLS = 96


@pytest.fixture(scope="module")
def device():
    lib = C.CDLL(HIP_LIB)
    lib.pm_gap_align_batch.restype = C.c_int
    lib.pm_gap_last_error.restype = C.c_char_p

    def run(blocks, slack=None):
        """-> per block: list of rows, or None where the device declined"""
        nseq = np.array([len(b) for b in blocks], np.int32)
        flat = [s.encode() for b in blocks for s in b]
        off = np.zeros(len(flat) + 1, np.int64)
        off[1:] = np.cumsum([len(s) for s in flat])
        chars = np.frombuffer(b"".join(flat) or b"\0", np.uint8).copy()
        maxc = np.array([min(DEVICE_COLS, (max(len(s) for s in b) * 3) // 2 + 16) if slack is None else slack for b in blocks], np.int32)
        row_off = np.zeros(len(blocks), np.int64)
        row_off[1:] = np.cumsum(nseq[:-1].astype(np.int64) * maxc[:-1])
        out = np.zeros(int((nseq.astype(np.int64) * maxc).sum()) + 1, np.uint8)
        cols = np.full(len(blocks), -7, np.int32)
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))   # noqa: E731
        rc = lib.pm_gap_align_batch(C.c_int(-1), C.c_int64(len(blocks)), p(nseq, C.c_int32), p(off, C.c_int64), p(chars, C.c_uint8),
                                    p(maxc, C.c_int32), p(row_off, C.c_int64), p(out, C.c_uint8), C.c_int64(len(out)), p(cols, C.c_int32))
        assert rc == 0, lib.pm_gap_last_error()
        res = []
        for j, b in enumerate(blocks):
            if cols[j] < 0:
                res.append(None)
                continue
            base = int(row_off[j])
            res.append([out[base + i * int(maxc[j]): base + i * int(maxc[j]) + int(cols[j])].tobytes().decode() for i in range(len(b))])
        return res
    return run


def fits(block):
    # the device's documented limits (include/parsnp_mum.h): widths, and an alphabet its one-byte row coding round-trips --
    # upper-case letters without 'U' (parsnp's ingest emits ACGTN only); the caller's host path takes everything else
    return len(block) >= 2 and all(0 < len(s) <= DEVICE_COLS for s in block) and not any(ch.islower() or ch == "U" for s in block for ch in s)


def test_committed_vectors(device):
    data = json.load(open(os.path.join(ROOT, "tests", "golden", "gapalign.json")))
    got = device([blk["in"] for blk in data])
    done = 0
    for blk, rows in zip(data, got):
        if rows is None:      # declined: only what is outside the limits (a sequence or an alignment wider than 96 columns)
            assert not fits(blk["in"]) or max(len(r) for r in blk["out"]) > (max(len(s) for s in blk["in"]) * 3) // 2 + 16 or max(len(r) for r in blk["out"]) > DEVICE_COLS, blk["in"]
            continue
        assert rows == blk["out"], blk["in"]
        done += 1
    assert done > 200


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_fresh_sets_against_host_restatement(device, aligner, seed):  # noqa: F811
    blks = gapgen.blocks(seed, 400, lengths=(1, 2, 3, 4, 5, 6, 7, 10, 15, 30, 60, 120))
    got = device(blks)
    done = 0
    for blk, rows in zip(blks, got):
        want = aligner(blk)
        if rows is None:
            assert not fits(blk) or len(want[0]) > min(DEVICE_COLS, (max(len(s) for s in blk) * 3) // 2 + 16), blk
            continue
        assert rows == want, blk
        done += 1
    assert done > 300


def test_many_sequences_per_gap(device, aligner):  # noqa: F811
    """gaps as the headline workload has them: 201 sequences that are copies of a few alleles, and 201 all different"""
    rng = random.Random(5)
    blks = []
    for k in range(60):
        L = rng.choice([2, 3, 5, 8, 13, 21, 40, 60])
        base = "".join(rng.choice("ACGT") for _ in range(L))
        alleles = [gapgen.mutate(rng, base, rng.choice([0.05, 0.2, 0.5])) for _ in range(rng.choice([2, 3, 5, 9]))]
        if k % 3 == 0:
            blks.append([gapgen.mutate(rng, base, 0.15) for _ in range(201)])
        else:
            blks.append([rng.choice(alleles) for _ in range(201)])
    blks.append(["ACGTN"[i % 5] * (1 + i % 7) for i in range(512)])      # the most sequences the device takes
    got = device(blks)
    for blk, rows in zip(blks, got):
        assert rows is not None, blk[:3]
        assert rows == aligner(blk), blk[:3]


def test_declines_and_limits(device):
    got = device([["ACGT", "ACG"], ["A" * 97, "A" * 60], ["ACGT"] * 513, ["ACGTACGTAA", "TTTTTTTTTT"]], slack=None)
    assert got[0] is not None and got[3] is not None
    assert got[1] is None and got[2] is None          # wider than 96 columns / more than 512 sequences
    # a row capacity smaller than the alignment: declined, not overrun
    tight = device([["ACGTACGTAA", "TTTTTTTTTTAC"]], slack=12)
    assert tight[0] is None or len(tight[0][0]) <= 12


@pytest.mark.skipif(not os.path.exists(gapgen.MUSCLE_REF), reason="oracle/_ref/muscle_ref not shipped")
def test_fresh_sets_against_reference(device):
    blks = [b for b in gapgen.blocks(31, 300, lengths=(1, 2, 3, 5, 7, 10, 15, 30, 60, 120)) if fits(b)]
    got = device(blks)
    for blk, want, rows in zip(blks, gapgen.reference_align(blks), got):
        if rows is not None:
            assert rows == want, blk


def test_groups_report_in_order_and_match_the_batch(device):
    """pm_gap_align_groups: the same jobs in four groups of consecutive jobs -- one with no job the device takes -- give
    the rows of the single batch, and `done` hears of every group, in order, with the group's rows already in place"""
    blks = gapgen.blocks(41, 240, lengths=(1, 2, 3, 5, 8, 13, 30, 60))
    blks[100:100] = [["A" * 97, "A" * 60]] * 3                      # a group made of declined jobs only
    want = device(blks)
    lib = C.CDLL(HIP_LIB)
    lib.pm_gap_align_groups.restype = C.c_int
    nseq = np.array([len(b) for b in blks], np.int32)
    flat = [s.encode() for b in blks for s in b]
    off = np.zeros(len(flat) + 1, np.int64)
    off[1:] = np.cumsum([len(s) for s in flat])
    chars = np.frombuffer(b"".join(flat), np.uint8).copy()
    maxc = np.array([min(DEVICE_COLS, (max(len(s) for s in b) * 3) // 2 + 16) for b in blks], np.int32)
    row_off = np.zeros(len(blks), np.int64)
    row_off[1:] = np.cumsum(nseq[:-1].astype(np.int64) * maxc[:-1])
    out = np.zeros(int((nseq.astype(np.int64) * maxc).sum()) + 1, np.uint8)
    cols = np.full(len(blks), -7, np.int32)
    group_end = np.array([100, 103, 180, len(blks)], np.int64)
    seen = []

    def rows_of(j):
        if cols[j] < 0:
            return None
        base = int(row_off[j])
        return [out[base + i * int(maxc[j]): base + i * int(maxc[j]) + int(cols[j])].tobytes().decode() for i in range(len(blks[j]))]

    CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int)

    def done(ctx, g):
        lo = 0 if g == 0 else int(group_end[g - 1])
        seen.append((g, [rows_of(j) for j in range(lo, int(group_end[g]))] == want[lo:int(group_end[g])]))

    cb = CB(done)
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))   # noqa: E731
    rc = lib.pm_gap_align_groups(C.c_int(-1), C.c_int64(len(blks)), p(nseq, C.c_int32), p(off, C.c_int64), p(chars, C.c_uint8), p(maxc, C.c_int32),
                                 p(row_off, C.c_int64), p(out, C.c_uint8), C.c_int64(len(out)), p(cols, C.c_int32), C.c_int(4), p(group_end, C.c_int64), cb, None)
    assert rc == 0
    assert seen == [(0, True), (1, True), (2, True), (3, True)]
    assert [rows_of(j) for j in range(len(blks))] == want
    assert all(w is None for w in want[100:103]) and sum(w is not None for w in want) > 200


def test_marker_free_build():
    """The gap kernel WITHOUT its (job, stage) markers and stage clocks (parsnp_amd/lib/exp/libparsnp_hip_no_MARKERS.so, built by
    csrc/Makefile: -DPM_GAP_NO_MARKERS).  Until round 5 such a build hung on the device on the first batch of more than one job: with
    align_job inlined into the kernel's job loop the compiler produced an exec-mask loop that only lane 0 ever left, and the markers'
    branches happened to break it up (gapalign_hip.hip: align_job).  The committed vectors, fresh seeded sets and a
    batch of many small jobs through that build, in a child process under a watchdog: a hang is a red test, not a stuck suite."""
    import subprocess
    import sys
    lib = os.path.join(ROOT, "parsnp_amd", "lib", "exp", "libparsnp_hip_no_MARKERS.so")
    assert os.path.exists(lib), "the marker-free build is part of `make -C parsnp_amd/csrc` (python -c 'import __graft_entry__ as g; g.build()')"
    env = dict(os.environ, PARSNP_HIP_LIB=lib)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k",
                        "committed_vectors or fresh_sets_against_host or many_sequences or declines or groups_report"],
                       capture_output=True, text=True, env=env, timeout=240, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-1000:]
    # growing batches -- 50 small jobs was where the inlined shape hung --, every step under the probe's own 20-s watchdog
    q = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gap_probe.py")], capture_output=True, text=True, env=env, timeout=200, cwd=ROOT)
    assert q.returncode == 0 and "STUCK" not in q.stdout and "2000 x 201 alleles" in q.stdout, q.stdout[-1500:] + q.stderr[-1500:]
