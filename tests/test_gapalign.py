"""Inter-MUM gap aligner (parsnp_amd/csrc/host/gapalign.cpp) against the reference's MUSCLE call
(src/parsnp.cpp:854-855 -> src/MuscleInterface.cpp:37-78 -> libMUSCLE 3.7): committed vectors produced by the reference
(tests/golden/gapalign.json, made by tests/golden/make_gapalign_golden.py) and, where oracle/_ref/muscle_ref exists, fresh
seeded sets.  The bar is identical rows."""
import ctypes
import json
import os
import subprocess

import pytest

import gapgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "parsnp_amd", "csrc", "host", "gapalign.cpp")
LIB = os.path.join(ROOT, "tests", "emu", "libgapalign.so")


@pytest.fixture(scope="module")
def aligner():
    hdr = SRC[:-4] + ".h"
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O3", "-mavx2", "-std=c++17", "-Wall", "-shared", "-fPIC", SRC, "-o", LIB], check=True)
    lib = ctypes.CDLL(LIB)
    lib.parsnp_gap_align.restype = ctypes.c_long
    lib.parsnp_gap_align.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_long]

    def run(seqs):
        cap = (1 << 16) + 4 * sum(len(s) for s in seqs) * (len(seqs) + 1)
        buf = ctypes.create_string_buffer(cap)
        r = lib.parsnp_gap_align("\n".join(seqs).encode(), buf, cap)
        assert r >= 0, r
        return buf.value.decode().split("\n")[:-1]
    return run


def test_committed_vectors(aligner):
    data = json.load(open(os.path.join(ROOT, "tests", "golden", "gapalign.json")))
    assert len(data) > 200
    for blk in data:
        assert aligner(blk["in"]) == blk["out"], blk["in"]


def test_rows_are_the_inputs_with_gaps(aligner):
    """size-independent property: stripping '-' gives back the input (after the reference's letter fix-up), equal row lengths"""
    for blk in gapgen.blocks(5, 200):
        rows = aligner(blk)
        assert len({len(r) for r in rows}) == 1
        assert [r.replace("-", "") for r in rows] == blk


def test_declines_bad_input(aligner):
    lib = ctypes.CDLL(LIB)
    lib.parsnp_gap_align.restype = ctypes.c_long
    buf = ctypes.create_string_buffer(64)
    assert lib.parsnp_gap_align(b"ACGT", buf, ctypes.c_long(64)) == -1        # one sequence: the reference never aligns it
    assert lib.parsnp_gap_align(b"ACGT\n\nAC", buf, ctypes.c_long(64)) == -1   # an empty sequence


@pytest.mark.skipif(not os.path.exists(gapgen.MUSCLE_REF), reason="oracle/_ref/muscle_ref not built (needs /root/reference)")
@pytest.mark.parametrize("seed", [11, 12, 13])
def test_fresh_sets_against_reference(aligner, seed):
    blks = gapgen.blocks(seed, 250)
    for blk, want in zip(blks, gapgen.reference_align(blks)):
        assert aligner(blk) == want, blk
