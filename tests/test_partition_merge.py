"""Partition merge: the product's native merge (parsnp_amd/csrc/host/partition_merge.cpp behind include/parsnp_merge.h) and
the Python restatement of the reference driver's partition.py (oracle/partition_oracle.py = partition.py:35-61, 86-216,
245-433, 507-736 without Biopython / pyspoa; test infrastructure).  partition.py cannot be imported here (Bio, spoa absent),
so parity with the reference's own output stays UNPINNED; what is checked:
  * hand-worked vectors for the interval arithmetic, the trimming and the block merge (restatement), and the hand-checked
    files of tests/golden/partition/ (both implementations, byte for byte; README.md there derives every expected line);
  * native == restatement byte for byte (trimmed files and merged file) on partitions of seeded sets run through the host
    binary (CPU checker provider);
  * the properties the merge must have whatever aligns the insertion columns: every trimmed partition has the same
    reference pieces, every record spells its genome interval, the merged file holds every genome once with rows of equal
    length."""
import filecmp
import os
import shutil
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import partition_oracle as pmg  # noqa: E402
from parsnp_amd import merge as native_merge  # noqa: E402
from parsnp_amd import partition_run, synth  # noqa: E402

COMP = bytes.maketrans(b"ACGTN", b"TGCAN")


def test_interval_intersection_and_cut():
    A, B = [(0, 10), (20, 30), (40, 50)], [(5, 25), (28, 45)]
    assert pmg.interval_intersection(A, B) == [[5, 10], [20, 25], [28, 30], [40, 45]]
    assert pmg.interval_intersection(A, []) == [] and pmg.interval_intersection([(0, 5)], [(5, 9)]) == []   # touching is not overlapping
    iv = [(0, 10), (8, 20), (20, 30)]
    pmg.cut_overlaps(iv)
    assert iv == [(0, 10), (11, 20), (20, 30)]
    got = pmg.intersected_intervals([{1: [(0, 100), (200, 300)]}, {1: [(50, 250)]}, {1: [(0, 95), (205, 400)], 2: [(0, 50)]}], 10)
    assert got == {1: [[50, 95], [205, 250]], 2: []}
    assert pmg.intersected_intervals([{1: [(0, 100)]}, {1: [(95, 200)]}], 10) == {1: []}                    # shorter than 10: dropped


def rec(name, start, end, strand, ident, seq):
    return pmg.Rec(name, start, end, strand, ident, seq)


def test_trim_lcb_hand_worked():
    # reference bases 100..109 (p100), one insertion column after the 4th base; a forward and a reverse query
    ref = rec(1, 99, 109, 1, "cluster7 s1:p100", "ACGT-ACGTAC")
    fwd = rec(2, 499, 510, 1, "cluster7 s1:p500", "ACGTTACGTAC")
    rev = rec(3, 899, 908, -1, "cluster7 s2:p909", "AC-TTACGT-C")
    out = pmg.trim_lcb([ref, fwd, rev], {1: [(102, 107)]}, 1, 4)
    assert len(out) == 1
    r, f, v = out[0]
    assert (r.seq, f.seq, v.seq) == ("GT-ACG", "GTTACG", "-TTACG")          # reference bases 102..106 + the insertion column
    assert (r.start, r.end, r.id) == (101, 106, "cluster4 s1:p102")
    assert (f.start, f.end, f.id) == (501, 507, "cluster4 s1:p502")
    # reverse record: bases cut on the left come off the END coordinate and the p anchor, bases cut on the right off START
    assert (v.start, v.end, v.id) == (901, 906, "cluster4 s2:p907")          # 2 bases cut left ("AC"), 2 right ("T-C")
    two = pmg.trim_lcb([ref, fwd, rev], {1: [(100, 103), (105, 110)]}, 1, 1)
    assert [b[0].seq for b in two] == ["ACG", "CGTAC"] and [b[0].id for b in two] == ["cluster1 s1:p100", "cluster2 s1:p105"]
    assert pmg.trim_lcb([ref, fwd, rev], {2: [(0, 1000)]}, 1, 1) == []


def test_merge_blocks_columns_and_insertions():
    ref1 = rec(1, 0, 6, 1, "cluster1 s1:p1", "ACG-TAC"); q1 = rec(2, 10, 17, 1, "cluster1 s1:p11", "ACGGTAC")
    ref2 = rec(1, 0, 6, 1, "cluster1 s1:p1", "ACGT--AC"); q2 = rec(2, 20, 28, 1, "cluster1 s1:p21", "ACGTTTAC")
    fidx = {("a", 1): 1, ("a", 2): 2, ("b", 1): 1, ("b", 2): 3}
    padded = lambda seqs: [s + "-" * (max(map(len, seqs)) - len(s)) for s in seqs]   # noqa: E731
    merged = pmg.merge_blocks([([ref1, q1], "a"), ([ref2, q2], "b")], fidx, aligner=padded)
    assert [m.name for m in merged] == [1, 2, 3]
    assert len({len(m.seq) for m in merged}) == 1
    assert [m.seq.replace("-", "") for m in merged] == ["ACGTAC", "ACGGTAC", "ACGTTTAC"]
    # reference-anchored columns line up: the reference row's bases sit in the same columns as each query's aligned bases
    assert merged[0].seq == "ACG-T--AC" and merged[1].seq == "ACGGT--AC" and merged[2].seq == "ACG-TTTAC"


def test_hand_derived_fixture(tmp_path):
    """tests/golden/partition: two tiny partitions whose expected trimmed and merged files were written out by hand
    (README.md there); the native merge and the restatement must both reproduce them byte for byte"""
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "partition")
    for impl in ("native", "restatement"):
        d = tmp_path / impl
        d.mkdir()
        xs = [str(d / n) for n in ("p1.xmfa", "p2.xmfa")]
        for x in xs:
            shutil.copy(os.path.join(gold, os.path.basename(x)), x)
        out = str(d / "parsnp.xmfa")
        if impl == "native":
            m = native_merge.merge_partitions(xs, out, keep_trimmed=True, threads=2)
            assert (m["clusters"], m["sequences"], m["ref_bases"]) == (3, 4, 48)
        else:
            m = pmg.merge_partitions(xs, out)
            assert (m["clusters"], m["sequences"]) == (3, 4) and m["intervals"] == {1: [[6, 31], [35, 45], [48, 61]]}
        for got, want in ((out, "parsnp.xmfa"), (xs[0] + ".trimmed", "p1.xmfa.trimmed"), (xs[1] + ".trimmed", "p2.xmfa.trimmed")):
            assert open(got).read() == open(os.path.join(gold, "expected", want)).read(), (impl, want)
    # malformed input and disagreeing partitions are errors, not crashes
    bad = str(tmp_path / "bad.xmfa")
    open(bad, "w").write("#FormatVersion Mauve\n> 1:1-4 + nocluster\nACGT\n=\n")
    with pytest.raises(RuntimeError):
        native_merge.merge_partitions([bad], str(tmp_path / "o.xmfa"))
    with pytest.raises(RuntimeError):
        native_merge.merge_partitions([str(tmp_path / "missing.xmfa")], str(tmp_path / "o.xmfa"))


def spelled(genome: bytes, r):
    s = genome[r.start:r.end]
    return (s.translate(COMP)[::-1] if r.strand == -1 else s).decode()


def test_three_partitions_end_to_end(cpu_checkers, tmp_path):
    ref, gs = synth.make("pop6x200k")
    rp, qs = synth.write_set(str(tmp_path / "in"), ref, gs)
    res = partition_run.run_partitioned(cpu_checkers, rp, qs, str(tmp_path / "out"), 2, keep_trimmed=True)
    assert [p["queries"] for p in res["partitions"]] == [2, 2, 2] and all(p["ok"] for p in res["partitions"])
    m = res["merged"]
    # the native merge (what run_partitioned called) against the restatement: trimmed files and merged file, byte for byte
    xs = [os.path.join(p["dir"], "parsnpAligner.xmfa") for p in res["partitions"]]
    for x in xs:
        shutil.move(x + ".trimmed", x + ".trimmed.native")
    om = pmg.merge_partitions(xs, str(tmp_path / "oracle.xmfa"))
    assert (om["clusters"], om["sequences"]) == (m["clusters"], m["sequences"])
    assert all(filecmp.cmp(x + ".trimmed", x + ".trimmed.native", shallow=False) for x in xs)
    assert filecmp.cmp(str(tmp_path / "oracle.xmfa"), m["xmfa"], shallow=False)
    assert m["sequences"] == 7 and m["clusters"] > 20
    genomes = {"ref.fna": ref}
    genomes.update({os.path.basename(q): g for q, g in zip(qs, gs)})
    # every trimmed partition: the same reference pieces, records spell their genome intervals
    pieces = None
    for p in res["partitions"]:
        x = os.path.join(p["dir"], "parsnpAligner.xmfa")
        names = {i: f for i, f, _, _ in pmg.read_header(x)}
        mine = []
        for lcb in pmg.read_lcbs(x + ".trimmed"):
            assert len({len(r.seq) for r in lcb}) == 1
            mine.append(pmg.lcb_interval(lcb))
            for r in lcb:
                assert r.seq.replace("-", "").upper() == spelled(genomes[names[r.name]], r), (x, r.id)
        assert pieces is None or mine == pieces
        pieces = mine
    want = sorted(tuple(iv) for ivs in pmg.intersected_intervals([pmg.chunk_intervals(os.path.join(p["dir"], "parsnpAligner.xmfa")) for p in res["partitions"]]).values() for iv in ivs)
    assert sorted(iv for _, iv in pieces) == want and len(pieces) == m["clusters"]
    # the merged file: every genome once, rows of one length per block, every record spells its genome interval
    hdr = pmg.read_header(m["xmfa"])
    assert [h[0] for h in hdr] == list(range(1, 8)) and sorted(h[1] for h in hdr) == sorted(genomes)
    names = {i: f for i, f, _, _ in hdr}
    blocks = list(pmg.read_lcbs(m["xmfa"]))
    assert len(blocks) == m["clusters"]
    covered = 0
    for lcb in blocks:
        assert [r.name for r in lcb] == list(range(1, 8))
        assert len({len(r.seq) for r in lcb}) == 1
        for r in lcb:
            assert r.seq.replace("-", "").upper() == spelled(genomes[names[r.name]], r), r.id
        covered += lcb[0].end - lcb[0].start
    assert covered == sum(b - a for a, b in want)
    assert covered > 0.8 * len(ref)
