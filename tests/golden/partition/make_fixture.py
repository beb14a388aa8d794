"""Hand-derived fixture for partition mode's merge step.  partition.py itself cannot run in the build
container (Biopython / pyspoa absent), so the EXPECTED files are not produced by any merge code: this script spells out,
from the reference string R and the coordinates worked out in README.md, what partition.py:35-61, :99-216, :245-433 must
write for two tiny partitions.  The inputs hold no two sequences with an insertion at the same place, so the result does
not depend on SPOA (an insertion carried by ONE sequence is its own alignment).

    python tests/golden/partition/make_fixture.py      # rewrites p1.xmfa p2.xmfa + expected/*
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
R = "ACGTTGCAAGGCTTAACCGGATATCGCGTATTGACCAGTCAGGTTCACGATGCATCCGAA"          # reference, positions 1..60
assert len(R) == 60


def header(entries, nblocks):
    out = "#FormatVersion Mauve\n#SequenceCount %d\n" % len(entries)
    for i, (f, h, n) in enumerate(entries, 1):
        out += "##SequenceIndex %d\n##SequenceFile %s\n##SequenceHeader %s\n##SequenceLength %dbp\n" % (i, f, h, n)
    return out + "#IntervalCount %d\n" % nblocks


def rec(idx, a, b, strand, cluster, p, row):
    return "> %d:%d-%d %s cluster%d s1:p%d\n%s\n" % (idx, a, b, strand, cluster, p, row)


def ref(a, b):      # reference bases a..b, 1-based inclusive
    return R[a - 1:b]


REF, Q1, Q2, Q3 = ("ref.fna", ">ref", 60), ("q1.fna", ">q1", 400), ("q2.fna", ">q2", 400), ("q3.fna", ">q3", 600)

# ---- partition 1: reference + q1.  Block 1 = reference 1..30, q1 lacks reference base 12.  Block 2 = reference 35..60,
# q1 on the reverse strand (its p is the END of the record, partition.py:75-78).
q1_b1 = ref(1, 11) + "-" + ref(13, 30)
p1 = header([REF, Q1], 2)
p1 += rec(1, 1, 30, "+", 1, 1, ref(1, 30)) + rec(2, 101, 129, "+", 1, 101, q1_b1) + "=\n"
p1 += rec(1, 35, 60, "+", 2, 35, ref(35, 60)) + rec(2, 200, 225, "-", 2, 225, ref(35, 60)) + "=\n"

# ---- partition 2: reference + q2 + q3.  Block 1 = reference 6..44 with ONE insertion column after reference base 20
# (q2 carries a T there, q3 a gap).  Block 2 = reference 48..60.
ins = lambda c: ref(6, 20) + c + ref(21, 44)      # noqa: E731
p2 = header([REF, Q2, Q3], 2)
p2 += rec(1, 6, 44, "+", 1, 6, ins("-")) + rec(2, 301, 340, "+", 1, 301, ins("T")) + rec(3, 501, 539, "+", 1, 501, ins("-")) + "=\n"
p2 += rec(1, 48, 60, "+", 2, 48, ref(48, 60)) + rec(2, 345, 357, "+", 2, 345, ref(48, 60)) + rec(3, 543, 555, "+", 2, 543, ref(48, 60)) + "=\n"

# ---- expected (README.md derives every number).  Intervals: p1 (1,31) (35,61); p2 (6,45) (48,61);
# intersection (6,31) (35,45) (48,61) -- all at least 10 long.
t1 = header([REF, Q1], 2)        # (the header is copied as it is: copy_header, partition.py:218-229)
t1 += rec(1, 6, 30, "+", 1, 6, ref(6, 30)) + rec(2, 106, 129, "+", 1, 106, ref(6, 11) + "-" + ref(13, 30)) + "=\n"
t1 += rec(1, 35, 44, "+", 2, 35, ref(35, 44)) + rec(2, 216, 225, "-", 2, 225, ref(35, 44)) + "=\n"
t1 += rec(1, 48, 60, "+", 3, 48, ref(48, 60)) + rec(2, 200, 212, "-", 3, 212, ref(48, 60)) + "=\n"
cut = lambda c: ref(6, 20) + c + ref(21, 30)      # noqa: E731
t2 = header([REF, Q2, Q3], 2)
t2 += rec(1, 6, 30, "+", 1, 6, cut("-")) + rec(2, 301, 326, "+", 1, 301, cut("T")) + rec(3, 501, 525, "+", 1, 501, cut("-")) + "=\n"
t2 += rec(1, 35, 44, "+", 2, 35, ref(35, 44)) + rec(2, 331, 340, "+", 2, 331, ref(35, 44)) + rec(3, 530, 539, "+", 2, 530, ref(35, 44)) + "=\n"
t2 += rec(1, 48, 60, "+", 3, 48, ref(48, 60)) + rec(2, 345, 357, "+", 3, 345, ref(48, 60)) + rec(3, 543, 555, "+", 3, 543, ref(48, 60)) + "=\n"
m = header([REF, Q1, Q2, Q3], 3)
m += (rec(1, 6, 30, "+", 1, 6, cut("-")) + rec(2, 106, 129, "+", 1, 106, ref(6, 11) + "-" + ref(13, 20) + "-" + ref(21, 30))
      + rec(3, 301, 326, "+", 1, 301, cut("T")) + rec(4, 501, 525, "+", 1, 501, cut("-")) + "=\n")
m += rec(1, 35, 44, "+", 2, 35, ref(35, 44)) + rec(2, 216, 225, "-", 2, 225, ref(35, 44)) + rec(3, 331, 340, "+", 2, 331, ref(35, 44)) + rec(4, 530, 539, "+", 2, 530, ref(35, 44)) + "=\n"
m += rec(1, 48, 60, "+", 3, 48, ref(48, 60)) + rec(2, 200, 212, "-", 3, 212, ref(48, 60)) + rec(3, 345, 357, "+", 3, 345, ref(48, 60)) + rec(4, 543, 555, "+", 3, 543, ref(48, 60)) + "=\n"

if __name__ == "__main__":
    os.makedirs(os.path.join(HERE, "expected"), exist_ok=True)
    for name, text in (("p1.xmfa", p1), ("p2.xmfa", p2), ("expected/p1.xmfa.trimmed", t1), ("expected/p2.xmfa.trimmed", t2), ("expected/parsnp.xmfa", m)):
        open(os.path.join(HERE, name), "w").write(text)
