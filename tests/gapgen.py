"""Seeded inputs for the gap-aligner parity tests: sets of diverged copies of a short sequence, the shape of the gaps
between adjacent MUMs that the XMFA writer aligns (reference: src/parsnp.cpp:790-865)."""
import os
import random
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MUSCLE_REF = os.path.join(ROOT, "oracle", "_ref", "muscle_ref")


def mutate(rng, s, rate, alpha="ACGT"):
    out = []
    for ch in s:
        r = rng.random()
        if r < rate / 3:
            continue
        if r < 2 * rate / 3:
            out.append(rng.choice(alpha))
        elif r < rate:
            out.append(ch)
            out.append(rng.choice(alpha))
        else:
            out.append(ch)
    return "".join(out) or rng.choice(alpha)


def block(rng, sizes=(2, 2, 3, 3, 4, 5, 6, 8, 12, 20, 40), lengths=(1, 2, 3, 4, 5, 6, 7, 10, 15, 30, 60, 120, 300)):
    n = rng.choice(sizes)
    length = rng.choice(lengths)
    alpha = "ACGT" if rng.random() < 0.7 else "ACGTN"
    base = "".join(rng.choice(alpha) for _ in range(length))
    rate = rng.choice([0.02, 0.1, 0.3, 0.6])
    mode = rng.random()
    seqs = []
    for i in range(n):
        if mode < 0.5:
            seqs.append(mutate(rng, base, rate, alpha))
        elif mode < 0.8:   # a few haplotypes shared by many genomes
            seqs.append(mutate(rng, base, rate, alpha) if i < 3 else seqs[rng.randrange(3)])
        else:              # unrelated
            seqs.append("".join(rng.choice(alpha) for _ in range(rng.randint(1, max(1, length)))))
    if rng.random() < 0.1:
        seqs[rng.randrange(n)] = "N"   # what the reference substitutes for an empty gap string (:806-808)
    return seqs


def blocks(seed, count, **kw):
    rng = random.Random(seed)
    return [block(rng, **kw) for _ in range(count)]


def golden_blocks():
    out = blocks(20250927, 220, lengths=(1, 2, 3, 4, 5, 6, 7, 10, 15, 30, 60))
    rng = random.Random(7)
    base = "".join(rng.choice("ACGT") for _ in range(40))
    out.append([mutate(rng, base, 0.05) for _ in range(200)])          # as many genomes as the headline workload
    base = "".join(rng.choice("ACGT") for _ in range(700))
    out.append([mutate(rng, base, 0.02) for _ in range(12)])           # long gap
    out.append(["A" * 700, "A" * 300 + "C" + "A" * 290, "A" * 255 + "G", "AAAAAA" * 50])   # 8-bit 6-mer counters wrap
    out.append(["ACGT" * 200, "ACGT" * 130 + "T", "CGTA" * 70])
    out.append(["NNNNNNNNNN", "ACGTNNNACGT", "NNNN", "ACGTACGTAC"])
    out.append(["ACGTACGT", "ACGTTACGT", "ACGACGT"])
    out.append(["acgtRYKM", "ACGTNNNN", "AC-GT.XU"])                    # case, IUPAC, gap characters, X, U
    return out


def reference_align(blks):
    inp = "\n\n".join("\n".join(b) for b in blks) + "\n"
    out = subprocess.run([MUSCLE_REF], input=inp.encode(), capture_output=True, check=True).stdout.decode()
    return [b.split("\n") for b in out.strip("\n").split("\n\n")]
