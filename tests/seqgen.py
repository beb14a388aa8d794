"""Seeded synthetic sequence generators shared by tests, golden-vector scripts and bench.py."""
import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_seq(rng, n, alphabet=b"ACGT"):
    a = np.frombuffer(alphabet, dtype=np.uint8)
    return a[rng.integers(0, len(a), n)].tobytes()


def mutate(rng, s: bytes, sub=0.02, indel=0.0):
    a = np.frombuffer(s, dtype=np.uint8).copy()
    m = rng.random(len(a)) < sub
    a[m] = BASES[rng.integers(0, 4, int(m.sum()))]
    if indel > 0:
        keep = rng.random(len(a)) >= indel
        a = a[keep]
    return a.tobytes()


def adversarial_case(rng, n_lo=8, n_hi=60, nq=1):
    """Small (R, Q...) with planted repeats, N runs, 2-letter alphabets, rotations and reverse-complement segments."""
    from oracles import revcomp
    alpha = [b"ACGT", b"AC", b"ACGTN", b"AN"][rng.integers(0, 4)]
    n = int(rng.integers(n_lo, n_hi))
    ref = bytearray(random_seq(rng, n, alpha))
    if rng.random() < 0.5 and n > 12:  # planted repeat inside R
        L = int(rng.integers(3, n // 3)); a = int(rng.integers(0, n - L)); b = int(rng.integers(0, n - L))
        ref[b:b + L] = ref[a:a + L]
    ref = bytes(ref)
    qs = []
    for _ in range(nq):
        mode = rng.integers(0, 6)
        if mode == 0:
            q = mutate(rng, ref, sub=0.1)
        elif mode == 1:
            r = int(rng.integers(0, n)); q = ref[r:] + ref[:r]
        elif mode == 2:
            q = revcomp(mutate(rng, ref, sub=0.05))
        elif mode == 3:
            a = int(rng.integers(0, n)); b = int(rng.integers(a, n + 1))
            q = mutate(rng, ref[:a] + revcomp(ref[a:b]) + ref[b:], sub=0.03)
        elif mode == 4:
            q = random_seq(rng, int(rng.integers(1, n_hi)), alpha)
        else:
            q = mutate(rng, ref + ref[: n // 2], sub=0.05, indel=0.03)
        if len(q) == 0:
            q = b"A"
        qs.append(q)
    return ref, qs
