"""Pins oracle/mum_oracle.c (the CPU restatement) to the REFERENCE's own csgmum code (oracle/_ref/libcsgmum_ref.so,
compiled from /root/reference by oracle/Makefile).  Skipped where the reference build is absent; the committed
golden vectors (tests/golden, test_golden.py) cover that case."""
import numpy as np
import pytest

import oracles
from seqgen import adversarial_case, mutate, random_seq

pytestmark = pytest.mark.skipif(not oracles.have_reference(), reason="oracle/_ref/libcsgmum_ref.so not built")


@pytest.fixture(scope="module")
def libs():
    return oracles.load_restatement(), oracles.load_reference()


def test_find_um_raw_and_propagated_random(libs):
    O, R = libs
    rng = np.random.default_rng(1)
    for it in range(1500):
        ref, (q,) = adversarial_case(rng)
        if not any(c in ref for c in q):   # Find_UM's start-up loop needs one query symbol with a root arc (mum.c:193-198)
            continue
        u0, e0, s0 = oracles.reference_find_um(R, ref, q)
        u1, e1, s1 = oracles.restatement_find_um(O, ref, q)
        assert np.array_equal(e0, e1) and np.array_equal(u0, u1), (it, ref, q)
        m = e0 > 0
        assert np.array_equal(s0[m], s1[m]), (it, ref, q)
        u0, e0, s0 = oracles.reference_find_um(R, ref, q, propagate=True)
        u1, e1, s1 = oracles.restatement_find_um(O, ref, q, propagate=True)
        assert np.array_equal(e0, e1) and np.array_equal(u0, u1), (it, ref, q)
        m = u0 < e0
        assert np.array_equal(s0[m], s1[m]), (it, ref, q)


def _same_candidates(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a[:4], b[:4]))


def test_multi_mum_random(libs):
    O, R = libs
    rng = np.random.default_rng(2)
    total = 0
    for it in range(1200):
        nq = int(rng.integers(1, 5))
        ref, qs = adversarial_case(rng, 10, 80, nq)
        if any(not any(c in ref for c in q) for q in qs) or any(not any(c in ref for c in oracles.revcomp(q)) for q in qs):
            continue
        minsize = int(rng.integers(3, 11))
        a = oracles.reference_multi_mum(R, [ref] + qs, minsize, want_master=True)
        b = oracles.restatement_multi_mum(O, [ref] + qs, minsize, 1, want_master=True)
        assert _same_candidates(a, b), (it, ref, qs, minsize)
        assert np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5]), (it, "master")
        # the reduced event stream (events >= minsize only) yields the same candidate list
        c = oracles.restatement_multi_mum(O, [ref] + qs, minsize, minsize)
        assert _same_candidates(a, c), (it, ref, qs, minsize, "reduced")
        total += len(a[0])
    assert total > 500


def test_multi_mum_medium(libs):
    O, R = libs
    rng = np.random.default_rng(3)
    for n, nq, sub in ((3000, 3, 0.03), (20000, 2, 0.02)):
        ref = random_seq(rng, n)
        qs = []
        for g in range(nq):
            q = mutate(rng, ref, sub=sub, indel=0.002)
            if g == 1:   # an inverted segment -> reverse-strand MUMs
                a = n // 3; b = a + n // 5
                q = q[:a] + oracles.revcomp(q[a:b]) + q[b:]
            qs.append(q)
        minsize = 12
        a = oracles.reference_multi_mum(R, [ref] + qs, minsize)
        b = oracles.restatement_multi_mum(O, [ref] + qs, minsize, minsize)
        assert len(a[0]) > 20
        assert _same_candidates(a, b)
        assert (a[3] == 0).any()


def test_min_length_table(libs):
    import os, subprocess
    O, _ = libs
    calc = os.path.join(oracles.REFDIR, "calc_ref")
    if not os.path.exists(calc):
        pytest.skip("calc_ref not built")
    rng = np.random.default_rng(4)
    S = sorted(set(list(range(1, 3000)) + [int(x) for x in rng.integers(1, 20_000_000, 3000)] + [2 ** k for k in range(1, 25)]
                   + [2 ** k - 1 for k in range(2, 25)] + [2 ** k + 1 for k in range(1, 25)]))
    for expr in ("1.1*(Log(S))", "25", "2*(Log(S))", "1.5*(Log(S))+3", "(Log(S))", "0.5*(Log(S))-1", "S/1000+7"):
        out = subprocess.run([calc, expr] + [str(s) for s in S], capture_output=True, text=True, check=True).stdout.split("\n")
        want = [int(x.split()[1]) for x in out if x]
        got = [O.oracle_min_length(expr.encode(), s) for s in S]
        assert got == want, expr


def test_mumi_coverage_random(libs):
    import ctypes as C
    O, R = libs
    O.oracle_mumi_coverage.restype = C.c_int64
    R.ref_mumi_coverage.restype = C.c_long
    rng = np.random.default_rng(3)
    tot = 0
    for it in range(1000):
        ref, (q,) = adversarial_case(rng, 20, int(rng.choice([60, 200])))
        if it % 3 == 0:
            q = mutate(rng, ref, sub=0.03)
        if it % 50 == 0:
            q = ref
        if not any(c in ref for c in q) or not any(c in ref for c in oracles.revcomp(q)):
            continue
        a = R.ref_mumi_coverage(ref, C.c_long(len(ref)), q, oracles.revcomp(q), C.c_long(len(q)), 2)
        b = O.oracle_mumi_coverage(ref, C.c_int64(len(ref)), q, C.c_int64(len(q)), 1)
        c = O.oracle_mumi_coverage(ref, C.c_int64(len(ref)), q, C.c_int64(len(q)), 15)   # events < 15 cannot matter
        assert a == b == c, (it, ref, q)
        tot += a
    assert tot > 10000
