"""CPU restatement (oracle/mum_oracle.c) and host arithmetic against the COMMITTED golden vectors (tests/golden,
generated from the reference by tests/golden/make_golden.py).  Runs without the reference and without a GPU."""
import glob
import json
import os
import subprocess
import tarfile

import numpy as np
import pytest

import oracles

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def O(cpu_checkers):
    return oracles.load_restatement()


def test_min_length_table_oracle(O):
    t = json.load(open(os.path.join(G, "calc_table.json")))
    for expr, want in t["minsize"].items():
        got = [O.oracle_min_length(expr.encode(), s) for s in t["S"]]
        assert got == want, expr


def test_min_length_table_product(cpu_checkers):
    t = json.load(open(os.path.join(G, "calc_table.json")))
    for expr, want in t["minsize"].items():
        out = subprocess.run([cpu_checkers, "--min-length", expr] + [str(s) for s in t["S"]], capture_output=True, text=True, check=True).stdout
        got = [int(l.split()[1]) for l in out.splitlines() if l]
        assert got == want, expr


def test_find_um_golden(O):
    z = np.load(os.path.join(G, "find_um.npz"))
    for i in range(50):
        ref = z["c%d_ref" % i].tobytes(); q = z["c%d_q" % i].tobytes()
        u, e, s = oracles.restatement_find_um(O, ref, q)
        raw = z["c%d_raw" % i]
        assert np.array_equal(raw[0], u) and np.array_equal(raw[1], e)
        assert np.array_equal(raw[2][e > 0], s[e > 0])
        u, e, s = oracles.restatement_find_um(O, ref, q, propagate=True)
        prop = z["c%d_prop" % i]
        assert np.array_equal(prop[0], u) and np.array_equal(prop[1], e)
        assert np.array_equal(prop[2][u < e], s[u < e])


def mers(tmp_path_factory=None, base=None):
    base = base or str(tmp_path_factory.mktemp("mers"))
    with tarfile.open(os.path.join(G, "mers_virus.tar.xz")) as t:
        t.extractall(base)
    ref = os.path.join(base, "mers_virus", "ref", "England1.fna")
    qs = sorted(glob.glob(os.path.join(base, "mers_virus", "genomes", "*.fna")))
    return ref, qs


def read_fasta(path):
    return "".join(l.strip() for l in open(path) if not l.startswith(">")).upper().encode()


def test_mers_golden(O, tmp_path):
    ref, qs = mers(base=str(tmp_path))
    z = np.load(os.path.join(G, "find_um.npz"))
    r = read_fasta(ref)
    for n in range(4):
        qi, strand = z["m%d_q" % n]
        q = read_fasta(qs[qi])
        if strand:
            q = oracles.revcomp(q)
        u, e, s = oracles.restatement_find_um(O, r, q, propagate=True)
        prop = z["m%d_prop" % n]
        assert np.array_equal(prop[0], u) and np.array_equal(prop[1], e)
        assert np.array_equal(prop[2][u < e], s[u < e])
    g2 = np.load(os.path.join(G, "mers_anchor.npz"))
    seqs = [r] + [read_fasta(p) for p in qs]
    for min_event in (1, 17):
        k, lon, sp, fw, mu, me = oracles.restatement_multi_mum(O, seqs, 17, min_event, want_master=True)
        assert np.array_equal(k, g2["k"]) and np.array_equal(lon, g2["lon"]) and np.array_equal(sp, g2["sp"]) and np.array_equal(fw, g2["fwd"])
        if min_event == 1:   # the reduced event stream changes Master only where it cannot matter
            assert np.array_equal(me, g2["masterEP"]) and np.array_equal(mu, g2["masterUP"])
