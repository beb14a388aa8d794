"""Regenerates the committed golden vectors from the REFERENCE (oracle/_ref, built from /root/reference by
oracle/Makefile).  Run in the build container only:  python tests/golden/make_golden.py

  calc_table.json   G5  minimum-MUM-length table from the reference's Converter/Calculator
  find_um.npz       G1  raw + propagated (UP,EP,SP) of 50 random small (R,Q) pairs and 4 MERS genome/strand pairs
  mers_anchor.npz   G2  candidate list + Master arrays of the MERS anchor pass (47 genomes)
  mumi.json         G4  all.mumi (calcmumi=1) of the reference binary on MERS, the messy multi-contig set and a p-limited set
  e2e.json          G3  XMFA md5, MUM/LCB signature md5 and log counters of the reference binary on
                        MERS, viral50, pop6x200k, rearr6x300k, pop12x400k, pop20x1m (21 x 1 Mb), bact8 (9 x 5 Mb) (inputs: tests/golden/mers_virus.tar.xz / parsnp_amd.synth seeds)
"""
import glob
import json
import os
import shutil
import subprocess
import sys
import tarfile
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracles  # noqa: E402
import xmfa_util  # noqa: E402
from parsnp_amd import driver, synth  # noqa: E402
from seqgen import adversarial_case  # noqa: E402

REFBIN = os.path.join(oracles.REFDIR, "parsnp_core_ref")
CALC = os.path.join(oracles.REFDIR, "calc_ref")
EXPRS = ["1.1*(Log(S))", "25", "2*(Log(S))", "1.5*(Log(S))+3", "(Log(S))", "0.5*(Log(S))-1", "S/1000+7"]


def read_fasta(path):
    return "".join(l.strip() for l in open(path) if not l.startswith(">")).upper().encode()


def mers_paths(tmp):
    with tarfile.open(os.path.join(HERE, "mers_virus.tar.xz")) as t:
        t.extractall(tmp)
    ref = os.path.join(tmp, "mers_virus", "ref", "England1.fna")
    qs = sorted(glob.glob(os.path.join(tmp, "mers_virus", "genomes", "*.fna")))
    return ref, qs


def e2e_goldens(tmp, mref, mqs, only=None, e2e=None):
    """XMFA md5 / MUM-LCB signature / log counters of the reference binary; only = names to (re)generate into e2e"""
    e2e = {} if e2e is None else e2e

    def run(name, make, **kw):
        if only and name not in only:
            return
        rp, qs = make()
        out = os.path.join(tmp, "out_" + name)
        rc, _ = driver.run_core(REFBIN, rp, qs, out, **kw)
        assert rc == 0, name
        x = os.path.join(out, "parsnpAligner.xmfa")
        e2e[name] = dict(xmfa_md5=xmfa_util.md5(x), signature=xmfa_util.mum_lcb_signature(x),
                         log=xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")),
                         ref_records=[h for h, _ in xmfa_util.records(x)[1] if h.startswith("> 1:")][:50])

    def synthetic(name, sub=None):
        def make():
            r, gs = synth.make(name)
            return synth.write_set(os.path.join(tmp, sub or name), r, gs)
        return make
    # file names matter (##SequenceFile): MERS under its own names, synthetic sets as ref.fna / g%04d.fna
    run("mers", lambda: (mref, mqs))
    for name in ("viral50", "pop6x200k", "rearr6x300k", "pop12x400k", "pop20x1m", "bact8", "poprearr10x400k", "popinv12x400k"):   # bact8 takes ~80 s
        run(name, synthetic(name))
    run("messy", lambda: synth.messy_set(os.path.join(tmp, "messy")))
    run("pchunk", synthetic("pop6x200k", "pchunk"), partpos=66660)      # 3 reference chunks + the <50 bp tail rule (src/parsnp.cpp:1527-1538)
    # draft assemblies: shuffled / reverse-complemented contigs joined by N runs
    run("draft8x300k", lambda: synth.draft_set(os.path.join(tmp, "draft8"), n=300_000, n_genomes=8, contigs=60))
    run("draft20x1m", lambda: synth.draft_set(os.path.join(tmp, "draft20"), n=1_000_000, n_genomes=20, contigs=300))
    return e2e


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--e2e-only":     # regenerate / add single end-to-end goldens
        tmp = tempfile.mkdtemp()
        mref, mqs = mers_paths(tmp)
        path = os.path.join(HERE, "e2e.json")
        e2e = e2e_goldens(tmp, mref, mqs, only=set(sys.argv[2:]), e2e=json.load(open(path)))
        json.dump(e2e, open(path, "w"), indent=1)
        shutil.rmtree(tmp)
        print("e2e goldens:", sorted(e2e))
        return
    R = oracles.load_reference()
    # ---- G5
    rng = np.random.default_rng(4)
    S = sorted(set(list(range(1, 400)) + [int(x) for x in rng.integers(1, 20_000_000, 400)] + [2 ** k for k in range(1, 25)]
                   + [2 ** k - 1 for k in range(2, 25)] + [2 ** k + 1 for k in range(1, 25)] + [30, 31, 1000, 30000, 1000000, 5000000]))
    table = {}
    for e in EXPRS:
        out = subprocess.run([CALC, e] + [str(s) for s in S], capture_output=True, text=True, check=True).stdout.split("\n")
        table[e] = [int(x.split()[1]) for x in out if x]
    json.dump({"S": S, "minsize": table}, open(os.path.join(HERE, "calc_table.json"), "w"))

    # ---- G1
    rng = np.random.default_rng(101)
    g1 = {}
    i = 0
    while i < 50:
        ref, (q,) = adversarial_case(rng, 10, 120)
        if not any(c in ref for c in q):
            continue
        u, e, s = oracles.reference_find_um(R, ref, q)
        pu, pe, ps = oracles.reference_find_um(R, ref, q, propagate=True)
        g1["c%d_ref" % i] = np.frombuffer(ref, np.uint8); g1["c%d_q" % i] = np.frombuffer(q, np.uint8)
        g1["c%d_raw" % i] = np.stack([u, e, s.astype(np.int64)]); g1["c%d_prop" % i] = np.stack([pu, pe, ps.astype(np.int64)])
        i += 1
    tmp = tempfile.mkdtemp()
    mref, mqs = mers_paths(tmp)
    ref = read_fasta(mref)
    for n, (qi, strand) in enumerate([(0, 0), (7, 1), (18, 0), (45, 1)]):
        q = read_fasta(mqs[qi])
        if strand:
            q = oracles.revcomp(q)
        pu, pe, ps = oracles.reference_find_um(R, ref, q, propagate=True)
        g1["m%d_q" % n] = np.array([qi, strand]); g1["m%d_prop" % n] = np.stack([pu, pe, ps.astype(np.int64)])
    np.savez_compressed(os.path.join(HERE, "find_um.npz"), **g1)

    # ---- G2
    seqs = [ref] + [read_fasta(p) for p in mqs]
    k, lon, sp, fw, mu, me = oracles.reference_multi_mum(R, seqs, 17, want_master=True)
    np.savez_compressed(os.path.join(HERE, "mers_anchor.npz"), k=k, lon=lon, sp=sp, fwd=fw, masterUP=mu, masterEP=me)

    # ---- G3
    e2e = e2e_goldens(tmp, mref, mqs)
    json.dump(e2e, open(os.path.join(HERE, "e2e.json"), "w"), indent=1)
    # ---- G4: calcmumi=1 -> all.mumi
    mumi = {}

    def run_mumi(name, rp, qs, **kw):
        out = os.path.join(tmp, "mumi_" + name)
        rc, _ = driver.run_core(REFBIN, rp, qs, out, calcmumi=1, **kw)
        assert rc == 0, name
        mumi[name] = sorted(open(os.path.join(out, "all.mumi")).read().split(), key=lambda x: int(x.split(":")[0]))
    run_mumi("mers", mref, mqs)
    rp, qs = synth.messy_set(os.path.join(tmp, "messy2"))
    run_mumi("messy", rp, qs)
    r, gs = synth.make("pop6x200k")
    rp, qs = synth.write_set(os.path.join(tmp, "pop_p"), r, gs)
    run_mumi("pop6x200k_p", rp, qs, partpos=40000)
    json.dump(mumi, open(os.path.join(HERE, "mumi.json"), "w"), indent=0)
    shutil.rmtree(tmp)
    print("goldens written:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
