"""Edge cases run side by side through the REFERENCE binary (oracle/_ref/parsnp_core_ref, built from /root/reference in
the build container and shipped to the GPU box as a prebuilt file) and through our parsnp_core: ini flags, tiny and
identical genomes, duplicated / reverse-complemented / truncated queries, N runs, non-default LCB parameters, heavy
rearrangement.  CPU run: host logic + CPU checker behind the C ABI; GPU run: the product binary."""
import os
import subprocess

import numpy as np
import pytest

import oracles
import xmfa_util
from parsnp_amd import driver, synth
from parsnp_amd.paths import CORE_BIN

REFBIN = os.path.join(oracles.REFDIR, "parsnp_core_ref")
pytestmark = pytest.mark.skipif(not os.path.exists(REFBIN), reason="reference binary not built/shipped")


def cases():
    ref, gs = synth.population(seed=41, n=60000, n_genomes=5, div=0.02, indel_frac=0.05)
    r2, g2 = synth.population(seed=42, n=400, n_genomes=4, div=0.02)
    r3, g3 = synth.pop_rearranged(seed=43, n=120000, n_genomes=6, div=0.03, frac=0.3, block=5000)
    rc = oracles.revcomp
    return {
        "reverse_flag": (ref, [gs[0], rc(gs[1]), gs[2]], dict(edit=("reverse2=0", "reverse2=1"))),
        "revcomp_query": (ref, [gs[0], rc(gs[1]), gs[2]], {}),
        "tiny400": (r2, g2, {}),
        "identical": (ref, [ref, ref], {}),
        "one_identical": (ref, [ref, gs[0]], {}),
        "dup_query": (ref, [gs[0] + gs[0][:20000], gs[1]], {}),
        "queries_equal": (ref, [gs[0], gs[0], gs[1]], {}),
        "leadingN": (b"N" * 500 + ref, [gs[0], b"N" * 77 + gs[1] + b"N" * 300], {}),
        "params_cd": (ref, gs[:3], dict(mincluster=100, clusterd=50)),
        "diagdiff_abs": (ref, gs[:3], dict(diagdiff=25)),
        "diagdiff_small": (ref, gs[:3], dict(diagdiff=0.01)),
        "fixed_lengths": (ref, gs[:3], dict(anchors="25", mums="12")),
        "short_query": (ref, [gs[0], gs[1][:3000]], {}),
        "heavy_rearr": (r3, g3, {}),
        "single_query": (ref, gs[:1], {}),
        # long single-symbol runs (scaffold gaps, homopolymers): one K-mer chain with thousands of entries
        "long_runs": (ref[:20000] + b"N" * 9000 + ref[20000:40000] + b"A" * 3000 + ref[40000:],
                      [gs[0][:20000] + b"N" * 9000 + gs[0][20000:40000] + b"A" * 2500 + gs[0][40000:],
                       gs[1][:20000] + b"N" * 6000 + gs[1][20000:],
                       gs[2][:30000] + b"N" * 12000 + gs[2][30000:]], {}),
        "longer_runs": (ref[:20000] + b"N" * 40000 + ref[20000:],
                        [gs[0][:20000] + b"N" * 40000 + gs[0][20000:], gs[1][:20000] + b"N" * 25000 + gs[1][20000:40000] + b"T" * 30000 + gs[1][40000:],
                         gs[2], gs[3][:50000] + b"N" * 50000 + gs[3][50000:]], {}),
        "unaligned_out": (r3, g3[:4], dict(unaligned=1)),          # parsnp.unalign (setUnalignableRegions)
        "recomb_blocks": (ref, gs[:3], dict(recombfilt=1)),         # blocks/b<k>/seq.fna
    }


CASES = cases()


def run(core, rp, qs, out, kw):
    kw = dict(kw)
    edit = kw.pop("edit", None)
    os.makedirs(out, exist_ok=True)
    ini = os.path.join(out, "parsnpAligner.ini")
    txt = driver.ini_text(rp, qs, out, **kw)
    if edit:
        txt = txt.replace(*edit)
    open(ini, "w").write(txt)
    p = subprocess.run([core, ini], cwd=out, capture_output=True, text=True, timeout=900)
    x = os.path.join(out, "parsnpAligner.xmfa")
    lg = os.path.join(out, "parsnpAligner.log")
    extra = []   # every other file the run left in its output directory (parsnp.unalign, blocks/b*/seq.fna)
    for d, _, fs in sorted(os.walk(out)):
        for f in sorted(fs):
            if f in ("parsnp.unalign", "seq.fna"):
                extra.append((os.path.relpath(os.path.join(d, f), out), xmfa_util.md5(os.path.join(d, f))))
    return (p.returncode, xmfa_util.mum_lcb_signature(x) if os.path.exists(x) else None,
            xmfa_util.log_counters(lg) if os.path.exists(lg) else None,
            xmfa_util.md5(x) if os.path.exists(x) else None, extra)


def side_by_side(core, name, tmp_path):
    ref, gs, kw = CASES[name]
    rp, qs = synth.write_set(str(tmp_path / "in"), ref, gs)
    a = run(REFBIN, rp, qs, str(tmp_path / "ref"), kw)
    b = run(core, rp, qs, str(tmp_path / "mine"), kw)
    assert a == b
    assert a[0] == 0
    if name == "unaligned_out":
        assert [f for f, _ in a[4]] == ["parsnp.unalign"]
    if name == "recomb_blocks":
        assert a[4] and all(f.startswith("blocks/") for f, _ in a[4])


@pytest.mark.parametrize("name", sorted(CASES))
def test_edge_case_host_logic(cpu_checkers, tmp_path, name):
    side_by_side(cpu_checkers, name, tmp_path)


def test_long_runs_with_emulated_engine(emu, tmp_path):
    """the engine's own kernels (run sequentially on the host) on thousands-long N / homopolymer runs: the run-length
    shortcut of RepeatLength and SeedExtend's walk over a K-mer chain with thousands of entries"""
    side_by_side(emu[1], "long_runs", tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_edge_case_on_gpu(tmp_path, name):
    side_by_side(CORE_BIN, name, tmp_path)
