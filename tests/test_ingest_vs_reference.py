"""FASTA files as they come in the wild -- CRLF, no final newline, lower case, blank lines, blanks inside lines, very long
lines, IUPAC codes / digits / '*' / '.', several contigs, a header of exactly and of more than getline's 2499 bytes, a bare
'>' -- through the REFERENCE binary and through our parsnp_core (host/ingest.cpp, the restatement of the reading loop of
src/parsnp.cpp:2913-3160; its sequence bytes go through a loop without branches and a byte histogram).  Same exit code,
same XMFA bytes, same log counters.  CPU only: ingest is host code, the engine behind it is the CPU checker."""
import os
import subprocess

import numpy as np
import pytest

import oracles
import xmfa_util
from parsnp_amd import driver, synth

REFBIN = os.path.join(oracles.REFDIR, "parsnp_core_ref")
pytestmark = pytest.mark.skipif(not os.path.exists(REFBIN), reason="reference binary not built/shipped")


def fasta(seq, width=80, nl="\n", hdr=">s", tail=True, lower=False, blank_every=0, spaces=False, contigs=None):
    s = seq.decode()
    if lower:
        s = s.lower()
    out = []
    parts = [s] if not contigs else [s[a:b] for a, b in zip([0] + contigs, contigs + [len(s)])]
    for k, part in enumerate(parts):
        out.append((hdr if k == 0 else ">c%d some text" % k) + nl)
        for i in range(0, len(part), width):
            line = part[i:i + width]
            if spaces and (i // width) % 7 == 3:
                line = line[:10] + " \t" + line[10:]
            out.append(line + nl)
            if blank_every and (i // width) % blank_every == blank_every - 1:
                out.append(nl)
    t = "".join(out)
    return (t if tail else t.rstrip("\r\n")).encode()


def sprinkle(seq, rng):
    b = bytearray(seq)
    for p in rng.integers(0, len(b), 60):
        b[p] = rng.choice(list(b"RYKMSWBDHVNXU-*1."))
    return bytes(b)


FORMATS = {
    "crlf": dict(nl="\r\n"), "no_trailing_newline": dict(tail=False), "lowercase": dict(lower=True), "blank_lines": dict(blank_every=5),
    "spaces_tabs": dict(spaces=True), "width_3000": dict(width=3000), "long_header": dict(hdr=">" + "x" * 2600),
    "header_2499": dict(hdr=">" + "x" * 2498), "contigs": dict(contigs=[9000, 9000, 25000]), "bare_gt": dict(hdr=">"),
}


def run(core, rp, qs, out):
    os.makedirs(out, exist_ok=True)
    ini = os.path.join(out, "parsnpAligner.ini")
    open(ini, "w").write(driver.ini_text(rp, qs, out))
    p = subprocess.run([core, ini], cwd=out, capture_output=True, text=True, timeout=600)
    x = os.path.join(out, "parsnpAligner.xmfa")
    lg = os.path.join(out, "parsnpAligner.log")
    return (p.returncode, xmfa_util.md5(x) if os.path.exists(x) else None,
            xmfa_util.log_counters(lg) if os.path.exists(x) else (open(lg).read() if os.path.exists(lg) else None))


@pytest.mark.parametrize("who", ["reference_file", "query_files", "all_files_with_codes"])
@pytest.mark.parametrize("fmt", sorted(FORMATS))
def test_fasta_format_vs_reference(cpu_checkers, tmp_path, fmt, who):
    ref, gs = synth.population(seed=51, n=40000, n_genomes=3, div=0.02, indel_frac=0.05)
    seqs = [ref] + gs
    if who == "all_files_with_codes":
        rng = np.random.default_rng(3)
        seqs = [sprinkle(s, rng) for s in seqs]
    os.makedirs(tmp_path / "in")
    paths = []
    for i, s in enumerate(seqs):
        odd = (who != "query_files" and i == 0) or (who != "reference_file" and i > 0)
        p = str(tmp_path / "in" / ("g%d.fna" % i))
        open(p, "wb").write(fasta(s, **(FORMATS[fmt] if odd else {})))
        paths.append(p)
    a = run(REFBIN, paths[0], paths[1:], str(tmp_path / "ref"))
    b = run(cpu_checkers, paths[0], paths[1:], str(tmp_path / "mine"))
    assert a == b
    assert a[0] == 0
