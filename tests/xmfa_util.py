"""XMFA comparison helpers.  MUM columns are lower case in every row, inter-MUM gap columns upper case or '-'
(src/parsnp.cpp:684-693, :746-753), so the lower-case projection of a record pins the MUM coordinates and the
'>' lines pin the LCB boundaries even where the gap aligner (MUSCLE in the reference) differs."""
import hashlib


def records(path):
    out, hdr, seq = [], None, []
    head = []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith("#"):
                head.append(line)
            elif line.startswith(">"):
                if hdr is not None:
                    out.append((hdr, "".join(seq)))
                hdr, seq = line, []
            elif line == "=":
                if hdr is not None:
                    out.append((hdr, "".join(seq)))
                hdr, seq = None, []
                out.append(("=", ""))
            else:
                seq.append(line)
    return head, out


def mum_lcb_signature(path):
    """header lines + per record ('> ...' line, lower-case columns only)"""
    head, recs = records(path)
    h = hashlib.md5()
    for line in head:
        h.update(line.encode() + b"\n")
    for hdr, seq in recs:
        h.update(hdr.encode() + b"\n")
        h.update("".join(c for c in seq if c.islower()).encode() + b"\n")
    return h.hexdigest()


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def log_counters(path):
    keep = ("Mum anchor size", "Number of MUM anchors found", "Number of MUMs found", "Total MUMs found", "Number of MUMs filtered",
            "Number of Clusters filtered", "Number of clusters created", "Average cluster length", "Total coverage among all sequences",
            "Cluster coverage in sequence")
    return [l.rstrip() for l in open(path) if l.strip().startswith(keep)]


_COMP = bytes.maketrans(b"ACGTN", b"TGCAN")


def consistency(path, genomes):
    """Size-independent self-check of an XMFA against the genomes it was made from (single-contig genomes, in file order):
    per LCB all rows have one length, MUM (lower-case) columns hold no gap and agree in every row, and every record
    spells genome[start-1:end] (reverse-complemented for '-' records) once its gap characters are removed.
    -> dict of counts; records whose LCB was overlap-trimmed by the writer (src/parsnp.cpp:928-952 shifts the start by a
    column count, a reference quirk) are counted separately in `shifted` instead of `bad_sequence`."""
    import re
    head, recs = records(path)
    stats = dict(lcbs=0, records=0, bad_length=0, bad_mum_column=0, bad_sequence=0, shifted=0, reverse=0)
    block = []
    for hdr, seq in recs:
        if hdr != "=":
            block.append((hdr, seq))
            continue
        if not block:
            continue
        stats["lcbs"] += 1
        lengths = {len(s) for _, s in block}
        if len(lengths) != 1:
            stats["bad_length"] += 1
        first = block[0][1]
        lower = [i for i, c in enumerate(first) if c.islower()]
        for _, s in block:
            if len(s) == len(first) and any(not s[i].islower() or s[i] != first[i] for i in lower):
                stats["bad_mum_column"] += 1
        for h, s in block:
            m = re.match(r"> (\d+):(\d+)-(\d+) ([+-]) ", h)
            g, a, b, strand = int(m.group(1)) - 1, int(m.group(2)), int(m.group(3)), m.group(4)
            stats["records"] += 1
            want = genomes[g][a - 1:b].upper()
            if strand == "-":
                stats["reverse"] += 1
                want = want.translate(_COMP)[::-1]
            got = s.replace("-", "").upper().encode()
            if got != want:
                # trimmed at the front by the writer: the sequence is a suffix of what the coordinates say, or vice versa
                if want.endswith(got) or got.endswith(want):
                    stats["shifted"] += 1
                else:
                    stats["bad_sequence"] += 1
        block = []
    return stats


def usable_threads(cap=16):
    """threads a test may start: the affinity mask capped by the container's CPU quota (cgroup cpu.max) and by `cap`"""
    import os
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(cap, n))


def native_consistency(path, genome_dir, merged=False, threads=None, intervals=None):
    """the same counts from oracle/_ref/xmfa_check (tests/emu/xmfa_check.cpp) -- for XMFA files of GB size.  genome_dir holds
    the FASTA files the header's ##SequenceFile entries name.  merged: a partition merge (see xmfa_check.cpp)."""
    import json
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "xmfa_check")
    cmd = [exe] + (["--merged"] if merged else []) + (["--intervals", intervals] if intervals else []) + [path, genome_dir, str(threads or usable_threads())]
    return json.loads(subprocess.run(cmd, check=True, capture_output=True, text=True, env=dict(os.environ, OMP_WAIT_POLICY="passive")).stdout)
