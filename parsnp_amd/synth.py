"""Seeded synthetic genome sets for the BASELINE.json configurations.

All sets use numpy default_rng(seed), bases uniform over ACGT, one contig per genome, FASTA wrapped at 80 columns,
files ref.fna + g0000.fna ... with headers >ref / >g0000 ...

  population model  a pool of segregating sites (Bernoulli density `div`) with one alternate allele each; every genome
                    carries each alternate allele with probability 1/2; a fraction of the sites are 1-bp deletions.
                    (Independent per-genome mutation leaves no core genome at 200 genomes.)
  musclefree model  segregating sites on a jittered grid, >= 40 bp apart: every inter-MUM gap is one column, so the
                    XMFA does not depend on the gap aligner and can be compared byte for byte.
  rearranged model  per-genome independent substitutions plus inversions / translocations of 20 kb blocks.
"""
import os

import numpy as np

_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def _alt(rng, ref_bases):
    """an alternate allele different from the reference base"""
    idx = np.searchsorted(_BASES, ref_bases)  # ACGT is sorted
    return _BASES[(idx + rng.integers(1, 4, len(ref_bases))) % 4]


def random_genome(rng, n):
    return _BASES[rng.integers(0, 4, n)]


def population(seed, n, n_genomes, div, indel_frac=0.0, sites=None, carry_seed=None, select=None):
    """-> (ref bytes, [genome bytes]) under the population model.  carry_seed: draw the genomes from a separate
    stream (same reference and site pool, different genomes -- one partition per rank in bench.py).  select: indices
    of the genomes to materialise (the random stream still advances over all of them, so genome i is the same bytes
    whether or not its neighbours are built) -- one partition of BASELINE config 4 without building 2000 genomes."""
    rng = np.random.default_rng(seed)
    ref = random_genome(rng, n)
    if sites is None:
        sites = np.flatnonzero(rng.random(n) < div)
    alt = _alt(rng, ref[sites])
    is_del = rng.random(len(sites)) < indel_frac
    if carry_seed is not None:
        rng = np.random.default_rng(carry_seed)
    out = []
    for gi in range(n_genomes):
        carry = rng.random(len(sites)) < 0.5
        if select is not None and gi not in select:
            continue
        g = ref.copy()
        sub = sites[carry & ~is_del]
        g[sub] = alt[carry & ~is_del]
        keep = np.ones(n, dtype=bool)
        keep[sites[carry & is_del]] = False
        out.append(g[keep].tobytes())
    return ref.tobytes(), out


def musclefree(seed, n, n_genomes, div):
    """sites on a grid of step 1/div with jitter <= step-40."""
    rng = np.random.default_rng(seed)
    step = int(round(1.0 / div))
    base = np.arange(step // 2, n - step, step)
    jitter = rng.integers(0, max(1, step - 40), len(base))
    sites = base + jitter
    return population(seed + 1, n, n_genomes, div, 0.0, sites=sites)


def rearranged(seed, n, n_genomes, div, frac=0.15, block=20000):
    rng = np.random.default_rng(seed)
    ref = random_genome(rng, n)
    out = []
    for _ in range(n_genomes):
        g = ref.copy()
        m = rng.random(n) < div
        g[m] = _alt(rng, g[m])
        s = g.tobytes()
        nblocks = max(1, int(frac * n / block))
        for _b in range(nblocks):
            a = int(rng.integers(0, max(1, len(s) - block)))
            seg = s[a:a + block]
            rest = s[:a] + s[a + block:]
            if rng.random() < 0.5:   # inversion in place
                s = s[:a] + seg.translate(_COMP)[::-1] + s[a + block:]
            else:                    # translocation
                b = int(rng.integers(0, len(rest)))
                s = rest[:b] + seg + rest[b:]
        out.append(s)
    return ref.tobytes(), out


def pop_rearranged(seed, n, n_genomes, div, indel_frac=0.05, frac=0.10, block=20000, carry_seed=None):
    """population model + per-genome inversions / translocations of `block`-sized segments covering `frac` of the genome
    (BASELINE config 5: stresses the recursive extension with asymmetric inter-MUM regions)."""
    ref, gs = population(seed, n, n_genomes, div, indel_frac, carry_seed=carry_seed)
    rng = np.random.default_rng(seed + 7919 if carry_seed is None else carry_seed + 7919)
    out = []
    for s in gs:
        for _b in range(max(1, int(frac * n / block))):
            a = int(rng.integers(0, max(1, len(s) - block)))
            seg = s[a:a + block]
            if rng.random() < 0.5:
                s = s[:a] + seg.translate(_COMP)[::-1] + s[a + block:]
            else:
                rest = s[:a] + s[a + block:]
                b = int(rng.integers(0, len(rest)))
                s = rest[:b] + seg + rest[b:]
        out.append(s)
    return ref, out


def pop_inverted(seed, n, n_genomes, div, indel_frac=0.05, inv_every=20, inv_len=200_000, carry_seed=None):
    """population model + ONE clean inversion of inv_len bases in every inv_every-th genome, each at its own place: what a bacterial
    population looks like to the anchor validation -- collinear but for a few inverted segments, whose anchor candidates the cheap
    running-extent test flags (they lie out of order in that genome) although hardly any of them overlaps anything."""
    ref, gs = population(seed, n, n_genomes, div, indel_frac, carry_seed=carry_seed)
    rng = np.random.default_rng(seed + 104729 if carry_seed is None else carry_seed + 104729)
    out = []
    for i, s in enumerate(gs):
        if i % inv_every == inv_every - 1 and len(s) > 2 * inv_len:
            a = int(rng.integers(0, len(s) - inv_len))
            s = s[:a] + s[a:a + inv_len].translate(_COMP)[::-1] + s[a + inv_len:]
        out.append(s)
    return ref, out


def write_multicontig(path, name, seq: bytes, cuts, width=80, crlf=False, lower=False):
    """FASTA with one record per contig (cuts = interior cut positions)"""
    nl = b"\r\n" if crlf else b"\n"
    with open(path, "wb") as f:
        edges = [0] + sorted(cuts) + [len(seq)]
        for c in range(len(edges) - 1):
            f.write(b">" + ("%s_contig%d some description" % (name, c + 1)).encode() + nl)
            part = seq[edges[c]:edges[c + 1]]
            if lower:
                part = part.lower()
            for i in range(0, len(part), width):
                f.write(part[i:i + width] + nl)


def write_fasta(path, name, seq: bytes, width=80):
    with open(path, "wb") as f:
        f.write(b">" + name.encode() + b"\n")
        a = np.frombuffer(seq, dtype=np.uint8)
        full = (len(a) // width) * width
        if full:
            rows = a[:full].reshape(-1, width)
            nl = np.full((rows.shape[0], 1), 10, dtype=np.uint8)
            f.write(np.hstack([rows, nl]).tobytes())
        if len(a) > full:
            f.write(a[full:].tobytes() + b"\n")


def write_set(outdir, ref: bytes, genomes, ids=None):
    """-> (ref path, [query paths]) with the file / header names the goldens were generated with (ids: the genome
    numbers to name the files after, default 0..)."""
    os.makedirs(outdir, exist_ok=True)
    rp = os.path.join(outdir, "ref.fna")
    write_fasta(rp, "ref", ref)
    qs = []
    for i, g in zip(ids if ids is not None else range(len(genomes)), genomes):
        p = os.path.join(outdir, "g%04d.fna" % i)
        write_fasta(p, "g%04d" % i, g)
        qs.append(p)
    return rp, qs


def write_contigs(path, name, contigs, width=80):
    with open(path, "wb") as f:
        for c, part in enumerate(contigs):
            f.write(b">" + ("%s_contig%d" % (name, c + 1)).encode() + b"\n")
            for i in range(0, len(part), width):
                f.write(part[i:i + width] + b"\n")


def draft_set(outdir, seed=31, n=300_000, n_genomes=8, contigs=60, div=0.02, indel_frac=0.05):
    """draft assemblies: every genome (the reference too) is cut at its own random positions into `contigs` contigs, the
    contigs are shuffled and about half of them reverse-complemented.  Ingest joins contigs with d+10 N's
    (src/parsnp.cpp:3040-3075), so every genome carries `contigs`-1 identical N runs: the repeat structure that the seed
    index handles worst, and many short reverse-strand LCBs.  -> (ref path, [query paths])"""
    rng = np.random.default_rng(seed)
    ref, gs = population(seed, n, n_genomes, div, indel_frac)
    os.makedirs(outdir, exist_ok=True)

    def cut(g):
        edges = [0] + sorted(int(x) for x in rng.choice(np.arange(200, len(g) - 200), contigs - 1, replace=False)) + [len(g)]
        parts = [g[edges[i]:edges[i + 1]] for i in range(contigs)]
        order = rng.permutation(contigs)
        return [parts[i].translate(_COMP)[::-1] if rng.random() < 0.5 else parts[i] for i in order]

    rp = os.path.join(outdir, "ref.fna")
    write_contigs(rp, "ref", cut(ref))
    qs = []
    for i, g in enumerate(gs):
        p = os.path.join(outdir, "g%04d.fna" % i)
        write_contigs(p, "g%04d" % i, cut(g))
        qs.append(p)
    return rp, qs


CONFIGS = {
    # name: (model, kwargs)  -- sizes of BASELINE.json configs 2, 3, 5 and the reduced sets used by tests
    "viral50": ("musclefree", dict(seed=3, n=30000, n_genomes=50, div=0.01)),
    "bact200": ("population", dict(seed=5, n=5_000_000, n_genomes=200, div=0.02, indel_frac=0.05)),
    "bact8": ("population", dict(seed=5, n=5_000_000, n_genomes=8, div=0.02, indel_frac=0.05)),
    "pop20x1m": ("population", dict(seed=7, n=1_000_000, n_genomes=20, div=0.02, indel_frac=0.05)),
    "pop6x200k": ("population", dict(seed=9, n=200_000, n_genomes=6, div=0.02, indel_frac=0.05)),
    "rearr6x300k": ("rearranged", dict(seed=11, n=300_000, n_genomes=6, div=0.004, frac=0.15)),
    # divergent, indel-rich population: one anchor candidate in forty overlaps an earlier one (the flagged / tangled routes of the validation)
    "pop12x400k": ("population", dict(seed=21, n=400_000, n_genomes=12, div=0.03, indel_frac=0.10)),
    "rearr500": ("pop_rearranged", dict(seed=13, n=5_000_000, n_genomes=500, div=0.05, frac=0.10)),
    "rearr50": ("pop_rearranged", dict(seed=13, n=5_000_000, n_genomes=50, div=0.05, frac=0.10)),   # = the first 50 genomes of rearr500
    # config 3 with an inverted 200-kb segment in every 20th genome (10 of 200): a third of the anchor candidates flagged, a few tangled
    "bact200inv": ("pop_inverted", dict(seed=5, n=5_000_000, n_genomes=200, div=0.02, indel_frac=0.05, inv_every=20, inv_len=200_000)),
    "popinv12x400k": ("pop_inverted", dict(seed=31, n=400_000, n_genomes=12, div=0.02, indel_frac=0.05, inv_every=5, inv_len=30_000)),   # 2 of 12 genomes inverted: the reduced form, with a golden
    "bact2000": ("population", dict(seed=6, n=5_000_000, n_genomes=2000, div=0.02, indel_frac=0.05)),
    "poprearr10x400k": ("pop_rearranged", dict(seed=13, n=400_000, n_genomes=10, div=0.05, frac=0.10)),
}


def config4_partition(part=0, n_total=2000, size=250):
    """genome numbers of partition `part` of BASELINE config 4 in the reference driver's order: file names sorted, shuffled
    with random.Random(42), cut into chunks of `size` (parsnp:29, :1509-1510, :1555-1564).  -> list of genome numbers"""
    import random
    names = sorted("g%04d.fna" % i for i in range(n_total))
    random.Random(42).shuffle(names)
    return [int(x[1:5]) for x in names[part * size:(part + 1) * size]]


def make_partition(part=0, **override):
    """(ref, genomes, ids) of one 250-genome partition of bact2000, genomes in the driver's order"""
    ids = config4_partition(part)
    model, kw = CONFIGS["bact2000"]
    kw = dict(kw, **override)
    ref, gs = population(select=set(ids), **kw)
    by_id = dict(zip(sorted(ids), gs))
    return ref, [by_id[i] for i in ids], ids


def make(name, **override):
    model, kw = CONFIGS[name]
    kw = dict(kw, **override)
    return {"population": population, "musclefree": musclefree, "rearranged": rearranged, "pop_rearranged": pop_rearranged, "pop_inverted": pop_inverted}[model](**kw)


def messy_set(outdir, seed=23, n=150_000, n_genomes=6):
    """multi-contig reference and queries, IUPAC codes, N runs, lower case, CRLF: exercises ingest (src/parsnp.cpp:2999-3160),
    the d+10 N padding between query contigs and the contig labels of the XMFA headers.  -> (ref path, [query paths])"""
    rng = np.random.default_rng(seed)
    ref, gs = population(seed, n, n_genomes, 0.02, 0.05)

    def dirty(s):
        a = np.frombuffer(s, dtype=np.uint8).copy()
        idx = rng.integers(0, len(a), 40)
        a[idx] = np.frombuffer(b"RYKMSWBDHVN-", dtype=np.uint8)[rng.integers(0, 12, 40)]
        st = int(rng.integers(1000, len(a) - 1000))
        a[st:st + int(rng.integers(5, 120))] = ord("N")
        return a.tobytes()

    os.makedirs(outdir, exist_ok=True)
    rp = os.path.join(outdir, "ref.fna")
    write_multicontig(rp, "ref", dirty(ref), sorted(int(x) for x in rng.integers(5000, n - 5000, 2)))
    qs = []
    for i, g in enumerate(gs):
        p = os.path.join(outdir, "g%04d.fna" % i)
        cuts = sorted(int(x) for x in rng.integers(5000, len(g) - 5000, int(rng.integers(1, 4))))
        write_multicontig(p, "g%04d" % i, dirty(g), cuts, crlf=(i == 1), lower=(i == 2))
        qs.append(p)
    return rp, qs
