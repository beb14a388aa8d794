"""TEST INFRASTRUCTURE ONLY -- Python restatement of the reference driver's partition.py, function by
function, without its Biopython / pyspoa dependencies.  The product's merge is native
(parsnp_amd/csrc/host/partition_merge.cpp behind include/parsnp_merge.h); tests compare it with this file byte for byte
and pin both against the hand-checked vectors of tests/golden/partition/.  Nothing under parsnp_amd/ imports this.

    interval_intersection, cut_overlaps         partition.py:35-61, :86-96
    chunk_intervals                             get_interval + get_chunked_intervals, :64-83, :507-536
    intersected_intervals                       get_intersected_intervals, :539-583 (intervals shorter than 10 dropped)
    trim_lcb / trim_xmfa                        trim, trim_single_xmfa, :99-216, :586-648 (prefix / suffix base counts, bisect)
    combine_header_info, write_combined_header  :245-318
    merge_blocks / merge_xmfas                  :320-433, :683-736

PARITY UNPINNED against the reference's own output: partition.py cannot be imported in the build container (Bio, spoa
absent, no network), so this restatement is pinned by code reading, by hand-checked vectors
(tests/golden/partition/README.md says how each expected file was derived) and by the properties
tests/test_partition_merge.py checks.  One deliberate difference: columns that are insertions relative to the reference
are re-aligned by the reference with spoa.poa (:386); SPOA is a third-party library that is not here, so
`align_insertions` is pluggable -- default: this project's gap aligner (the libMUSCLE restatement behind parsnp_core's
XMFA writer), else left-justified padding.  Reference-anchored columns, coordinates, headers and block order do not
depend on it.

XMFA records are read as Bio.AlignIO's "mauve" parser presents them to partition.py: start = printed start - 1, end =
printed end, strand +1/-1, name = the sequence index, id = the text after the strand ("clusterN sC:pP")."""
import bisect
import ctypes
import math
import os
import re
from collections import defaultdict

CHUNK_PREFIX = "chunk"
_HDR = re.compile(r"^> (\d+):(\d+)-(\d+) ([+-]) (.*)$")
_ID = re.compile(r"^cluster(\d+) s(\d+):p(\d+)")


class Rec:
    __slots__ = ("name", "start", "end", "strand", "id", "seq")

    def __init__(self, name, start, end, strand, ident, seq):
        self.name, self.start, self.end, self.strand, self.id, self.seq = name, start, end, strand, ident, seq

    def copy(self, seq=None):
        return Rec(self.name, self.start, self.end, self.strand, self.id, self.seq if seq is None else seq)


def read_header(path):
    """-> [(index, file, header, length)] of the ##Sequence... lines, in file order (combine_header_info's parse, :262-276)"""
    out, cur = [], {}
    with open(path) as f:
        for line in f:
            if not line.startswith("#"):
                break
            line = line.rstrip("\n")
            if line.startswith("##SequenceIndex"):
                cur = {"index": int(line.split(" ")[1])}
            elif line.startswith("##SequenceFile"):
                cur["file"] = line.split(" ")[1]
            elif line.startswith("##SequenceHeader"):
                cur["header"] = line.split(" ")[1]
            elif line.startswith("##SequenceLength"):
                cur["length"] = int(line.split(" ")[1][:-2])
                out.append((cur["index"], cur["file"], cur["header"], cur["length"]))
    return out


def read_lcbs(path):
    """generator of LCBs (lists of Rec) of an XMFA file"""
    block, hdr, seq = [], None, []

    def flush():
        if hdr is not None:
            m = _HDR.match(hdr)
            start, end = int(m.group(2)), int(m.group(3))
            if end != 0:            # Mauve's "0-0" marks a sequence that is absent from the block
                start -= 1
            block.append(Rec(int(m.group(1)), start, end, 1 if m.group(4) == "+" else -1, m.group(5), "".join(seq)))
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith("#"):
                continue
            if line.startswith(">"):
                flush()
                hdr, seq = line, []
            elif line == "=":
                flush()
                hdr, seq = None, []
                if block:
                    yield block
                block = []
            else:
                seq.append(line)
    flush()
    if block:
        yield block


# ---------------------------------------------------------------------------------------------- intervals
def interval_intersection(A, B):
    """partition.py:35-61 (half-open style test lo < hi, as written there)"""
    ans, i, j = [], 0, 0
    while i < len(A) and j < len(B):
        lo, hi = max(A[i][0], B[j][0]), min(A[i][1], B[j][1])
        if lo < hi:
            ans.append([lo, hi])
        if A[i][1] < B[j][1]:
            i += 1
        else:
            j += 1
    return ans


def cut_overlaps(ilist):
    """partition.py:86-96: an interval that starts before its predecessor ends is cut back to start one past it"""
    for i in range(len(ilist) - 1):
        if ilist[i][1] > ilist[i + 1][0]:
            ilist[i + 1] = (ilist[i][1] + 1, ilist[i + 1][1])


def lcb_interval(lcb):
    """get_interval, :64-83: (reference contig index, (start, end)) of the block's first record"""
    rec = lcb[0]
    aln_len = rec.end - rec.start
    _, contig, startpos = [int(x) for x in _ID.match(rec.id).groups()]
    if rec.strand == -1:
        return contig, (startpos - aln_len, startpos)
    return contig, (startpos, startpos + aln_len)


def chunk_intervals(xmfa_path):
    """{contig: sorted, overlap-cut [(start, end)]} of one partition's XMFA (get_chunked_intervals, :507-536)"""
    d = defaultdict(list)
    for lcb in read_lcbs(xmfa_path):
        contig, iv = lcb_interval(lcb)
        d[contig].append(iv)
    for ivs in d.values():
        ivs.sort()
        cut_overlaps(ivs)
    return d


def intersected_intervals(per_chunk, min_interval_size=10):
    """get_intersected_intervals, :539-583.  per_chunk: list of {contig: intervals}"""
    cur = {k: [tuple(x) for x in v] for k, v in per_chunk[0].items()}
    for d in per_chunk:
        for contig in set(cur) | set(d):
            cur[contig] = interval_intersection(cur.get(contig, []), d.get(contig, []))
    return {k: [iv for iv in v if iv[1] - iv[0] >= min_interval_size] for k, v in cur.items()}


# ---------------------------------------------------------------------------------------------- trimming
def _prefix_bases(seq):
    ps = [0] * (len(seq) + 1)
    for i, ch in enumerate(seq, 1):
        ps[i] = ps[i - 1] + (0 if ch == "-" else 1)
    return ps


def trim_lcb(lcb, ref_intervals, seqidx, cluster_start):
    """trim, :99-216: cut one block into the pieces whose reference coordinates are the given intervals"""
    ref = next((r for r in lcb if r.name == seqidx), None)
    if ref is None:
        raise ValueError("Reference alignment not found!")
    aln_len = ref.end - ref.start
    _, contig, super_start = [int(x) for x in _ID.match(ref.id).groups()]
    if ref.strand == -1:
        super_start, super_end = super_start - aln_len, super_start
    else:
        super_end = super_start + aln_len
    pieces = interval_intersection(ref_intervals.get(contig, []), [(super_start, super_end)])
    ncol = len(ref.seq)
    ref_psum = _prefix_bases(ref.seq)
    ref_ssum = _prefix_bases(ref.seq[::-1])
    psum = [_prefix_bases(r.seq[:ncol]) for r in lcb]
    ssum = [_prefix_bases(r.seq[::-1][:ncol]) for r in lcb]
    out = []
    for k, (a, b) in enumerate(pieces):
        left_cols = bisect.bisect_left(ref_psum, a - super_start)
        right_cols = bisect.bisect_left(ref_ssum, super_end - b)
        block = []
        for x, rec in enumerate(lcb):
            _, rcontig, startpos = [int(v) for v in _ID.match(rec.id).groups()]
            lb, rb = psum[x][left_cols], ssum[x][right_cols]
            new = rec.copy(rec.seq[left_cols:-right_cols] if right_cols > 0 else rec.seq[left_cols:])
            if rec.strand == -1:
                new.start += rb; new.end -= lb; startpos -= lb
            else:
                new.start += lb; new.end -= rb; startpos += lb
            new.id = "cluster%d s%d:p%d" % (cluster_start + k, rcontig, startpos)
            block.append(new)
        out.append(block)
    return out


def write_lcb(block, fh):
    """write_aln_to_fna, :231-243"""
    for rec in block:
        fh.write("> %s:%d-%d %s %s\n" % (rec.name, rec.start + 1, rec.end, "+" if rec.strand == 1 else "-", rec.id))
        for i in range(math.ceil(len(rec.seq) / 80)):
            fh.write(rec.seq[i * 80:(i + 1) * 80] + "\n")


def trim_xmfa(xmfa_path, ref_intervals):
    """trim_single_xmfa, :586-618 -> number of clusters written to <xmfa>.trimmed"""
    out = xmfa_path + ".trimmed"
    cluster_start = 1
    with open(xmfa_path) as fin, open(out, "w") as fout:
        for line in fin:
            if line.startswith("#"):
                fout.write(line)
            else:
                break
        for lcb in read_lcbs(xmfa_path):
            for block in trim_lcb(lcb, ref_intervals, 1, cluster_start):
                write_lcb(block, fout)
                fout.write("=\n")
                cluster_start += 1
    return cluster_start - 1


# ---------------------------------------------------------------------------------------------- merging
def combine_header_info(xmfa_list):
    """:245-292 -> ({(index, file, header, length): new index}, {(xmfa, old index): new index}); a (file, header) pair
    that was seen before (the reference, present in every partition) is not added again"""
    fidx_to_new, seq_to_idx, seen = {}, {}, set()
    for x in xmfa_list:
        for entry in read_header(x):
            if (entry[1], entry[2]) not in seen:
                seen.add((entry[1], entry[2]))
                fidx_to_new[(x, entry[0])] = len(fidx_to_new) + 1
                seq_to_idx[entry] = fidx_to_new[(x, entry[0])]
    return seq_to_idx, fidx_to_new


def write_combined_header(seq_to_idx, cluster_count, out_path):
    """:295-318"""
    with open(out_path, "w") as f:
        f.write("#FormatVersion Mauve\n")
        f.write("#SequenceCount %d\n" % len(seq_to_idx))
        for entry in sorted(seq_to_idx, key=lambda k: seq_to_idx[k]):
            f.write("##SequenceIndex %d\n##SequenceFile %s\n##SequenceHeader %s\n##SequenceLength %dbp\n" % (seq_to_idx[entry], entry[1], entry[2], entry[3]))
        f.write("#IntervalCount %d\n" % cluster_count)


_GAP_LIB = None


def _gap_aligner():
    """this project's inter-MUM gap aligner (host restatement of libMUSCLE) from libparsnp_core.so, or None"""
    global _GAP_LIB
    if _GAP_LIB is None:
        _GAP_LIB = False
        try:
            here = os.path.dirname(os.path.abspath(__file__))
            lib = ctypes.CDLL(os.environ.get("PARSNP_GAP_LIB") or os.path.join(here, "..", "tests", "emu", "libgapalign.so"))
            lib.parsnp_gap_align.restype = ctypes.c_long
            lib.parsnp_gap_align.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_long]
            _GAP_LIB = lib
        except (OSError, AttributeError):
            pass
    return _GAP_LIB or None


def align_insertions(seqs):
    """stand-in for spoa.poa(seqs)[1] (:386): rows of equal length holding the given strings.  The reference's partial-order
    alignment is not reproducible without SPOA; see the module docstring."""
    if len(seqs) > 1 and all(seqs):
        lib = _gap_aligner()
        if lib is not None:
            cap = (1 << 16) + 4 * sum(len(s) for s in seqs) * (len(seqs) + 1)
            buf = ctypes.create_string_buffer(cap)
            if lib.parsnp_gap_align("\n".join(seqs).encode(), buf, cap) >= 0:
                return buf.value.decode().split("\n")[:-1]
    width = max(len(s) for s in seqs)
    return [s + "-" * (width - len(s)) for s in seqs]


def merge_blocks(blocks, fidx_to_new, aligner=align_insertions):
    """merge_blocks, :320-433.  blocks: [(lcb, xmfa path)] of the SAME trimmed cluster, one per partition.  Columns in which
    every partition's reference row holds a base are concatenated partition after partition; runs of columns in which some
    reference row holds a gap are insertions: their bases are collected per sequence and re-aligned among themselves."""
    first, first_file = blocks[0]
    combined, name_to_idx, col = [], {}, {}
    for bi, (lcb, xf) in enumerate(blocks):
        for rec in (lcb if bi == 0 else lcb[1:]):
            new = rec.copy("")
            new.name = fidx_to_new[(xf, rec.name)]
            new.id = new.id.split("/")[0]
            name_to_idx[new.name] = len(combined)
            combined.append(new)
        col[xf] = 0
    parts = [[] for _ in combined]
    sorted_names = sorted(name_to_idx)
    gap_sequences = defaultdict(list)
    all_done = False
    while not all_done:
        in_gap, all_done = False, True
        for lcb, xf in blocks:
            c = col[xf]
            if c >= len(lcb[0].seq) or lcb[0].seq[c] == "-":
                in_gap = True
            if c < len(lcb[0].seq):
                all_done = False
        if (not in_gap or all_done) and gap_sequences:
            names = list(gap_sequences)
            rows = aligner(["".join(gap_sequences[n]) for n in names])
            width = max(len(r) for r in rows)
            where = {n: i for i, n in enumerate(names)}
            for n in sorted_names:
                parts[name_to_idx[n]].append(rows[where[n]] if n in where else "-" * width)
            gap_sequences = defaultdict(list)
        elif not in_gap and not all_done:
            for bi, (lcb, xf) in enumerate(blocks):
                c = col[xf]
                for rec in (lcb if bi == 0 else lcb[1:]):
                    parts[name_to_idx[fidx_to_new[(xf, rec.name)]]].append(rec.seq[c])
                col[xf] += 1
        elif not all_done:
            for bi, (lcb, xf) in enumerate(blocks):
                c = col[xf]
                while c < len(lcb[0].seq) and lcb[0].seq[c] == "-":
                    for rec in (lcb if bi == 0 else lcb[1:]):
                        if c < len(rec.seq) and rec.seq[c] != "-":
                            gap_sequences[fidx_to_new[(xf, rec.name)]].append(rec.seq[c])
                    c += 1
                col[xf] = c
    for rec, p in zip(combined, parts):
        rec.seq = "".join(p)
    return combined


def merge_xmfas(out_path, trimmed_xmfas, num_clusters, aligner=align_insertions):
    """merge_xmfas, :683-736"""
    seq_to_idx, fidx_to_new = combine_header_info(trimmed_xmfas)
    write_combined_header(seq_to_idx, num_clusters, out_path)
    iters = [read_lcbs(x) for x in trimmed_xmfas]
    with open(out_path, "a") as f:
        for _ in range(num_clusters):
            merged = merge_blocks([(next(it), x) for it, x in zip(iters, trimmed_xmfas)], fidx_to_new, aligner)
            write_lcb(merged, f)
            f.write("=\n")
    return len(seq_to_idx)


def merge_partitions(partition_xmfas, out_path, min_interval_size=10, aligner=align_insertions):
    """the whole of partition.py's post-processing for the given per-partition XMFA files -> merged XMFA at out_path.
    Returns dict(intervals=..., clusters=..., sequences=...)"""
    per_chunk = [chunk_intervals(x) for x in partition_xmfas]
    inter = intersected_intervals(per_chunk, min_interval_size)
    counts = {trim_xmfa(x, inter) for x in partition_xmfas}
    if len(counts) != 1:
        raise RuntimeError("One of the partitions has a different number of clusters after trimming...")   # partition.py:644-646
    n = counts.pop()
    nseq = merge_xmfas(out_path, [x + ".trimmed" for x in partition_xmfas], n, aligner)
    return dict(intervals=inter, clusters=n, sequences=nseq)
