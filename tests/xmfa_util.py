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
