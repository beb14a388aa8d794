"""Partition mode across the GPUs of one node (BASELINE config 4): the reference driver splits the query genomes into
chunks of ~min_partition_size and runs one independent parsnp_core per chunk (parsnp:1553-1597, a multiprocessing
Pool over OS processes).  Here every rank of a torch.distributed job owns one GPU and takes the chunks
rank, rank+world, ...; partitions never talk to each other while they run (no data-path collective).  Afterwards the
per-partition reference intervals of the LCBs are all-gathered (objects; RCCL/gloo) so that every rank holds the
intersection the reference's partition.py starts its merge from (partition.py:35-61, 539-583).

Rank 0 then merges the partitions' XMFA files into <outdir>/parsnp.xmfa (parsnp_amd.merge -> the native
parsnp_partition_merge of include/parsnp_merge.h: interval intersection, trimming, block merge -- partition.py:507-736)."""
import math
import os
import re

from . import driver

CHUNK_PREFIX = "chunk"


def plan_partitions(finalfiles, min_partition_size):
    """chunk lists exactly as the driver builds them (parsnp:1555-1564)"""
    full = len(finalfiles) // min_partition_size
    size = len(finalfiles) // full
    return [finalfiles[i * size:(i + 1) * size] for i in range(math.ceil(len(finalfiles) / size))]


def ref_intervals(xmfa_path):
    """[(start, end)] of the reference record of every LCB, 1-based inclusive as printed ('> 1:a-b')"""
    import mmap
    with open(xmfa_path, "rb") as f:
        if os.fstat(f.fileno()).st_size == 0:
            return []
        with mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as mm:      # 1.2 GB per partition at 250 x 5 Mb: one pass in C
            return [(int(m.group(1)), int(m.group(2))) for m in re.finditer(rb"^> 1:(\d+)-(\d+) ", mm, re.M)]


def intersect(interval_lists):
    """reference positions covered by an LCB in EVERY partition, as sorted disjoint intervals"""
    cur = None
    for ivs in interval_lists:
        ivs = sorted(ivs)
        if cur is None:
            cur = ivs
            continue
        out, i, j = [], 0, 0
        while i < len(cur) and j < len(ivs):
            a, b = max(cur[i][0], ivs[j][0]), min(cur[i][1], ivs[j][1])
            if a <= b:
                out.append((a, b))
            if cur[i][1] < ivs[j][1]:
                i += 1
            else:
                j += 1
        cur = out
    return cur or []


def run_partitioned(core_bin, ref, finalfiles, outdir, min_partition_size, rank=0, world=1, dist=None, local_rank=None, merge=True, keep_trimmed=False,
                    **ini_kw):
    """-> dict(partitions=[...], intersection=[...]) on every rank.  `dist`: an initialised torch.distributed module (or
    None for a single process)."""
    chunks = plan_partitions(finalfiles, min_partition_size)
    mine = []
    env = dict(os.environ)
    if local_rank is not None:
        env["HIP_VISIBLE_DEVICES"] = str(local_rank)   # one process per GPU
    for idx in range(rank, len(chunks), world):
        cdir = os.path.join(outdir, "partition", "%s-%010d-out" % (CHUNK_PREFIX, idx))
        rc, _ = driver.run_core(core_bin, ref, chunks[idx], cdir, env=env, **ini_kw)
        x = os.path.join(cdir, "parsnpAligner.xmfa")
        ok = rc == 0 and os.path.exists(x)
        mine.append(dict(index=idx, rc=rc, ok=ok, queries=len(chunks[idx]), dir=cdir, intervals=ref_intervals(x) if ok else []))
    if dist is not None and world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        parts = sorted((p for g in gathered for p in g), key=lambda p: p["index"])
    else:
        parts = mine
    good = [p for p in parts if p["ok"]]   # failed partitions are dropped, the rest merged (parsnp:1594-1599)
    merged = None
    if merge and rank == 0 and good:
        from . import merge as native_merge
        merged = native_merge.merge_partitions([os.path.join(p["dir"], "parsnpAligner.xmfa") for p in good], os.path.join(outdir, "parsnp.xmfa"),
                                               keep_trimmed=keep_trimmed)
        merged = dict(clusters=merged["clusters"], sequences=merged["sequences"], ref_bases=merged["ref_bases"], insertions=merged["insertions"], xmfa=os.path.join(outdir, "parsnp.xmfa"))
    if merge and dist is not None and world > 1:      # every rank returns the same view
        box = [merged]
        dist.broadcast_object_list(box, src=0)
        merged = box[0]
    return dict(partitions=parts, intersection=intersect([p["intervals"] for p in good]) if good else [], merged=merged)
