"""Partition mode's merge step (include/parsnp_merge.h): the per-partition alignments -> one parsnp.xmfa.  Native host
code in libparsnp_core.so (parsnp_amd/csrc/host/partition_merge.cpp = the reference driver's partition.py:35-61, 86-216,
245-433, 507-736); this is the ctypes stub a maintainer of the reference driver would put in place of the four
partition.py calls at parsnp:1601-1615."""
import ctypes as C
import os

from .paths import LIB_DIR

CORE_LIB = os.path.join(LIB_DIR, "libparsnp_core.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        path = os.environ.get("PARSNP_CORE_LIB") or CORE_LIB
        if not os.path.exists(path):
            raise RuntimeError("%s not built (python -c 'import __graft_entry__ as g; g.build()')" % path)
        lib = C.CDLL(path)
        lib.parsnp_partition_merge.restype = C.c_int
        lib.parsnp_partition_merge.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_long, C.c_int, C.c_int,
                                               C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long), C.c_char_p, C.c_long]
        _lib = lib
    return _lib


def merge_partitions(partition_xmfas, out_path, min_interval_size=10, threads=None, keep_trimmed=False):
    """-> dict(clusters=, sequences=, ref_bases=); raises RuntimeError with the library's message on failure"""
    lib = _load()
    if threads is None:      # the CPUs this process may really use: affinity mask, capped by the container's CPU quota
        threads = len(os.sched_getaffinity(0))
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if quota != "max":
                threads = min(threads, int(int(quota) / int(period)))
        except (OSError, ValueError):
            pass
        threads = max(1, min(16, threads))
    arr = (C.c_char_p * len(partition_xmfas))(*[p.encode() for p in partition_xmfas])
    clusters, sequences, bases = C.c_long(0), C.c_long(0), C.c_long(0)
    err = C.create_string_buffer(1024)
    rc = lib.parsnp_partition_merge(len(partition_xmfas), arr, out_path.encode(), min_interval_size, threads, 1 if keep_trimmed else 0,
                                    C.byref(clusters), C.byref(sequences), C.byref(bases), err, len(err))
    if rc:
        raise RuntimeError("partition merge failed: " + err.value.decode(errors="replace"))
    ins = [C.c_long(0) for _ in range(4)]
    lib.parsnp_partition_merge_insertions(*[C.byref(x) for x in ins])
    # insertions: where the merged file could depend on the insertion aligner (SPOA in the reference)
    return dict(clusters=clusters.value, sequences=sequences.value, ref_bases=bases.value,
                insertions=dict(runs=ins[0].value, shared=ins[1].value, shared_diverse=ins[2].value, shared_columns=ins[3].value))
