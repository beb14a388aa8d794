"""ctypes bindings for the CHECKERS used by tests: oracle/_ref/libmum_oracle.so (our CPU restatement) and,
when it has been built from /root/reference, oracle/_ref/libcsgmum_ref.so (the reference's own csgmum code).
Test infrastructure only -- nothing under parsnp_amd/ imports this."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")

_COMP = bytes.maketrans(b"ACGTUacgtu", b"TGCAATGCAA")


def revcomp(s: bytes) -> bytes:
    """Aligner::reversec (src/parsnp.cpp:1294-1393) on ingested symbols: ACGT complemented, everything else N."""
    t = bytes((c if c in b"ACGTUacgtu" else ord("N")) for c in s).translate(_COMP)
    return t[::-1]


def _build(target):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), target], check=True)


def load_restatement():
    path = os.path.join(REFDIR, "libmum_oracle.so")
    if not os.path.exists(path):
        _build("restatement")
    lib = C.CDLL(path)
    lib.oracle_min_length.restype = C.c_int32
    lib.oracle_min_length.argtypes = [C.c_char_p, C.c_int64]
    lib.oracle_events.restype = C.c_int64
    return lib


def have_reference():
    return os.path.exists(os.path.join(REFDIR, "libcsgmum_ref.so"))


def load_reference():
    lib = C.CDLL(os.path.join(REFDIR, "libcsgmum_ref.so"))
    return lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def restatement_find_um(lib, ref: bytes, q: bytes, min_len=1, propagate=False):
    n = len(ref)
    UP = np.zeros(n, np.int32); EP = np.zeros(n, np.int32); SP = np.zeros(n, np.int64)
    rc = lib.oracle_find_um(ref, C.c_int64(n), q, C.c_int64(len(q)), C.c_int(min_len),
                            _p(UP, C.c_int32), _p(EP, C.c_int32), _p(SP, C.c_int64))
    assert rc == 0
    if propagate:
        lib.oracle_propagate(C.c_int64(n), _p(UP, C.c_int32), _p(EP, C.c_int32), _p(SP, C.c_int64))
    return UP, EP, SP


def restatement_events(lib, ref: bytes, q: bytes, min_len=1):
    cap = len(q) + 1
    j = np.zeros(cap, np.int64); l = np.zeros(cap, np.int64); ln = np.zeros(cap, np.int32); rp = np.zeros(cap, np.int32)
    c = lib.oracle_events(ref, C.c_int64(len(ref)), q, C.c_int64(len(q)), C.c_int(min_len), C.c_int64(cap),
                          _p(j, C.c_int64), _p(l, C.c_int64), _p(ln, C.c_int32), _p(rp, C.c_int32))
    return j[:c].copy(), l[:c].copy(), ln[:c].copy(), rp[:c].copy()


def reference_find_um(lib, ref: bytes, q: bytes, propagate=False, factor=2):
    n = len(ref)
    UP = np.zeros(n, np.int32); EP = np.zeros(n, np.int32); SP = np.zeros(n, np.uint64)
    fn = lib.ref_find_um_propagated if propagate else lib.ref_find_um
    rc = fn(ref, C.c_long(n), q, C.c_long(len(q)), C.c_int(factor), _p(UP, C.c_int32), _p(EP, C.c_int32), _p(SP, C.c_ulong))
    assert rc == 0
    return UP, EP, SP.astype(np.int64)


def _collect(c, q, pk, pl, ps, pf, free, sp_t):
    c = int(c)
    k = np.ctypeslib.as_array(pk, (max(c, 1),))[:c].astype(np.int64).copy()
    lon = np.ctypeslib.as_array(pl, (max(c, 1),))[:c].astype(np.int32).copy()
    sp = np.ctypeslib.as_array(ps, (max(c * q, 1),))[:c * q].astype(np.int64).reshape(c, q).copy()
    fw = np.ctypeslib.as_array(C.cast(pf, C.POINTER(C.c_uint8)), (max(c * q, 1),))[:c * q].astype(np.uint8).reshape(c, q).copy()
    for p in (pk, pl, ps, pf):
        free(p)
    return k, lon, sp, fw


def restatement_multi_mum(lib, seqs, minsize, min_event_len=1, want_master=False):
    cnt = len(seqs); q = cnt - 1
    arr = (C.c_char_p * cnt)(*seqs)
    lens = (C.c_int64 * cnt)(*[len(s) for s in seqs])
    c = C.c_int64(); pk = C.POINTER(C.c_int64)(); pl = C.POINTER(C.c_int32)(); ps = C.POINTER(C.c_int64)(); pf = C.POINTER(C.c_uint8)()
    n = len(seqs[0])
    mU = np.zeros(max(n, 1), np.int32); mE = np.zeros(max(n, 1), np.int32)
    rc = lib.oracle_multi_mum(C.c_int(cnt), arr, lens, C.c_int(minsize), C.c_int(min_event_len), C.byref(c), C.byref(pk),
                              C.byref(pl), C.byref(ps), C.byref(pf), _p(mU, C.c_int32), _p(mE, C.c_int32))
    assert rc == 0
    lib.oracle_free.argtypes = [C.c_void_p]
    out = _collect(c.value, q, pk, pl, ps, pf, lambda p: lib.oracle_free(C.cast(p, C.c_void_p)), C.c_int64)
    return out + ((mU[:n], mE[:n]) if want_master else ())


def reference_multi_mum(lib, seqs, minsize, factor=2, want_master=False):
    cnt = len(seqs); q = cnt - 1
    arr = (C.c_char_p * cnt)(*seqs)
    rcs = (C.c_char_p * cnt)(*[revcomp(s) for s in seqs])
    lens = (C.c_long * cnt)(*[len(s) for s in seqs])
    c = C.c_long(); pk = C.POINTER(C.c_long)(); pl = C.POINTER(C.c_int)(); ps = C.POINTER(C.c_ulong)(); pf = C.POINTER(C.c_char)()
    n = len(seqs[0])
    mU = np.zeros(max(n, 1), np.int32); mE = np.zeros(max(n, 1), np.int32)
    rc = lib.ref_multi_mum(C.c_int(cnt), arr, lens, rcs, C.c_int(minsize), C.c_int(factor), C.byref(c), C.byref(pk),
                           C.byref(pl), C.byref(ps), C.byref(pf), _p(mU, C.c_int32), _p(mE, C.c_int32))
    assert rc == 0
    lib.ref_free.argtypes = [C.c_void_p]
    out = _collect(c.value, q, pk, pl, ps, pf, lambda p: lib.ref_free(C.cast(p, C.c_void_p)), C.c_ulong)
    return out + ((mU[:n], mE[:n]) if want_master else ())
