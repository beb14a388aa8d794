"""ctypes binding of the C ABI in include/parsnp_mum.h.

`load(path=None)` opens parsnp_amd/lib/libparsnp_hip.so (the product) by default and raises if it is missing: there is
no CPU fallback on the product side.  Tests pass an explicit path to open their CPU checkers through the same
binding."""
import ctypes as C
import os

import numpy as np

from .paths import HIP_LIB


class PmError(RuntimeError):
    pass


class Lib:
    def __init__(self, path=None):
        path = path or HIP_LIB
        if not os.path.exists(path):
            raise PmError("engine library %s not built (python -c 'import __graft_entry__ as g; g.build()')" % path)
        self.path = path
        L = self.L = C.CDLL(path)
        L.pm_last_error.restype = C.c_char_p
        L.pm_provider.restype = C.c_char_p
        L.pm_session_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64)]
        L.pm_session_destroy.argtypes = [C.c_void_p]
        L.pm_multi_mum_batch.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                         C.POINTER(C.c_void_p)]
        for name, t in (("pm_result_regions", C.c_int64), ("pm_result_total", C.c_int64), ("pm_result_offsets", C.POINTER(C.c_int64)),
                        ("pm_result_k", C.POINTER(C.c_int32)), ("pm_result_lon", C.POINTER(C.c_int32)),
                        ("pm_result_sp", C.POINTER(C.c_int32)), ("pm_result_fwd", C.POINTER(C.c_uint8))):
            getattr(L, name).restype = t
            getattr(L, name).argtypes = [C.c_void_p]
        L.pm_result_free.argtypes = [C.c_void_p]
        L.pm_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_float)]

    @property
    def provider(self):
        return self.L.pm_provider().decode()

    def _check(self, rc):
        if rc != 0:
            raise PmError("%s (code %d)" % (self.L.pm_last_error().decode(), rc))

    def rccl_unique_id(self) -> bytes:
        ident = (C.c_uint8 * 128)()
        self._check(self.L.pm_rccl_unique_id(ident))
        return bytes(ident)

    def find_events(self, ref: bytes, query: bytes, min_len: int, strand: int = 0):
        cap = len(query) + 16
        j = np.zeros(cap, np.int64); l = np.zeros(cap, np.int64); ln = np.zeros(cap, np.int32); rp = np.zeros(cap, np.int32)
        cnt = C.c_int64()
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
        self._check(self.L.pm_find_events(ref, C.c_int64(len(ref)), query, C.c_int64(len(query)), C.c_int32(min_len), C.c_int(strand),
                                          C.c_int64(cap), C.byref(cnt), p(j, C.c_int64), p(l, C.c_int64), p(ln, C.c_int32), p(rp, C.c_int32)))
        c = cnt.value
        return j[:c].copy(), l[:c].copy(), ln[:c].copy(), rp[:c].copy()


class Session:
    """genomes resident on the device; genome 0 is the reference"""

    def __init__(self, lib: Lib, seqs, device=-1, rccl=None):
        """rccl = (rank, world, id bytes): a sharded session whose exchanges run over the engine's own RCCL communicator
        (pm_session_create_rccl); `Lib.rccl_unique_id()` makes the id on rank 0"""
        self.lib = lib
        self.n = len(seqs)
        self._seqs = [bytes(s) for s in seqs]
        arr = (C.c_char_p * self.n)(*self._seqs)
        lens = (C.c_int64 * self.n)(*[len(s) for s in self._seqs])
        h = C.c_void_p()
        if rccl is not None:
            rank, world, ident = rccl
            lib.L.pm_session_create_rccl.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64),
                                                     C.c_int, C.c_int, C.POINTER(C.c_uint8)]
            lib._check(lib.L.pm_session_create_rccl(C.byref(h), device, self.n, arr, lens, rank, world, (C.c_uint8 * 128).from_buffer_copy(ident)))
        else:
            lib._check(lib.L.pm_session_create(C.byref(h), device, self.n, arr, lens))
        self.h = h

    def close(self):
        if self.h:
            self.lib.L.pm_session_destroy(self.h)
            self.h = None

    def tune(self, key: str, value: int):
        """pm_session_tune: "work_budget", "dirty_min" (include/parsnp_mum.h)"""
        self.lib.L.pm_session_tune.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        self.lib._check(self.lib.L.pm_session_tune(self.h, key.encode(), value))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def multi_mum_batch(self, starts, lens, minsize):
        """starts, lens: [n_regions, n_genomes]; minsize: [n_regions] -> list of (k, lon, sp[c, n-1], fwd[c, n-1]) per region"""
        starts = np.ascontiguousarray(starts, np.int64); lens = np.ascontiguousarray(lens, np.int64)
        minsize = np.ascontiguousarray(minsize, np.int32)
        nreg = starts.shape[0]
        assert starts.shape == (nreg, self.n) == lens.shape and minsize.shape == (nreg,)
        res = C.c_void_p()
        L = self.lib.L
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
        self.lib._check(L.pm_multi_mum_batch(self.h, nreg, p(starts, C.c_int64), p(lens, C.c_int64), p(minsize, C.c_int32), C.byref(res)))
        try:
            total = L.pm_result_total(res); q = self.n - 1
            off = np.ctypeslib.as_array(L.pm_result_offsets(res), (nreg + 1,)).copy()
            if total:
                k = np.ctypeslib.as_array(L.pm_result_k(res), (total,)).copy()
                lon = np.ctypeslib.as_array(L.pm_result_lon(res), (total,)).copy()
                sp = np.ctypeslib.as_array(L.pm_result_sp(res), (total * q,)).astype(np.int64).reshape(total, q)
                fw = np.ctypeslib.as_array(L.pm_result_fwd(res), (total * q,)).copy().reshape(total, q)
            else:
                k = np.zeros(0, np.int32); lon = np.zeros(0, np.int32); sp = np.zeros((0, q), np.int64); fw = np.zeros((0, q), np.uint8)
        finally:
            L.pm_result_free(res)
        return [(k[off[r]:off[r + 1]].astype(np.int64), lon[off[r]:off[r + 1]], sp[off[r]:off[r + 1]], fw[off[r]:off[r + 1]]) for r in range(nreg)]

    def whole(self, minsize):
        """the anchor call: one region = every genome in full"""
        lens = np.array([[len(s) for s in self._seqs]], np.int64)
        return self.multi_mum_batch(np.zeros_like(lens), lens, [minsize])[0]

    def mumi_coverage(self):
        """calcmumi: reference positions covered by pairwise MUMs >= 15, per query genome (whole genomes)"""
        lens = (C.c_int64 * self.n)(*[len(s) for s in self._seqs])
        starts = (C.c_int64 * self.n)(*([0] * self.n))
        cov = (C.c_int64 * max(1, self.n - 1))()
        self.lib.L.pm_mumi_coverage.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        self.lib._check(self.lib.L.pm_mumi_coverage(self.h, starts, lens, cov))
        return [int(cov[i]) for i in range(self.n - 1)]

    def last_timing(self):
        cnt = C.c_int(64); names = (C.c_char_p * 64)(); ms = (C.c_float * 64)()
        self.lib.L.pm_last_timing(self.h, C.byref(cnt), names, ms)
        return [(names[i].decode(), float(ms[i])) for i in range(cnt.value)]
