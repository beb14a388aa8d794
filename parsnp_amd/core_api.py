"""In-process handle on the parsnp_core host path (parsnp_amd/lib/libparsnp_core.so): open once (ingest + upload),
step phases A-D (anchor MUMs, recursive extension, LCB formation) any number of times, write XMFA/log."""
import ctypes as C
import json
import os

from .paths import LIB_DIR

CORE_LIB = os.path.join(LIB_DIR, "libparsnp_core.so")


class CoreRun:
    def __init__(self, ini_path, lib_path=None):
        path = lib_path or CORE_LIB
        if not os.path.exists(path):
            raise RuntimeError("%s not built (python -c 'import __graft_entry__ as g; g.build()')" % path)
        self.L = C.CDLL(path)
        self.L.pc_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        self.L.pc_step.argtypes = [C.c_void_p]
        self.L.pc_step.restype = C.c_char_p
        self.L.pc_step_brief.argtypes = [C.c_void_p]
        self.L.pc_step_brief.restype = C.c_char_p
        self.L.pc_write.argtypes = [C.c_void_p]
        self.L.pc_close.argtypes = [C.c_void_p]
        h = C.c_void_p()
        rc = self.L.pc_open(ini_path.encode(), C.byref(h))
        if rc:
            raise RuntimeError("parsnp_core could not start (exit code %d)" % rc)
        self.h = h

    def step(self, intervals=True):
        """one pass of phases A-D -> the report; intervals=False leaves out `lcb_ref_intervals` (partition mode's exchange reads them)"""
        return json.loads((self.L.pc_step if intervals else self.L.pc_step_brief)(self.h).decode())

    def write(self):
        return self.L.pc_write(self.h)

    def close(self):
        if self.h:
            self.L.pc_close(self.h)
            self.h = None
