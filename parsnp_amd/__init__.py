"""parsnp_amd -- MI355X-native replacement for the MUM-finding / recursive-extension / LCB hot path of marbl/parsnp.

The product is the `parsnp_core` binary (parsnp_amd/bin/parsnp_core) and the C-ABI engine behind it
(parsnp_amd/lib/libparsnp_hip.so, declared in include/parsnp_mum.h).  This package holds the build recipe, a ctypes
binding of the C ABI, the .ini writer that mirrors the reference's Python driver, and the synthetic-input generators
used by tests and bench.py."""
from .paths import BIN_DIR, LIB_DIR, ROOT  # noqa: F401
