import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "parsnp_amd")
CSRC = os.path.join(PKG, "csrc")
BIN_DIR = os.path.join(PKG, "bin")
LIB_DIR = os.path.join(PKG, "lib")
CORE_BIN = os.path.join(BIN_DIR, "parsnp_core")
HIP_LIB = os.environ.get("PARSNP_HIP_LIB") or os.path.join(LIB_DIR, "libparsnp_hip.so")      # (PARSNP_HIP_LIB: a measurement build of the library, scripts/gap_nomark.sh)
CORE_HOOKS_BIN = os.path.join(BIN_DIR, "parsnp_core_hooks")   # the same sources with the test hooks compiled in (csrc/host/hooks.h)
