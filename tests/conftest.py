import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

EMU_LIB = os.path.join(ROOT, "tests", "emu", "libpm_emu.so")
EMU_CORE = os.path.join(ROOT, "tests", "emu", "parsnp_core_emu")
ORACLE_CORE = os.path.join(ROOT, "oracle", "_ref", "parsnp_core_oracle")
REF_CORE = os.path.join(ROOT, "oracle", "_ref", "parsnp_core_ref")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources if os.path.exists(s))


@pytest.fixture(scope="session")
def cpu_checkers():
    """oracle restatement + CPU provider + host binary linked to it (test infrastructure, built on demand)"""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "restatement", "hosttest"], check=True)
    return ORACLE_CORE


@pytest.fixture(scope="session")
def emu():
    """sequential host execution of the engine's kernel functors (tests/emu), for logic checks without a GPU"""
    eng = os.path.join(ROOT, "parsnp_amd", "csrc", "engine")
    host = os.path.join(ROOT, "parsnp_amd", "csrc", "host")
    src = os.path.join(ROOT, "tests", "emu", "engine_emu.cpp")
    deps = [src] + [os.path.join(eng, f) for f in os.listdir(eng)] + [os.path.join(ROOT, "include", "parsnp_mum.h")]
    if not _newer(EMU_LIB, deps):
        # a tiny scan chunk makes the cross-chunk look-back of the scan kernels run on small test inputs
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-w", "-DPM_WAVE_EVENTS=5", src, "-o", EMU_LIB], check=True)
    hsrc = [os.path.join(host, f) for f in os.listdir(host) if f.endswith(".cpp") and f not in ("capi.cpp", "merge_main.cpp")]
    if not _newer(EMU_CORE, hsrc + [os.path.join(host, f) for f in os.listdir(host)] + [EMU_LIB]):
        subprocess.run(["g++", "-O3", "-mavx2", "-std=c++17", "-fopenmp", "-w", "-DPARSNP_TEST_HOOKS"] + hsrc + ["-L" + os.path.dirname(EMU_LIB), "-lpm_emu",
                        "-Wl,-rpath,$ORIGIN", "-o", EMU_CORE], check=True)
    core_lib = os.path.join(ROOT, "tests", "emu", "libparsnp_core_emu.so")
    lsrc = [os.path.join(host, f) for f in os.listdir(host) if f.endswith(".cpp") and f not in ("main.cpp", "merge_main.cpp")]
    if not _newer(core_lib, lsrc + [os.path.join(host, f) for f in os.listdir(host)] + [EMU_LIB]):
        subprocess.run(["g++", "-O3", "-mavx2", "-std=c++17", "-fopenmp", "-fPIC", "-shared", "-w", "-DPARSNP_TEST_HOOKS"] + lsrc +
                       ["-L" + os.path.dirname(EMU_LIB), "-lpm_emu", "-Wl,-rpath,$ORIGIN", "-o", core_lib], check=True)
    return EMU_LIB, EMU_CORE
