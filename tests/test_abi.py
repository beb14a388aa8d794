"""The C-ABI library loads and exports every symbol include/parsnp_mum.h declares (no compute calls: no GPU here)."""
import ctypes
import os
import re

import pytest

from parsnp_amd.paths import HIP_LIB, ROOT


def declared():
    txt = open(os.path.join(ROOT, "include", "parsnp_mum.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pm_[a-z_]+)\s*\(", txt)))


def test_header_declares_the_surface():
    d = declared()
    for name in ("pm_session_create", "pm_session_destroy", "pm_multi_mum_batch", "pm_result_free", "pm_find_events", "pm_last_error"):
        assert name in d


def test_hip_library_exports_every_declared_symbol():
    if not os.path.exists(HIP_LIB):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(HIP_LIB)
    for name in declared():
        assert hasattr(lib, name), name
    lib.pm_provider.restype = ctypes.c_char_p
    assert lib.pm_provider() == b"hip"


def test_product_has_no_cpu_path(tmp_path):
    """without a GPU the product must fail loudly, not fall back"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from parsnp_amd.binding import Lib, PmError, Session
    with pytest.raises(PmError):
        Session(Lib(HIP_LIB), [b"ACGT", b"ACGT"])


def test_bench_and_core_refuse_to_run_without_a_gpu(tmp_path):
    """bench.py and parsnp_core exit non-zero with a message when no HIP device is present (parsnp_core: code 3)"""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--cpu-sample", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU path" in (r.stdout + r.stderr)
    from parsnp_amd import driver, synth
    from parsnp_amd.paths import CORE_BIN
    ref, gs = synth.make("pop6x200k", n=5000, n_genomes=2)
    rp, qs = synth.write_set(str(tmp_path / "in"), ref, gs)
    rc, _ = driver.run_core(CORE_BIN, rp, qs, str(tmp_path / "out"))
    assert rc == 3
    assert not os.path.exists(str(tmp_path / "out" / "parsnpAligner.xmfa"))


def test_product_does_not_reference_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "parsnp_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                if re.search(r"oracle/|mum_oracle|pm_oracle|libpm_emu|tests/emu", open(os.path.join(base, f), errors="ignore").read()):
                    bad.append(os.path.join(base, f))
    # comments in headers may MENTION the test providers; nothing may include, link or load them
    for p in bad:
        txt = open(p, errors="ignore").read()
        assert not re.search(r'#include\s*"[^"]*(oracle|emu)', txt), p
        assert not re.search(r"(CDLL|dlopen|-l)\s*\(?[\"']?[^\n]*(pm_oracle|mum_oracle|pm_emu)", txt), p


def test_gap_groups_argument_checks():
    """pm_gap_align_groups before it touches a device: inconsistent group boundaries are PM_EINVAL; an empty batch reports
    every group at once and succeeds"""
    lib = ctypes.CDLL(HIP_LIB)
    lib.pm_gap_align_groups.restype = ctypes.c_int
    CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int)
    seen = []
    cb = CB(lambda ctx, g: seen.append(g))
    i32 = (ctypes.c_int32 * 4)()
    i64 = (ctypes.c_int64 * 4)()
    u8 = (ctypes.c_uint8 * 4)()
    args = lambda n_jobs, ends: (ctypes.c_int(-1), ctypes.c_int64(n_jobs), i32, i64, u8, i32, i64, u8, ctypes.c_int64(4), i32,   # noqa: E731
                                 ctypes.c_int(len(ends)), (ctypes.c_int64 * len(ends))(*ends), cb, None)
    assert lib.pm_gap_align_groups(*args(2, [1, 3])) == -2 and seen == []           # the last boundary is not n_jobs (PM_EINVAL)
    assert lib.pm_gap_align_groups(*args(2, [2, 1, 2])) == -2 and seen == []        # boundaries go backwards
    assert lib.pm_gap_align_groups(*args(0, [0, 0, 0])) == 0 and seen == [0, 1, 2]  # nothing to align: every group reported
