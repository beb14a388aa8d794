#!/usr/bin/env python3
"""A longer side-by-side campaign than the committed fuzz tests, on the CPU checker (build container, needs
oracle/_ref/parsnp_core_ref): seeds beyond those of tests/test_fuzz_vs_reference.py, every run threaded, with the rarely
taken host routes forced (parallel validation of short lists, free/tangled split of the flagged candidates) and every
derived or early-exit seed region re-done with the full bitmap walk.
    python scripts/fuzz_campaign.py [first_small last_small [n_big]]      (default 24 160 14: ~15 min)
PARSNP_FUZZ_CORE=hip runs the PRODUCT's sources with the test hooks compiled in (parsnp_amd/bin/parsnp_core_hooks, on the GPU box) instead of the CPU checker, with
the device-side shortcuts forced onto the small sets as well (MUM rows + overlap flags from the device, prejudged chaining
verdicts; the inter-MUM gaps go through the device aligner anyway).  PARSNP_FUZZ_CORE=emu: the host code over the kernel
emulation, for the routes that need the engine's anchor table (requests by reference, the batch computed ahead, the layout
image) without a GPU."""
import os
import pathlib
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.chdir(ROOT)
import test_fuzz_vs_reference as F  # noqa: E402

core = os.path.join(ROOT, "oracle", "_ref", "parsnp_core_oracle")
os.environ.update(PARSNP_PARALLEL_MIN="2", PARSNP_FREE_MIN="1", PARSNP_CHECK_NEIGHBOURS="1")
if os.environ.get("PARSNP_FUZZ_CORE") == "hip":
    core = os.path.join(ROOT, "parsnp_amd", "bin", "parsnp_core_hooks")      # the product's sources with the test hooks compiled in
    os.environ.update(PM_DIRTY_MIN="2")
if os.environ.get("PARSNP_FUZZ_CORE") == "emu":
    # the host code over the kernel emulation (tests/emu/parsnp_core_emu, built by the test suite's `emu` fixture): the routes
    # that need the engine's anchor table -- requests by reference, the batch computed ahead, the layout image -- on the CPU
    core = os.path.join(ROOT, "tests", "emu", "parsnp_core_emu")
    os.environ.update(PM_DIRTY_MIN="2", PARSNP_CHECK_ZERO="1")
first, last = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (24, 160)
n_big = int(sys.argv[3]) if len(sys.argv) > 3 else 14
bad = []
routes = {}


def one(seed, big):
    base = pathlib.Path(tempfile.mkdtemp(prefix="fz_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None))
    try:
        try:
            ref, gs, kw, contigs = F.random_case(seed, big)
        except ValueError as e:      # (a generator corner: a block deleted from a genome that is then empty)
            print("seed", seed, "not generated:", e, flush=True)
            return
        if kw.get("threads", 1) < 2:
            kw["threads"] = 3
        rp, qs = F.write(str(base / "in"), ref, gs, contigs, seed)
        a = F.run(F.REFBIN, rp, qs, str(base / "ref"), kw)
        for again in range(6):      # (the REFERENCE binary dies of SIGSEGV on some inputs in some runs -- seed 14009 with 3 threads: 6 runs of 8 -- and gives the same bytes whenever it survives)
            if a[0] is None or a[0] >= 0:
                break
            print("seed", seed, big, "the reference binary ended with signal", -a[0], "-- run again", flush=True)
            shutil.rmtree(base / "ref", ignore_errors=True)
            a = F.run(F.REFBIN, rp, qs, str(base / "ref"), kw)
        os.environ["PARSNP_TIMING"] = str(base / "timing.json")      # which route the run took (the summary line at the end)
        b = F.run(core, rp, qs, str(base / "mine"), kw)
        os.environ.pop("PARSNP_TIMING", None)
        try:
            import json
            tj = json.load(open(base / "timing.json"))
            routes[(tj.get("resident"), tj.get("resident_why") or "")] = routes.get((tj.get("resident"), tj.get("resident_why") or ""), 0) + 1
            routes["outside_writes"] = routes.get("outside_writes", 0) + int(tj.get("outside_writes", 0))
            if tj.get("outside_writes", 0) and tj.get("resident"):
                routes["resident runs with outside writes"] = routes.get("resident runs with outside writes", 0) + 1
                print("seed", seed, big, "stayed on the route with", tj.get("outside_writes"), "accepted member(s) outside their region", flush=True)
            if "accepted outside its region" in (tj.get("resident_why") or ""):
                print("seed", seed, big, "left the route:", tj.get("resident_why"), flush=True)
        except Exception:   # noqa: BLE001
            pass
        if a != b:
            bad.append((seed, big, kw))
            print("MISMATCH", seed, big, kw, flush=True)
    finally:
        shutil.rmtree(base, ignore_errors=True)


for seed in range(first, last):
    one(seed, False)
print("small sets done:", last - first, "mismatches:", len(bad), flush=True)
for seed in range(n_big):
    one(seed, True)
print("all done:", last - first + n_big, "runs, mismatches:", bad, flush=True)
print("routes:", routes, flush=True)
sys.exit(1 if bad else 0)
