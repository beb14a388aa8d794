"""The full-size flows of BASELINE configs 4 and 5, parameterised by size: tests/test_gpu_big.py runs them at BASELINE size on
the MI355X (product binary), tests/test_flows_small.py at a reduced size on the CPU checker -- the same assertions, so the
test logic itself is exercised before it costs GPU time.  Progress lines go to stdout and, where gpurun_out/ exists, to
gpurun_out/flows_progress.log as they happen (a run that is cut off still says how far it came)."""
import os
import shutil
import subprocess
import sys
import time

import xmfa_util
from parsnp_amd import driver, partition_run, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_T0 = time.time()


def say(msg):
    line = "[flows %7.1f s] %s" % (time.time() - _T0, msg)
    print(line, flush=True)
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "flows_progress.log"), "a") as f:
            f.write(line + "\n")


def config4_flow(core_bin, d, population_kw, part_size, n_parts, threads, golden=None, min_lcbs=20, extra_env=None):
    """n_parts x part_size genomes of the population model in the reference driver's order (sorted, Random(42) shuffle,
    parsnp:1509-1510), cut into partitions as parsnp:1553-1564 does, every partition through `core_bin` one after the other,
    then the native merge (include/parsnp_merge.h).  Checked: partition 0 against `golden` (whole-XMFA md5 + log counters of
    the REFERENCE binary) when given; every partition's XMFA self-consistent (part_size + 1 rows per block, MUM columns, every
    record spells its genome interval); every trimmed partition holds the same reference intervals; the merged parsnp.xmfa
    holds all sequences in every block, every record spells its genome interval, its reference bases are the
    intersection's."""
    n_total = part_size * n_parts
    kw = dict(population_kw, n_genomes=n_total)
    say("config 4: generating %d genomes x %.1f Mb" % (n_total, kw["n"] / 1e6))
    ref, gs = synth.population(**kw)
    rp, qs = synth.write_set(os.path.join(d, "in"), ref, gs)
    ref_len = len(ref)
    del gs
    say("config 4: %d partitions of %d through %s" % (n_parts, part_size, os.path.basename(core_bin)))
    t1 = time.time()
    if extra_env:
        os.environ.update(extra_env)
    res = partition_run.run_partitioned(core_bin, rp, driver.driver_order(qs), os.path.join(d, "out"), part_size, keep_trimmed=True, threads=threads)
    t2 = time.time()
    parts = res["partitions"]
    assert len(parts) == n_parts and all(p["ok"] and p["queries"] == part_size for p in parts), [(p["index"], p["rc"]) for p in parts]
    m = res["merged"]
    say("config 4: partitions + merge %.1f s; %d clusters, %d sequences, %d reference bases" % (t2 - t1, m["clusters"], m["sequences"], m["ref_bases"]))
    # where the merged file could depend on the insertion aligner (the reference: spoa.poa, partition.py:386): runs of
    # insertion columns that collected bases from more than one sequence, and those whose sequences are not all one string
    ins = m["insertions"]
    say("config 4: insertion runs in the merge: %d, shared by several sequences: %d (%d merged columns), of those with differing sequences: %d"
        % (ins["runs"], ins["shared"], ins["shared_columns"], ins["shared_diverse"]))
    res["insertion_exposure"] = ins
    if golden:      # the driver's first chunk = the golden's partition 0
        x0 = os.path.join(parts[0]["dir"], "parsnpAligner.xmfa")
        assert xmfa_util.log_counters(os.path.join(parts[0]["dir"], "parsnpAligner.log")) == golden["log"]
        assert xmfa_util.md5(x0) == golden["xmfa_md5"]
        say("config 4: partition 0 = the reference binary's golden (md5 + log counters)")
    pieces, tr = None, None
    gdir = os.path.join(d, "in")
    for p in parts:
        x = os.path.join(p["dir"], "parsnpAligner.xmfa")
        st = xmfa_util.native_consistency(x, gdir)
        assert st["bad_length"] == 0 and st["bad_mum_column"] == 0 and st["bad_sequence"] == 0 and st["missing_genomes"] == 0, (p["index"], st)
        assert st["min_rows"] == part_size + 1 and st["max_rows"] == part_size + 1 and st["lcbs"] > min_lcbs, st
        iv = x + ".trimmed.iv"
        tr = xmfa_util.native_consistency(x + ".trimmed", gdir, intervals=iv)
        assert tr["bad_length"] == 0 and tr["bad_sequence"] == 0 and tr["shifted"] == 0 and tr["missing_genomes"] == 0, (p["index"], tr)
        mine = open(iv).read()
        assert pieces is None or mine == pieces, "partition %d: trimmed reference intervals differ" % p["index"]
        pieces = mine
        os.remove(x + ".trimmed")
        say("config 4: partition %d checked (%d LCBs, %d trimmed pieces)" % (p["index"], st["lcbs"], tr["lcbs"]))
    assert m["sequences"] == n_total + 1 and m["clusters"] == len(pieces.splitlines()) > min_lcbs
    ms = xmfa_util.native_consistency(m["xmfa"], gdir, merged=True)
    assert ms["lcbs"] == m["clusters"] and ms["min_rows"] == n_total + 1 and ms["max_rows"] == n_total + 1 and ms["sequences"] == n_total + 1, ms
    assert ms["bad_length"] == 0 and ms["bad_mum_column"] == 0 and ms["bad_sequence"] == 0 and ms["shifted"] == 0 and ms["missing_genomes"] == 0, ms
    assert ms["ref_bases"] == m["ref_bases"] == tr["ref_bases"]
    assert m["ref_bases"] > 0.8 * ref_len
    say("config 4: merged XMFA checked, %.2f GB; total %.1f s" % (os.path.getsize(m["xmfa"]) / 1e9, time.time() - t1))
    return res


def config5_flow(core_bin, d, workload, override, threads, ranks, min_lcbs=20, min_reverse=10, sharded_env=None, golden=None):
    """`workload` (population model + rearrangements) --no-partition through `core_bin`: XMFA self-consistency (one row per
    genome in every block, MUM columns, every record spells its genome interval, reverse-strand records present),
    run-to-run determinism, and the same bytes from the sharded form of the run -- `ranks` ranks, each with its block of the
    query genomes resident (parsnp_amd.sharded: on a box with fewer GPUs than ranks they share the GPU and exchange over gloo)."""
    say("config 5: generating %s %s" % (workload, override))
    ref, gs = synth.make(workload, **override)
    rp, qs = synth.write_set(os.path.join(d, "in"), ref, gs)
    n = len(gs) + 1
    del gs
    sums, walls = [], []
    for rep in range(2):
        out = os.path.join(d, "out%d" % rep)
        t = time.time()
        timing = os.path.join(d, "timing%d.json" % rep)
        rc, _ = driver.run_core(core_bin, rp, qs, out, threads=threads, env=dict(os.environ, OMP_WAIT_POLICY="passive"), timing=timing)
        walls.append(time.time() - t)
        assert rc == 0, open(os.path.join(out, "parsnp-aligner.err")).read()[-2000:]
        sums.append(xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")))
        if golden and rep == 0:
            # the REFERENCE binary's bytes at this size (tests/golden/e2e_big.json: oracle/_ref/parsnp_core_ref, 67 minutes in the build
            # container), from the device-resident route
            import json
            tj = json.load(open(timing))
            assert tj["resident"] == 1 and tj["resident_retry"] == 0, tj
            assert len(qs) == golden["n_queries"]
            assert xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")) == golden["log"]
            assert sums[0] == golden["xmfa_md5"], "the XMFA differs from the reference binary's"
            say("config 5: the reference binary's bytes (md5 %s), resident route" % sums[0])
        assert "NOTE" not in open(os.path.join(out, "parsnpAligner.log")).read()
        say("config 5: run %d, whole process %.1f s" % (rep, walls[-1]))
    assert sums[0] == sums[1]
    st = xmfa_util.native_consistency(os.path.join(d, "out0", "parsnpAligner.xmfa"), os.path.join(d, "in"))
    assert st["lcbs"] > min_lcbs and st["min_rows"] == n and st["max_rows"] == n, st
    assert st["bad_length"] == 0 and st["bad_mum_column"] == 0 and st["bad_sequence"] == 0 and st["missing_genomes"] == 0, st
    assert st["reverse"] > min_reverse, st
    say("config 5: XMFA checked: %s" % st)
    shutil.rmtree(os.path.join(d, "out1"), ignore_errors=True)
    out = os.path.join(d, "sharded")
    os.makedirs(out)
    ini = os.path.join(out, "run.ini")
    open(ini, "w").write(driver.ini_text(rp, qs, out, threads=max(2, threads // ranks)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % ranks, "--master-addr", "127.0.0.1", "--master-port", "29591",
           "-m", "parsnp_amd.sharded", ini]
    t = time.time()
    p = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT, OMP_WAIT_POLICY="passive", **(sharded_env or {})),
                       cwd=out, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")) == sums[0]
    assert xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")) == xmfa_util.log_counters(os.path.join(d, "out0", "parsnpAligner.log"))
    say("config 5: sharded x%d: same bytes, %.1f s" % (ranks, time.time() - t))
    return st
