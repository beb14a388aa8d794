"""The N>1 path (partition mode: one partition per rank, results all-gathered) with 2 processes on the gloo backend.
On CPU the partitions run the test build of the host (CPU checker behind the C ABI); on the GPU box the same code
drives parsnp_amd/bin/parsnp_core with one GPU per rank."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

WORKER = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from parsnp_amd import partition_run
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
core, ref, outdir, listfile = sys.argv[2:6]
files = open(listfile).read().split()
res = partition_run.run_partitioned(core, ref, files, outdir, 3, rank, world, dist)
json.dump(res, open(os.path.join(outdir, "result_rank%d.json" % rank), "w"))
dist.barrier()
dist.destroy_process_group()
'''


def test_partition_plan_matches_driver_arithmetic():
    from parsnp_amd import partition_run
    files = ["g%03d" % i for i in range(2000)]
    chunks = partition_run.plan_partitions(files, 250)
    assert [len(c) for c in chunks] == [250] * 8 and sum(chunks, []) == files
    chunks = partition_run.plan_partitions(files[:1100], 250)     # 4 full partitions of 275
    assert [len(c) for c in chunks] == [275] * 4
    chunks = partition_run.plan_partitions(files[:13], 3)         # size 13 // 4 = 3 -> 5 chunks, the last one short
    assert [len(c) for c in chunks] == [3, 3, 3, 3, 1]


def test_interval_intersection():
    from parsnp_amd import partition_run
    a = [(1, 100), (200, 300)]; b = [(50, 250)]; c = [(60, 70), (90, 220), (290, 400)]
    assert partition_run.intersect([a, b]) == [(50, 100), (200, 250)]
    assert partition_run.intersect([a, b, c]) == [(60, 70), (90, 100), (200, 220)]
    assert partition_run.intersect([a]) == a and partition_run.intersect([]) == []


def test_two_ranks_gloo(cpu_checkers, tmp_path):
    from parsnp_amd import driver, partition_run, synth
    import xmfa_util
    ref, gs = synth.population(seed=31, n=40000, n_genomes=12, div=0.02, indel_frac=0.05)
    rp, qs = synth.write_set(str(tmp_path / "in"), ref, gs)
    files = driver.driver_order(qs)
    listfile = tmp_path / "files.txt"
    listfile.write_text("\n".join(files))
    worker = tmp_path / "worker.py"
    worker.write_text(WORKER)
    out2 = str(tmp_path / "two")
    os.makedirs(out2)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29561", str(worker), ROOT, cpu_checkers, rp, out2, str(listfile)]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    r0 = json.load(open(os.path.join(out2, "result_rank0.json")))
    r1 = json.load(open(os.path.join(out2, "result_rank1.json")))
    assert r0 == r1                                          # every rank holds the merged view
    assert [p["index"] for p in r0["partitions"]] == [0, 1, 2, 3] and all(p["ok"] for p in r0["partitions"])
    # single-process run of the same plan: identical partitions, identical intersection
    out1 = str(tmp_path / "one")
    single = partition_run.run_partitioned(cpu_checkers, rp, files, out1, 3)
    assert [p["intervals"] for p in single["partitions"]] == [[list(i) for i in p["intervals"]] for p in r0["partitions"]] or \
           [[list(i) for i in p["intervals"]] for p in single["partitions"]] == [p["intervals"] for p in r0["partitions"]]
    assert [list(i) for i in single["intersection"]] == r0["intersection"]
    assert len(r0["intersection"]) >= 1
    # rank 0 merged the four partitions into one XMFA: every genome once, the same clusters as the single-process merge
    assert r0["merged"]["sequences"] == 13 and r0["merged"]["clusters"] == single["merged"]["clusters"] >= 1
    assert xmfa_util.md5(r0["merged"]["xmfa"]) == xmfa_util.md5(single["merged"]["xmfa"])
    for a, b in zip(single["partitions"], r0["partitions"]):
        assert xmfa_util.md5(os.path.join(a["dir"], "parsnpAligner.xmfa")) == xmfa_util.md5(os.path.join(b["dir"], "parsnpAligner.xmfa"))


EXCHANGE = r'''
import sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from bench import exchange_intervals
dist.init_process_group("gloo")
r = dist.get_rank()
iv = [[1, 100], [200, 300]] if r == 0 else [[50, 250], [260, 270], [400, 500]]
got = exchange_intervals(torch, dist, "cpu", iv)
assert got == 113, got
assert exchange_intervals(torch, dist, "cpu", [] if r == 0 else iv) == 0
dist.barrier()
dist.destroy_process_group()
'''


def test_bench_interval_exchange_two_ranks(tmp_path):
    """bench.py's N>1 exchange step (all-gather of ragged LCB interval lists + intersection) on gloo"""
    w = tmp_path / "ex.py"
    w.write_text(EXCHANGE)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29563", str(w), ROOT]
    p = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]


@pytest.mark.parametrize("name,world,route", [("pop6x200k", 2, "host"), ("pop6x200k", 2, "resident"), ("pop12x400k", 3, "resident"), ("poprearr10x400k", 3, "host"),
                                              ("poprearr10x400k", 2, "resident_all_lists"), ("mumi", 2, "host")])
def test_sharded_run_gloo(emu, tmp_path, name, world, route):
    """query genomes sharded over ranks, Master.EP all-reduced (min), candidate columns all-gathered.
    The engine here is the host-emulated kernel code (tests/emu) and the collectives run on gloo; the result must be
    the single-process result (= the reference binary's golden).  route "resident": thresholds lowered so that the small sets
    take the resident route (every rank validates the same candidate rows on its own device: no host work to replicate and
    no further exchange); "resident_all_lists" sends the rearranged set's anchor list through it as well: clusters of waiting regions
    that meet in some genome wait for one another on every rank alike (round 6; until then the ranks left the route together)."""
    import json
    import test_host_logic as H
    import xmfa_util
    from parsnp_amd import driver
    core_lib = os.path.join(ROOT, "tests", "emu", "libparsnp_core_emu.so")
    mumi = name == "mumi"
    rp, qs, kw = H.mumi_inputs("pop6x200k_p", str(tmp_path)) if mumi else H.harsh_inputs(name, str(tmp_path))
    out = str(tmp_path / "out")
    os.makedirs(out)
    ini = os.path.join(out, "run.ini")
    open(ini, "w").write(driver.ini_text(rp, qs, out, calcmumi=1 if mumi else 0, **kw))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PARSNP_CORE_LIB=core_lib, PYTHONPATH=ROOT)
    log = str(tmp_path / "route.log")
    if route != "host":
        env.update(PM_DIRTY_MIN="8", PARSNP_PARALLEL_MIN="8", PARSNP_RESIDENT_LOG=log, PM_FLAGGED_DIV="1" if route == "resident_all_lists" else "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
           "--master-port", "29571", "-m", "parsnp_amd.sharded", ini]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=out, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    if mumi:
        lines = sorted(open(os.path.join(out, "all.mumi")).read().split(), key=lambda x: int(x.split(":")[0]))
        assert lines == H.MUMI["pop6x200k_p"]
    else:
        assert xmfa_util.mum_lcb_signature(os.path.join(out, "parsnpAligner.xmfa")) == H.E2E[name]["signature"]
        assert xmfa_util.md5(os.path.join(out, "parsnpAligner.xmfa")) == H.E2E[name]["xmfa_md5"]
        assert xmfa_util.log_counters(os.path.join(out, "parsnpAligner.log")) == H.E2E[name]["log"]
        if route != "host":      # every rank took (or left) the route
            lines = open(log).read().split("\n")[:-1]
            assert len(lines) == world and all("resident=1" in ln and "retry=0" in ln for ln in lines), lines
