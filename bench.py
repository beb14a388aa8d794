#!/usr/bin/env python3
"""bench.py -- genomes/sec of the MUM + LCB hot path (phases A-D of parsnp_core: anchor multi-MUM search,
recursive inter-anchor extension, LCB formation) on MI355X, on BASELINE.json's headline configuration.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload bact200] [--genomes G]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic genomes (reference + G queries) whose packed
sequences are already resident in HBM (ingest, upload and XMFA writing are outside the timed region and reported
separately).  N > 1: partition mode's natural split -- every rank owns one partition (the shared reference + its own
G query genomes) on its own GPU, no data-path collective; value = genomes of all ranks / max-over-ranks time.  THE N > 1
HEADLINE IS THEREFORE N x G GENOMES IN N ALIGNMENTS ("scaling": "weak", `total_genomes` = N x G) -- a larger job than the
metric's one 200-genome alignment; that one job on N GPUs is the line's `sharded_strong`: the SAME G genomes as one
alignment sharded over the N GPUs (strong scaling: all-reduce(min) + all-gather per engine call over the engine's RCCL
communicator), measured by a child process per rank; DESIGN.md section 5 holds the predicted curve of both.  `n_ranks_seen_by_rccl` and `per_rank` (host threads budgeted from the container's CPU quota,
cores kept busy, ms per step of every rank) say what the numbers were measured on.

The JSON line also carries
  roofline      dominant kernel (SeedExtend), timed live with HIP events on the engine's stream over every engine launch of
                the timed steps: achieved = algorithmic bytes per launch (m/4 + 16 m + 16 n per query genome and
                region) / average launch duration; the anchor launch alone is reported beside it
  cpu_baseline  the REFERENCE binary (oracle/_ref/parsnp_core_ref, built from /root/reference in the build container)
                on the host cores of this box, on a bounded sample of the same workload, single thread (the reference's
                MUM+LCB path is single-threaded, src/parsnp.cpp:1600-1619).
"""
import argparse
import gc
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
RANDOM_PEAK_GBS = 3400.0   # measured ceiling of scattered 64-byte requests (scripts/hbm_calib.hip gather kernels, profiles/r01/calibration.json: 54 G requests/s): what an index probe can reach
PROFILE_ROUND = "r06"   # profiles/<round>/traffic_seed_extend.json, calibration.json: the PMC passes of the shipped binary


def engine_src_sha256():
    """sha256 over the sources of the engine translation unit (scripts/profile_summary.py stamps the same)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("kernels.h", "store_kernels.h", "engine_core.h", "engine_hip.hip", "abi_glue.h"):
        p = os.path.join(ROOT, "parsnp_amd", "csrc", "engine", f)
        if not os.path.exists(p):
            return None
        h.update(open(p, "rb").read())
    return h.hexdigest()


def so_sha256():
    """sha256 of the HIP library this process runs: the PMC traffic figure is only quoted for the binary it was measured on"""
    import hashlib
    from parsnp_amd.paths import HIP_LIB
    return hashlib.sha256(open(HIP_LIB, "rb").read()).hexdigest()


def anchor_alg(G, m_avg, n_ref, stride, phases):
    """byte model of the anchor launch's event search: G query genomes streamed once, the reference window once, a 64-B
    index request per leader (8 B per sample), 64 B per sample SeedRest probes, 16 B per event"""
    samples = (m_avg - 16) // max(1, stride) + 1
    return G * (m_avg / 2 + 8 * samples) + n_ref / 2 + 64 * phases.get("rest_samples", 0.0) + 16 * phases.get("events", 0.0)


def issue_fraction(pdir):
    """share of the anchor dispatch's cycles in which SeedExtend's SIMDs issue vector instructions, from the SQ counters of THIS
    build (scripts/sqcounters.sh) and the measured int32 issue rate of a gfx950 SIMD (scripts/valu_calib.hip); None without both"""
    try:
        sq = json.load(open(os.path.join(pdir, "sq_seed_extend.json")))
        cal = json.load(open(os.path.join(pdir, "calibration.json")))
    except (OSError, ValueError):
        return None
    if sq.get("so_sha256") != so_sha256():      # the counters of another build say nothing about this one
        return None
    ipc = (cal.get("valu") or {}).get("valu_int32_wave64_instructions_per_cycle_per_simd")
    k = sq.get("kernels", {}).get("SeedExtend", {})
    c = k.get("counters", {})
    g = lambda n: c.get(n, {}).get("max_per_record", 0.0)      # noqa: E731 -- the records of the longest dispatch (the anchor launch)
    n_disp = max(1, k.get("dispatch_ns", {}).get("n", 1))
    per_dispatch = max(1, c.get("SQ_INSTS_VALU", {}).get("records", n_disp) // n_disp)      # counter records per dispatch (one per XCD and sampled shader engine)
    if not ipc or not g("SQ_INSTS_VALU") or not g("GRBM_GUI_ACTIVE"):
        return None
    simds = 256 * 4
    valu_total = g("SQ_INSTS_VALU") * per_dispatch
    cycles = g("GRBM_GUI_ACTIVE")
    return {"valu_instructions": int(valu_total), "dispatch_cycles": int(cycles), "simds": simds, "cycles_per_valu": round(1.0 / ipc, 3),
            "valu_per_wave": int(g("SQ_INSTS_VALU") / max(1.0, g("SQ_WAVES"))),
            "vmem_rd_per_wave": int(round(g("SQ_INSTS_VMEM_RD") / max(1.0, g("SQ_WAVES")))), "vmem_wr_per_wave": int(round(g("SQ_INSTS_VMEM_WR") / max(1.0, g("SQ_WAVES")))),
            "issue_frac": round(valu_total / simds / ipc / cycles, 4),
            "wait_frac": round(g("SQ_WAIT_ANY") / max(1.0, g("SQ_WAVE_CYCLES")), 4),
            "source": "profiles/%s/sq_seed_extend.json + calibration.json" % PROFILE_ROUND}


def make_inputs(workdir, workload, genomes, rank):
    from parsnp_amd import synth
    model, kw = synth.CONFIGS[workload]
    kw = dict(kw)
    if genomes:
        kw["n_genomes"] = genomes
    if model == "population":
        kw["carry_seed"] = None if rank == 0 else 1000 + rank   # same reference + site pool, a different partition per rank
    ref, gs = getattr(synth, model)(**kw)
    rp, qs = synth.write_set(os.path.join(workdir, "in"), ref, gs)
    return rp, qs, len(ref), sum(len(g) for g in gs) / max(1, len(gs)), kw


def cpu_baseline(workdir, rp, qs, sample_queries):
    """the reference binary on a bounded sample (ref + the first `sample_queries` genomes), 1 thread; where the prebuilt
    reference binary did not travel, the CPU restatement of the path (oracle provider behind the same host code)"""
    from parsnp_amd import driver
    refbin = os.path.join(ROOT, "oracle", "_ref", "parsnp_core_ref")
    kind, what = "reference", "reference parsnp_core (oracle/_ref)"
    if not os.path.exists(refbin):
        refbin = os.path.join(ROOT, "oracle", "_ref", "parsnp_core_oracle")
        kind, what = "port", "CPU restatement (oracle/_ref/parsnp_core_oracle: this repo's host code over oracle/mum_oracle.c)"
        sample_queries = 1
        if not os.path.exists(refbin):
            return None
    out = os.path.join(workdir, "cpu_baseline")
    t0 = time.time()
    rc, _ = driver.run_core(refbin, rp, qs[:sample_queries], out, threads=1)
    wall = time.time() - t0
    if rc != 0:
        return None
    log = open(os.path.join(out, "parsnpAligner.log")).read()
    timers = [float(x) for x in re.findall(r"(?:anchor search|coarsening|filtering|Inter-clustering) elapsed time:\s+([0-9.]+)s", log)]
    path_s = sum(timers)
    if path_s <= 0:
        path_s = wall
    return {"value": round(sample_queries / path_s, 5), "unit": "genomes/s", "cores": 1, "kind": kind,
            "sample": "%s on ref + first %d query genomes of the workload; phases A-D by its own "
                      "1-s log timers = %.0f s, whole process %.1f s; host has %d cores" % (what, sample_queries, path_s, wall, os.cpu_count())}


def usable_cpus():
    """CPUs this process may use: affinity mask, capped by the cgroup CPU quota (cpu.max) where there is one"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def bind_to_numa_node(torch, dev, local_rank, local_world):
    """Keep this rank's threads and memory on ONE NUMA node (what `numactl --cpunodebind` would do): the host phases walk
    per-genome bitmaps that one thread zeroes and others read, and remote accesses cost 20 % of the step on the 2-socket
    GPU boxes.  Node = the GPU's own node where sysfs tells, else spread the ranks evenly.  -> node or None"""
    try:
        nodes = sorted(int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit())
        if len(nodes) < 2:
            return None
        def gpu_node(d):
            try:
                p = torch.cuda.get_device_properties(d)
                bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
                v = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
                return v if v in nodes else None
            except Exception:   # noqa: BLE001 -- older torch / no sysfs entry
                return None
        # the GPUs' own nodes, if sysfs knows them for every rank of this node and they spread the ranks evenly
        per_rank = [gpu_node(r % torch.cuda.device_count()) for r in range(max(1, local_world))]
        balanced = None not in per_rank and max(per_rank.count(x) for x in nodes) <= -(-max(1, local_world) // len(nodes))
        if balanced:
            node = per_rank[local_rank % len(per_rank)]
        else:
            node = nodes[(local_rank * len(nodes)) // max(1, local_world) % len(nodes)]
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
            return node
    except (OSError, ValueError):
        pass
    return None


def exchange_intervals(torch, dist, tdev, intervals):
    """all-gather the per-rank [start,end] lists (padded to the longest) and intersect them -> bases aligned in every partition"""
    from parsnp_amd.partition_run import intersect
    world = dist.get_world_size()
    mine = torch.tensor(intervals if intervals else [[0, -1]], dtype=torch.int64, device=tdev).reshape(-1, 2)
    count = torch.tensor([mine.shape[0]], dtype=torch.int64, device=tdev)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count)
    longest = int(max(int(c.item()) for c in counts))
    padded = torch.full((longest, 2), -1, dtype=torch.int64, device=tdev)
    padded[:, 0] = 0
    padded[:mine.shape[0]] = mine
    gathered = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded)
    lists = [[(int(a), int(b)) for a, b in g[:int(c.item())].cpu().tolist() if b >= a] for g, c in zip(gathered, counts)]
    return sum(b - a + 1 for a, b in intersect(lists))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)      # ~5 s of timed region at 200 x 5 Mb: long enough for an outside GPU-activity sampler to see
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="bact200")
    ap.add_argument("--genomes", type=int, default=0, help="override the number of query genomes per partition")
    ap.add_argument("--cpu-sample", type=int, default=2, help="query genomes in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--host-threads", type=int, default=0,
                    help="ini [LCB] cores: host threads for ingest, candidate validation, output (0 = 24, fewer when the CPUs this "
                         "container may use, shared by the ranks of the node, do not allow it)")
    ap.add_argument("--mode", default="both", choices=["both", "partition", "sharded"],
                    help="N > 1: 'partition' = one independent partition per GPU, weak scaling, no data-path collective (the headline "
                         "value); 'sharded' = ONE alignment (the workload's genomes) sharded over the GPUs, strong scaling, both exchanges "
                         "of every engine call over the engine's own RCCL communicator (device buffers); 'both' (default) = the partition "
                         "measurement as the line's value and, for N > 1, the sharded measurement of the same workload beside it "
                         "(`sharded_strong`, run by one child process per rank so that it cannot take the headline down with it)")
    ap.add_argument("--inputs", default="", help="directory with ref.fna + g*.fna to use instead of generating the workload (the sharded child of --mode both)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="a tunable of the engine session (pm_session_tune: group_small=0, ...) for before/after measurements; named in config.tune")
    ap.add_argument("--other-configs", default="auto", choices=["auto", "on", "off"],
                    help="ms per step of BASELINE's other single-GPU configurations (viral50 = config 2, rearr500 = config 5) and of config 3 with ten inverted segments (bact200inv) in the line's "
                         "`other_configs`, each measured by a child run of this script; auto = on for the default workload at N = 1")
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()

    # OpenMP workers between the short parallel host phases (~20 per step): the GPU boxes cap the CPU time of the container
    # (cgroup quota), and workers that spin without bound burn it (measured in round 1: "active" cost 50 % of the throughput);
    # workers that sleep at once cost a futex wake-up per phase.  A short bounded spin (20 000 iterations, ~10 us) before
    # sleeping is the measured best when one rank has the quota to itself (200 x 5 Mb, 24 threads: 25.2 ms per step against
    # 26.4 passive, 11 instead of 8 cores busy); ranks that share a node's quota sleep at once.
    if int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))) == 1 and usable_cpus() >= 12:
        os.environ.setdefault("OMP_WAIT_POLICY", "active")
        os.environ.setdefault("GOMP_SPINCOUNT", "20000")
    else:
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU path); run it through gpurun")
    dev = local_rank % torch.cuda.device_count()   # (more ranks than GPUs only happens in smoke tests of this script)
    torch.cuda.set_device(dev)
    # torch ships its own HIP runtime; the engine links the system one, whose "current device" torch cannot set:
    # tell the engine which GPU this rank owns
    os.environ["PARSNP_DEVICE"] = str(dev)
    dist = None
    tdev = "cuda"
    sharded = args.mode == "sharded"
    # (PARSNP_BENCH_CHILD_TEST=1: exercise the child-process plumbing of `both` with the one rank a single-GPU box allows)
    both = args.mode == "both" and (world > 1 or os.environ.get("PARSNP_BENCH_CHILD_TEST") == "1")
    if world > 1:
        import torch.distributed as dist
        if torch.cuda.device_count() >= world and not sharded:
            dist.init_process_group("nccl")   # RCCL, one rank per GPU
        else:
            # smoke test of this script on a box with fewer GPUs than ranks; and the sharded mode, where torch.distributed is
            # only the rendezvous that carries the engine's RCCL id (the process then holds ONE RCCL: the engine's)
            dist.init_process_group("gloo")
            tdev = "cpu"
    if sharded and world > torch.cuda.device_count():
        sys.exit("bench.py --mode sharded needs one GPU per rank (RCCL refuses two ranks on one device)")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    numa_node = None if os.environ.get("PARSNP_BENCH_NO_BIND") else bind_to_numa_node(
        torch, dev, local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    from parsnp_amd import driver
    from parsnp_amd.core_api import CoreRun
    if args.host_threads <= 0:
        # the parallel host phases are short bursts, so 1.5 threads per usable CPU still helps; all ranks share the node
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        args.host_threads = max(4, min(24, int(1.5 * usable_cpus() / max(1, local_world))))

    # scratch for the synthetic FASTA (~1 GB per rank at 200 x 5 Mb) and, on rank 0, the XMFA (~1 GB): RAM disk when it has
    # room for every rank of this node, else the default temp dir
    scratch = None
    if os.path.isdir("/dev/shm"):
        try:
            if shutil.disk_usage("/dev/shm").free > (3 << 30) * world:
                scratch = "/dev/shm"
        except OSError:
            pass
    workdir = tempfile.mkdtemp(prefix="parsnp_bench_r%d_" % rank, dir=scratch)
    try:
        t0 = time.time()
        if args.inputs:      # files some other process generated (same workload, same seeds)
            import glob
            rp, qs = os.path.join(args.inputs, "ref.fna"), sorted(glob.glob(os.path.join(args.inputs, "g*.fna")))
            size = lambda f: sum(len(l) - 1 for l in open(f, "rb") if not l.startswith(b">"))      # noqa: E731
            n_ref, m_avg = size(rp), sum(size(q) for q in qs[:4]) / max(1, len(qs[:4]))
            kw = dict(__import__("parsnp_amd.synth", fromlist=["CONFIGS"]).CONFIGS[args.workload][1])
        else:
            rp, qs, n_ref, m_avg, kw = make_inputs(workdir, args.workload, args.genomes, 0 if sharded else rank)   # sharded: every rank the same genomes
        gen_s = time.time() - t0
        out = os.path.join(workdir, "out")
        os.makedirs(out, exist_ok=True)
        ini = os.path.join(out, "parsnpAligner.ini")
        open(ini, "w").write(driver.ini_text(rp, qs, out, threads=args.host_threads))
        # quiet: the reference's progress chatter goes to a file
        so, se = os.dup(1), os.dup(2)
        logf = os.open(os.path.join(out, "bench.log"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
        os.dup2(logf, 1); os.dup2(logf, 2)
        try:
            if sharded:
                from parsnp_amd.sharded import ShardedRun

                class _Solo:      # the one-rank case needs no process group
                    @staticmethod
                    def get_rank(): return 0
                    @staticmethod
                    def get_world_size(): return 1
                    @staticmethod
                    def broadcast_object_list(box, src=0): return None
                run = ShardedRun(ini, dist if dist is not None else _Solo, "cpu", None, rccl=True)
            else:
                run = CoreRun(ini)
            for kv in args.tune:
                key, _, val = kv.partition("=")
                run.L.pc_tune.argtypes = [__import__("ctypes").c_void_p, __import__("ctypes").c_char_p, __import__("ctypes").c_longlong]
                if run.L.pc_tune(run.h, key.encode(), int(val)) != 0:
                    raise SystemExit("bench.py: --tune %s refused by the engine" % kv)
            cold_step_s = None
            # the harness's own interpreter must not stall the passes it times: a full cyclic collection with torch imported
            # costs ~40 ms and would otherwise fire inside the first (cold) pass or one of the few timed steps
            gc.collect()
            gc.disable()
            for w in range(args.warmup):
                tc = time.perf_counter()
                run.step()
                if w == 0:
                    cold_step_s = time.perf_counter() - tc        # the first pass: fresh arenas, first device allocations
            # the interpreter's cyclic collector would otherwise fire inside one of the few timed steps (a full collection
            # with torch imported costs ~40 ms): collect now, keep it off while timing
            gc.collect()
            gc.disable()
            barrier()
            t0 = time.perf_counter()
            cpu0 = sum(os.times()[:2])
            reports = []
            merged_bp = None
            step_ms = []
            for _ in range(args.steps):
                ts = time.perf_counter()
                rep = run.step(intervals=dist is not None and not sharded)      # (the LCBs' reference intervals only where the exchange step reads them)
                step_ms.append(round(1e3 * (time.perf_counter() - ts), 2))
                if dist is not None and not sharded:
                    # partition mode's exchange step: every rank's LCB reference intervals are all-gathered (RCCL) and
                    # intersected -- the positions aligned in EVERY partition (partition.py:35-61, 539-583)
                    merged_bp = exchange_intervals(torch, dist, tdev, rep["lcb_ref_intervals"])
                rep.pop("lcb_ref_intervals", None)
                reports.append(rep)
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            host_cores_busy = (sum(os.times()[:2]) - cpu0) / elapsed      # CPU seconds of this process (all threads) per second of the timed region
            gc.enable()
            barrier()
            # the timed region ends with the MUM list, the LCBs and their counters on the host; the ROWS of the MUMs stay on the device
            # until the writer asks (resident route).  What fetching them costs is measured here, once, and reported beside ms_per_step
            materialize_ms = None
            if rank == 0 and not sharded:
                run.L.pc_materialize.argtypes = [__import__("ctypes").c_void_p]
                run.L.pc_materialize.restype = __import__("ctypes").c_double
                materialize_ms = round(float(run.L.pc_materialize(run.h)), 3)
            t_out = time.time()
            if rank == 0:
                run.write()          # XMFA + log of one partition: outside the timed region, reported as split_s.output
            output_s = time.time() - t_out
            # how many ranks RCCL itself counts: the engine's communicator (sharded), torch's (partition mode's interval exchange)
            if sharded:
                run.L.pc_rccl_ranks.argtypes = [__import__("ctypes").c_void_p]
                rccl_ranks = int(run.L.pc_rccl_ranks(run.h))
            else:
                rccl_ranks = dist.get_world_size() if (dist is not None and dist.get_backend() == "nccl") else (None if dist is None else 0)      # None: no communicator exists (N = 1)
            run.close()
        finally:
            os.dup2(so, 1); os.dup2(se, 2)
            if os.environ.get("PARSNP_BENCH_LOG") and rank == 0:     # keep the host's chatter (PARSNP_DEBUG_TIMERS laps)
                shutil.copyfile(os.path.join(out, "bench.log"), os.environ["PARSNP_BENCH_LOG"])
        per_rank = [{"rank": rank, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "host_threads": args.host_threads, "host_cores_busy": round(host_cores_busy, 2)}]
        sharded_strong = None
        if dist is not None:
            box = [None] * world
            dist.all_gather_object(box, per_rank[0])
            per_rank = box
        if both:
            # the same workload as ONE alignment sharded over the ranks (strong scaling): a child process per rank -- its own HIP
            # context and the engine's own RCCL, rendezvous one port up -- on rank 0's genome files; a failure or a hang of that
            # never-before-exercised path costs this key, not the line
            box = [os.path.join(workdir, "in") if rank == 0 else None]
            if dist is not None:
                dist.broadcast_object_list(box, src=0)
            if torch.cuda.device_count() >= world:
                env = dict(os.environ, MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29500")) + 23), PARSNP_RCCL_TIMEOUT="60")
                cmd = [sys.executable, os.path.abspath(__file__), "--mode", "sharded", "--gpus", str(world), "--steps", str(args.steps), "--warmup", str(args.warmup),
                       "--workload", args.workload, "--cpu-sample", "0", "--inputs", box[0], "--host-threads", str(args.host_threads)]
                try:
                    pr = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)      # (a healthy child: ~40 s; the line must not wait 15 minutes for a hung one)
                    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
                    if rank == 0:
                        if pr.returncode == 0 and lines:
                            cj = json.loads(lines[-1])
                            sharded_strong = {k: cj.get(k) for k in ("value", "unit", "ms_per_step", "scaling", "n_gpus", "steps", "n_ranks_seen_by_rccl", "per_rank", "step_ms",
                                                                     "engine_ms", "split_s", "mums", "lcbs", "core_bp_aligned")}
                            sharded_strong["config"] = cj.get("config", {}).get("parallelism")
                        else:
                            sharded_strong = {"error": "child exit code %d: %s" % (pr.returncode, pr.stderr[-400:])}
                except subprocess.TimeoutExpired:
                    sharded_strong = {"error": "the sharded child did not finish within 300 s"}
            else:
                sharded_strong = {"skipped": "needs one GPU per rank (%d ranks, %d GPUs visible)" % (world, torch.cuda.device_count())}
            if dist is not None:
                dist.barrier()
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            cb = torch.tensor([reports[-1]["core_bp"]], dtype=torch.float64, device=tdev)
            dist.all_reduce(cb, op=dist.ReduceOp.SUM)
            core_bp_total = int(cb.item()) if not sharded else reports[-1]["core_bp"]      # sharded: every rank holds the one alignment
        else:
            core_bp_total = reports[-1]["core_bp"]

        if rank == 0:
            rep = reports[-1]
            G = len(qs)
            value = (1 if sharded else world) * G * args.steps / elapsed       # sharded: the G genomes are the whole job
            # dominant kernel: per-phase HIP-event times of every engine launch of the timed steps (anchor launch +
            # recursion launches), summed per step and averaged over the steps
            phases, totals = {}, {}
            for r in reports:
                for k, v in r["anchor_ms"].items():
                    phases[k] = phases.get(k, 0.0) + v / len(reports)
                for k, v in r["engine_ms"].items():
                    totals[k] = totals.get(k, 0.0) + v / len(reports)
            # (the event search of a recursion batch's small regions has its own phase mark, `grouped_events`: one phase with the rest of the search)
            for tab in (phases, totals):
                if "grouped_events" in tab:
                    tab["seed_extend"] = tab.get("seed_extend", 0.0) + tab.pop("grouped_events")
            counts = ("budget_retries", "events", "rest_samples", "n_positions", "n_candidates", "n_accepted", "n_grouped", "exact_cluster_tests", "deferred_regions", "tail_repeats", "outside_writes")          # counts that travel in the timing list, not times
            kernels = {k: v for k, v in totals.items() if k not in ("setup", "download", "units", "call_wall") + counts}
            dom = max(kernels, key=kernels.get) if kernels else None
            launches = sum(r["finder_calls"] for r in reports) / len(reports)          # engine launches per step
            survey_step = sum(r["alg_bytes"] for r in reports) / len(reports)          # SURVEY 8d bytes of all of them
            # what THIS engine's event search must move per step: (m + n)/2 per (region, query genome) -- the query piece and the
            # reference window once, 16 B per 32 bases -- + one 64-B index request per sampled K-mer (pairs that fit 128 bases use
            # no index) + 16 B per event it appends; summed by the host over every pair it sends (Stats::alg_bytes_kernel)
            events_step = totals.get("events", 0.0)
            rest_step = totals.get("rest_samples", 0.0)
            alg_step = sum(r.get("alg_bytes_kernel", 0) for r in reports) / len(reports) + 64.0 * rest_step + 16.0 * events_step
            b_alg = m_avg / 4 + 16 * m_avg + 16 * n_ref          # bytes per query genome of the anchor launch
            anchor_stride = max(1, int(reports[-1].get("anchor_minsize", 25)) - 16 + 1)     # sampling step of the anchor launch (K = 16)
            roof = None
            traffic = None
            traffic_raw = None
            traffic_note = "no PMC pass on file for this binary"
            pdir = os.path.join(ROOT, "profiles", PROFILE_ROUND)
            tpath = os.path.join(pdir, "traffic_seed_extend.json")
            if dom == "seed_extend" and args.workload == "bact200" and G == 200 and os.path.exists(tpath):
                # fabric-side bytes per launch of this kernel from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) on this exact
                # workload; see the file for provenance, the calibration and the correction applied.  Quoted only for the
                # binary the passes ran on (sha256 of libparsnp_hip.so stamped into the file by scripts/profile_summary.py).
                tj = json.load(open(tpath))
                if tj.get("so_sha256") == so_sha256() or (tj.get("engine_src_sha256") and tj.get("engine_src_sha256") == engine_src_sha256()):
                    traffic = tj.get("hbm_bytes_per_launch")
                    if tj.get("hbm_bytes_per_step") and launches:      # (per step on file: divided by THIS run's engine calls per step)
                        traffic = tj["hbm_bytes_per_step"] / launches
                    traffic_raw = tj.get("raw")
                    traffic_note = tj.get("note", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, this binary (profiles/%s)" % PROFILE_ROUND)
                else:
                    traffic_note = "profiles/%s/traffic_seed_extend.json was measured on another build of the engine: not quoted" % PROFILE_ROUND
            if dom and launches:
                launch_ms = kernels[dom] / launches
                alg_gbs = alg_step / (kernels[dom] * 1e-3) / 1e9
                survey_gbs = survey_step / (kernels[dom] * 1e-3) / 1e9
                traffic_gbs = traffic / (launch_ms * 1e-3) / 1e9 if traffic else None
                # frac: what the fabric counters saw per launch / the HIP-event time of the launch / 8 TB/s where a PMC pass of THIS
                # build is on file -- an upper bound of the HBM fraction (Infinity-Cache hits are in it) -- else the byte model, which
                # is built never to exceed the counters (round 4's charged an index request per sample: 1.9 x the counters).  The
                # kernel does not live under the byte roof: `limiter` / `issue_frac` say what it waits for.
                issue = issue_fraction(pdir)
                if issue and issue.get("issue_frac") is not None:
                    limiter = ("integer issue and the request rate of the memory system under it, not byte bandwidth: the anchor dispatch of SeedExtend issues %d vector, "
                               "%d vector-memory-read and %d vector-memory-write instructions per 128-sample wavefront (SQ counters of this build, profiles/%s/sq_seed_extend.json; "
                               "round 5: 1 200 / 56 / 5 -- the right arms are resolved over the wavefront's lanes since round 6); at the measured %.2f cycles per wave64 int32 VALU "
                               "instruction and SIMD (scripts/valu_calib.hip, profiles/%s/calibration.json) they occupy %.0f %% of the dispatch's cycles (`issue_frac`), while %.0f %% of the "
                               "wave-cycles of its wavefronts are spent waiting.  One round trip less in the leaders' chain changed nothing, half the index load factor did (a wavefront "
                               "runs as long as its longest probe sequence).  SeedRest waits on dependent scattered reads outright; GroupedPairEvents / SmallPairEvents are register and LDS arithmetic."
                               % (issue["valu_per_wave"], issue.get("vmem_rd_per_wave", 0), issue.get("vmem_wr_per_wave", 0), PROFILE_ROUND, issue["cycles_per_valu"], PROFILE_ROUND,
                                  100 * issue["issue_frac"], 100 * issue.get("wait_frac", 0)))
                else:
                    limiter = ("integer issue under dependent scattered reads (SQ counters of round 6: 828 vector + 25 vector memory instructions per 128-sample wavefront "
                               "of SeedExtend, 57 % of its wave-cycles waiting); no SQ pass of this build on file, so no issue fraction is quoted")
                frac_traffic = round(traffic_gbs / HBM_PEAK_GBS, 5) if traffic_gbs else None
                roof = {"bound": "hbm",      # (the contract's two rooflines; what the kernel really waits for: `limiter`, `issue_frac`)
                        "limiter": limiter,
                        "issue_frac": issue.get("issue_frac") if issue else None, "issue": issue,
                        "kernel": "seed_extend (SeedExtend + SeedRest + GroupedPairEvents + SmallPairEvents: bytes and time both cover these four)" if dom == "seed_extend" else dom,
                        "achieved": round(traffic_gbs if traffic_gbs else alg_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": frac_traffic if frac_traffic is not None else round(alg_gbs / HBM_PEAK_GBS, 5),
                        "frac_basis": ("fabric-side counter traffic of this build per launch (FETCH_SIZE + WRITE_SIZE, corrected; Infinity-Cache hits included: an upper bound of the HBM bytes) / HIP-event time of the launch"
                                       if frac_traffic is not None else "byte model (no PMC pass of this build on file) / HIP-event time of the launch"),
                        "alg_model": "per (region, query genome) m/2 + 8 B per sampled K-mer (a 64-B index request per leader = one sample in eight; none for pairs that fit 128 bases), "
                                     "per region n/2, 64 B per sample SeedRest probes on its own, 16 B per event",
                        "alg_achieved": round(alg_gbs, 2), "alg_frac": round(alg_gbs / HBM_PEAK_GBS, 5),
                        "alg_bytes_per_launch": int(alg_step / launches), "events_per_step": int(events_step), "rest_samples_per_step": int(rest_step),
                        "alg_query_stream_bytes_per_launch": int(sum(r.get("alg_bytes_query", 0) for r in reports) / len(reports) / launches),
                        "launch_ms": round(launch_ms, 4), "launches_per_step": launches,
                        "traffic": traffic, "traffic_raw": traffic_raw, "traffic_note": traffic_note,
                        "traffic_includes_infinity_cache_hits": True,
                        "traffic_rate": round(traffic_gbs, 2) if traffic_gbs else None,
                        "model_exceeds_traffic": (alg_step / launches > traffic) if traffic else None,
                        "peak_random": RANDOM_PEAK_GBS, "frac_of_peak_random": round(traffic_gbs / RANDOM_PEAK_GBS, 5) if traffic_gbs else None,
                        "survey_8d": {"not_a_bound": True,
                                      "model": "m/4 + 16 m + 16 n per (region, query genome): one 8-byte probe and 16 bytes of state per query SUFFIX -- "
                                               "this engine samples every (minsize-15)th K-mer and keeps no dense per-genome state, so it moves a fraction of these "
                                               "bytes: the figure (a 'fraction' above 1) compares speeds, it bounds nothing",
                                      "bytes_per_launch": int(survey_step / launches), "achieved": round(survey_gbs, 2), "frac": round(survey_gbs / HBM_PEAK_GBS, 5)},
                        "anchor_launch": {"launch_ms": round(phases.get(dom, 0.0), 4), "survey_8d_bytes": int(b_alg * G),
                                          "alg_bytes": int(anchor_alg(G, m_avg, n_ref, anchor_stride, phases)),
                                          "achieved": round(anchor_alg(G, m_avg, n_ref, anchor_stride, phases) / (phases[dom] * 1e-3) / 1e9, 2) if phases.get(dom) else None}}
            # the device phases of a step against the byte roofline, largest first: algorithmic bytes per step (models in DESIGN.md 3,
            # from the counts the engine reports: events, reference positions, candidates, accepted MUM rows) / HIP-event time of
            # the phase on the engine's stream; reproducible from profiles/<round>/kernel_stats.csv (the kernels of a phase are named)
            ev, npos, ncand, nacc = totals.get("events", 0.0), totals.get("n_positions", 0.0), totals.get("n_candidates", 0.0), totals.get("n_accepted", 0.0)
            ngen = G + 1
            buckets = npos / 256.0 * G      # (pair, 256-position block) buckets of a step's searches
            models = {
                "seed_extend": ("SeedExtend + SeedRest + GroupedPairEvents + SmallPairEvents", alg_step, "(m + n)/2 per (region, query genome) + 64 B per sampled K-mer + 16 B per event"),
                "sort": ("EventBucketCount + exclusive scan + EventPlace + EventOrder (the events in order by (pair, 256-position block) buckets; tune bucket_sort = 0: CompactEvents + rocPRIM radix sort)",
                         56.0 * ev + 24.0 * buckets, "per event the key read for the count (8 B), the record read and written once (32 B) and read again by its bucket's thread (16 B); per bucket its counter written, scanned and read (24 B)"),
                "scan": ("CoarseFromBuckets + WaveSummary + WaveScan", 60.0 * ev + 12.0 * buckets, "16 B read twice (summary pass, scan pass) + 28 B of resolved state written per event; 8 B in, 4 B out per bucket for the readers' table"),
                "master_ep": ("MasterEPSeg (+ CoarseFill over the grouped events)", 12.0 * ev + 4.0 * npos, "12 B per event + 4 B per reference position"),
                "fold": ("FoldCandidates", 69.0 * ncand * G, "per (candidate, query genome): 28 B running state + 2 x (16 B winner + 4 B repeat length) in, 5 B out"),
                "compact": ("CompactCandidates + Dirty* + store append", (5.0 * ncand * G) + 3 * 5.0 * nacc * ngen, "5 B in per (candidate, genome); 5 B per (accepted row, genome) out, once more read by the overlap flags and once copied into the MUM store"),
                "settle": ("SettleClean + StoreMark + Collide* + SettleFlagged / Tangled", 20.0 * reports[-1]["anchors"] * ngen + 3 * (n_ref + 1) * ngen / 8.0, "4 B row entry + two 8-byte words per (anchor, genome) + three images of 1 bit per base cleared"),
                "index": ("IndexInsert", 24.0 * npos, "8 B slot + 16 B sequence window per reference position"),
                "repeat": ("RunLength + RepeatLength", 28.0 * npos, "16 B sequence window + 8 B slot + 4 B out per reference position"),
                "seeds": ("AnchorList + SeedCount + SeedPlace", 2 * 20.0 * reports[-1]["anchors"] * ngen, "per (anchor, genome) and side: 4 B row entry + two 8-byte image words"),
                "validate": ("Clusters* + ClusterValidate", 20.0 * max(0, reports[-1]["mums"] - reports[-1]["anchors"]) * ngen * 3, "per (candidate, genome): 4 B row entry + two image words, ~3 candidates per accepted MUM"),
                "chain": ("Foreign* + Chain*", 2 * 2 * 4.0 * reports[-1]["mums"] * ngen, "two passes over the MUM list, two rows of 4 B per genome per pair"),
            }
            # ... and beside every model what the fabric counters saw for the phase's kernels in one step of THIS build
            # (profiles/<round>/traffic_phases.json, scripts/profile_summary.py: FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc
            # passes, summed over the kernels of the phase; quoted only for the library it was measured on, on this workload)
            phase_traffic, phase_traffic_note = {}, "no PMC pass of this build on file (profiles/%s/traffic_phases.json)" % PROFILE_ROUND
            try:
                pj = json.load(open(os.path.join(pdir, "traffic_phases.json")))
                if args.workload == "bact200" and G == 200 and pj.get("so_sha256") == so_sha256():
                    phase_traffic, phase_traffic_note = pj.get("phases", {}), pj.get("note")
                elif pj.get("so_sha256") != so_sha256():
                    phase_traffic_note = "profiles/%s/traffic_phases.json was measured on another build of the engine: not quoted" % PROFILE_ROUND
            except (OSError, ValueError):
                pass
            ktable = []
            for k, (names, nbytes, model) in models.items():
                ms = totals.get(k, 0.0)
                if ms > 0:
                    row = {"phase": k, "kernels": names, "alg_bytes_per_step": int(nbytes), "ms_per_step": round(ms, 4), "achieved_GBs": round(nbytes / (ms * 1e-3) / 1e9, 1),
                           "frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "model": model}
                    t = phase_traffic.get(k)
                    if t:      # counters: raw = FETCH_SIZE + WRITE_SIZE; upper = 2 x FETCH_SIZE + WRITE_SIZE (coalesced 16 B/lane reads are tallied at half)
                        row.update({"traffic_bytes_per_step": int(t["raw_bytes_per_step"]), "traffic_upper_bytes_per_step": int(t["upper_bytes_per_step"]),
                                    "traffic_GBs": round(t["raw_bytes_per_step"] / (ms * 1e-3) / 1e9, 1), "traffic_frac": round(t["raw_bytes_per_step"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "traffic_over_alg": round(t["raw_bytes_per_step"] / nbytes, 2) if nbytes else None, "traffic_kernels": t.get("kernels")})
                    else:
                        row["traffic_bytes_per_step"] = None
                    ktable.append(row)
            ktable.sort(key=lambda r: -r["ms_per_step"])
            # where the step's wall time is not device time: ms_per_step - the HIP-event phases of every engine call of a step
            # (the list logic of the host, the round trips, the launches themselves)
            kernel_ms = sum(v for k, v in totals.items() if k not in ("call_wall",) + counts)
            ms_per_step = 1e3 * elapsed / args.steps
            # BASELINE's other single-GPU configurations, a child run each (their own process: nothing of them is resident here)
            other = None
            want_other = args.other_configs == "on" or (args.other_configs == "auto" and args.workload == "bact200" and not args.genomes and world == 1 and not args.inputs)
            if want_other:
                other = {}
                for wl, st in (("viral50", 20), ("bact200inv", 20), ("rearr500", 3)):
                    cmd = [sys.executable, os.path.abspath(__file__), "--workload", wl, "--steps", str(st), "--warmup", "1", "--cpu-sample", "0", "--other-configs", "off",
                           "--host-threads", str(args.host_threads)]
                    try:
                        pr = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                        lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
                        if pr.returncode == 0 and lines:
                            cj = json.loads(lines[-1])
                            other[wl] = {k: cj.get(k) for k in ("ms_per_step", "value", "steps", "host_cores_busy", "resident_route", "device_chain_steps", "mums", "lcbs", "anchors", "pcie_bytes_per_step", "host_ms_outside_kernels")}
                            other[wl]["workload"] = cj.get("config", {}).get("workload")
                        else:
                            other[wl] = {"error": "child exit code %d: %s" % (pr.returncode, pr.stderr[-300:])}
                    except subprocess.TimeoutExpired:
                        other[wl] = {"error": "did not finish within 600 s"}
            if roof is not None:
                roof["kernels"] = ktable
                roof["kernels_traffic_note"] = phase_traffic_note
            line = {
                "metric": "genomes/sec (MUM+LCB end-to-end)", "value": round(value, 4), "unit": "genomes/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
                "total_genomes": (1 if sharded else world) * G,      # N > 1, partition mode: N x G genomes in N alignments (weak scaling) -- not ONE job of G genomes on N GPUs, which is `sharded_strong`
                "device_ms_per_step": round(kernel_ms, 3), "host_ms_outside_kernels": round(ms_per_step - kernel_ms, 3),
                "materialize_ms": materialize_ms,      # (outside the timed region) the rows of the final MUM list fetched for the writer, once
                "device_chain_steps": sum(int(r.get("device_chain", 0)) for r in reports),      # steps whose phases C-D came from the device in one call (pm_store_chain_*)
                "other_configs": other,
                "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
                "config": {"workload": "%s: %d query genomes x %.2f Mb vs 1 reference (%s model, %s), --no-partition per GPU%s"
                           % (args.workload, G, n_ref / 1e6, dict(bact200="population").get(args.workload, "synthetic"),
                              ", ".join("%s=%s" % (k, v) for k, v in sorted(kw.items()) if k not in ("n", "n_genomes")),
                              "" if world == 1 else ("; ONE alignment sharded over %d ranks" % world if sharded else "; one partition per rank, %d ranks" % world)),
                           "genomes_per_gpu": G if not sharded else round(G / world, 2), "genome_bp": n_ref, "host_threads": args.host_threads, "tune": args.tune or None, "host_cpus_usable": usable_cpus(), "numa_node": numa_node, "parallelism": ("sharded x%d (engine RCCL: all-reduce(min) + all-gather per engine call)" % world) if sharded else "partition-per-gpu x%d" % world},
                "n_ranks_seen_by_rccl": rccl_ranks, "per_rank": per_rank,
                "sharded_strong": sharded_strong,
                "step_ms": step_ms[:40], "step_ms_quantiles": [sorted(step_ms)[int(q * (len(step_ms) - 1))] for q in (0.0, 0.1, 0.5, 0.9, 0.99, 1.0)], "step_ms_slowest": [dict(step=i, ms=x, path_ms=round(1e3 * reports[i]["path_s"], 2), anchor_ms=round(1e3 * reports[i]["anchor_s"], 2), extend_ms=round(1e3 * reports[i]["extend_s"], 2), lcb_ms=round(1e3 * reports[i]["lcb_s"], 2), setup_ms=round(1e3 * reports[i]["setup_s"], 2), call_wall=round(reports[i]["engine_ms"].get("call_wall", 0), 2), phases={k: round(v, 2) for k, v in reports[i]["engine_ms"].items() if isinstance(v, float) and v > 0.5 and v < 100 and k != "call_wall"}) for x, i in sorted(((x, i) for i, x in enumerate(step_ms)), reverse=True)[:6]], "host_cores_busy": round(host_cores_busy, 2),
                # what the engine moved over the host link per step (pm_session_traffic: every copy it issued), and which route the
                # steps took (resident: MUM rows, layout and regions stayed on the device, parsnp_amd/csrc/host/resident.cpp)
                "pcie_bytes_per_step": {"h2d": int(sum(r.get("h2d_bytes", 0) for r in reports) / len(reports)),
                                        "d2h": int(sum(r.get("d2h_bytes", 0) for r in reports) / len(reports))},
                "resident_route": {"steps": sum(int(r.get("resident", 0)) for r in reports), "left_and_repeated_on_the_host_route": sum(int(r.get("resident_retry", 0)) for r in reports),
                                   "why_not": next((r.get("resident_why") for r in reversed(reports) if r.get("resident_why")), None)},      # (the last step's reason when a step did not stay on the route)
                "core_bp_aligned": core_bp_total,
                "core_bp_in_every_partition": merged_bp,
                "mums": rep["mums"], "anchors": rep["anchors"], "lcbs": rep["lcbs"],
                "split_s": {"path": rep["path_s"], "setup": rep.get("setup_s"), "anchor": rep["anchor_s"], "extend": rep["extend_s"], "lcb": rep["lcb_s"],
                            "engine_calls_wall": rep["finder_s"], "ingest": rep["ingest_s"], "upload": rep["upload_s"],
                            "output": output_s, "generate": gen_s},
                "cold": None if cold_step_s is None else {
                    # one whole process, nothing resident: FASTA ingest + upload/packing + the first pass + XMFA/log writing
                    "cold_step_s": round(cold_step_s, 4),
                    "wall_s": round(rep["ingest_s"] + rep["upload_s"] + cold_step_s + output_s, 4),
                    "cold_wall_genomes_per_s": round(G / (rep["ingest_s"] + rep["upload_s"] + cold_step_s + output_s), 2)},
                "host_split_s": rep.get("host_split_s"),
                "engine_ms": {k: round(v, 3) for k, v in rep["engine_ms"].items()},
                "anchor_launch_ms": {k: round(v, 3) for k, v in phases.items()},
                "regions": {"processed": rep["regions_processed"], "engine_calls": rep["finder_calls"], "cache_misses": rep["cache_misses"],
                            "speculative_rounds": rep["spec_rounds"]},
                "roofline": roof,
                "build": {"so_sha256": so_sha256(), "engine_src_sha256": engine_src_sha256()},      # what ran: profiles/<round>/*.json carry the same stamps
                "cpu_baseline": cpu_baseline(workdir, rp, qs, args.cpu_sample) if (args.cpu_sample > 0 and world == 1) else None,
            }
            print(json.dumps(line))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
    finally:
        if not args.keep:
            shutil.rmtree(workdir, ignore_errors=True)


if __name__ == "__main__":
    main()
