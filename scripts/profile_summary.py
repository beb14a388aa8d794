#!/usr/bin/env python3
"""Condense the rocprofv3 output of scripts/profile.sh into the small files kept under profiles/:
  kernel_stats.csv           rocprofv3 --stats table of the bench command (per-kernel calls / total / average)
  pmc_per_kernel.json        FETCH_SIZE and WRITE_SIZE (separate passes) summed per kernel, with dispatch counts
  calibration.json           FETCH_SIZE / WRITE_SIZE of scripts/hbm_calib.hip's three patterns against their known byte counts
  traffic_seed_extend.json   HBM bytes per launch of SeedExtend, corrected with the calibration of its own access pattern
"""
import csv
import glob
import json
import os
import re
import sys


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    return hits[0] if hits else None


def short(name):
    """pm_kernel<pm::X> and pm_wave_kernel<pm::X> -> X; the engine's own plain kernels and the library's by their function name"""
    m = re.search(r"pm_(?:wave_)?kernel<pm::(\w+)>", name)
    if m:
        return m.group(1)
    m = re.search(r"(pm_fill16|pm_fill_many|gap_align_kernel)", name)
    if m:
        return m.group(1)
    m = re.search(r"rocprim::\w+::detail::(?:trampoline_kernel<rocprim::\w+::detail::wrapped_(\w+?)_config|(\w+?)<)", name)
    if m:
        return "rocprim_" + (m.group(1) or m.group(2))
    name = name.split("(")[0].strip()
    return name[5:] if name.startswith("void ") else name


# the device phases of a bench step (bench.py: `roofline.kernels`) and the kernels they launch -- the fabric-side counter bytes of a
# phase are the sums over these (traffic_phases.json); a rocPRIM kernel belongs to the phase that calls the primitive most
PHASES = {
    "seed_extend": ["SeedExtend", "SeedRest", "GroupedPairEvents", "SmallPairEvents"],
    "sort": ["SliceOffsets", "CompactEvents", "PairBucketBase", "EventBucketCount", "EventPlace", "EventOrder", "rocprim_radix_sort_onesweep", "rocprim_radix_sort_block_sort", "rocprim_merge_sort_block_merge", "rocprim_merge_sort_block_sort"],
    "scan": ["PairBounds", "GroupedBounds", "CoarseFromBuckets", "WaveSummary", "WaveScan"],
    "master_ep": ["CoarseFill", "MasterEPSeg", "MasterEP"],
    "fold": ["FoldCandidates", "CandMark", "CandWrite"],
    "compact": ["OkCount", "CompactCandidates", "CompactSp", "DirtyExtent", "DirtyPrefix", "DirtyMark", "DirtyMerge"],
    "settle": ["SettleClean", "StoreMarkOrdered", "StoreMark", "LayoutSentinel", "CollideMark", "CollideTest", "CollideClear", "CountTangled", "SettleFlagged", "SettleTangled",
               "TangleOwner", "TangleSettle", "TangleClear", "StoreInfoOut"],
    "index": ["IndexInsert"],
    "repeat": ["RunLength", "RepeatLength"],
    "seeds": ["ChainFlag", "AnchorList", "SeedCount", "SeedPlace", "SeedWalk"],
    "validate": ["ClustersDisjoint", "ClusterExtents", "ClusterInvolved", "ClusterDefer", "ReaderMark", "MarkerLook", "ReaderLook", "ClusterValidate", "OutsideWriteCheck", "StageGate"],
    "chain": ["ForeignBound", "ForeignScan", "ForeignDecideHits", "ChainKeys", "ChainJudge", "ChainJudgeReverse", "ChainHeads", "ChainLcbSum", "ChainDissolve", "ChainUnmark", "ChainCompact", "ChainFill", "ChainOut"],
}


def db_rows(d, sql):
    """rows of a query against the rocpd database rocprofv3 wrote into d (its default output format), or None"""
    import sqlite3
    path = find(d, "_results.db")
    if not path:
        return None
    return list(sqlite3.connect(path).execute(sql))


def pmc(d):
    """-> {kernel: {"dispatches": n, "sum": counter total}} from the rocpd database or *_counter_collection.csv"""
    out = {}
    rows = db_rows(d, "select name, counter_name, counter_value from pmc_events")
    if rows is not None:
        for name, counter, value in rows:
            e = out.setdefault(short(name), {"dispatches": 0, "sum": 0.0, "counter": counter})
            e["dispatches"] += 1
            e["sum"] += float(value)
        return out
    path = find(d, "counter_collection.csv")
    if not path:
        return out
    for row in csv.DictReader(open(path)):
        k = short(row["Kernel_Name"])
        e = out.setdefault(k, {"dispatches": 0, "sum": 0.0, "counter": row["Counter_Name"]})
        e["dispatches"] += 1
        e["sum"] += float(row["Counter_Value"])
    return out


def traced_steps(out):
    """steps + warm-up of the bench run the --stats pass traced (its own JSON line)"""
    try:
        bj = [l for l in open(os.path.join(out, "summary", "bench_under_rocprof.json")).read().splitlines() if l.startswith("{")][-1]
        d = json.loads(bj)
        return int(d["steps"]) + int(d["warmup"])
    except Exception:   # noqa: BLE001
        return 4


def lib_sha():
    """sha256 of the libparsnp_hip.so the passes ran on: bench.py quotes the traffic only for this binary"""
    import hashlib
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "parsnp_amd", "lib", "libparsnp_hip.so")
    return hashlib.sha256(open(p, "rb").read()).hexdigest() if os.path.exists(p) else None


def engine_src_sha():
    """sha256 over the sources of the engine translation unit (the kernels the passes measured): unlike the library hash
    it survives edits to the other translation unit of libparsnp_hip.so (the gap aligner)"""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "parsnp_amd", "csrc", "engine")
    h = hashlib.sha256()
    for f in ("kernels.h", "store_kernels.h", "engine_core.h", "engine_hip.hip", "abi_glue.h"):
        h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()


def main():
    out = sys.argv[1]
    summ = os.path.join(out, "summary")
    os.makedirs(summ, exist_ok=True)
    ks = find(os.path.join(out, "stats"), "kernel_stats.csv")
    stats = {}
    rows = db_rows(os.path.join(out, "stats"), "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                   "from kernels group by name order by sum(duration) desc")
    if rows is not None:
        total = sum(r[2] for r in rows) or 1
        with open(os.path.join(summ, "kernel_stats.csv"), "w") as f:
            f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
            for name, calls, tot, avg, mn, mx in rows:
                f.write('"%s",%d,%d,%.3f,%.2f,%d,%d\n' % (name, calls, tot, avg, 100.0 * tot / total, mn, mx))
                stats[short(name)] = {"calls": calls, "avg_ms": avg / 1e6, "total_ms": tot / 1e6, "min_ms": mn / 1e6, "max_ms": mx / 1e6}
    elif ks:
        rows = list(csv.DictReader(open(ks)))
        with open(os.path.join(summ, "kernel_stats.csv"), "w") as f:
            f.write(open(ks).read())
        for r in rows:
            stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "total_ms": float(r["TotalDurationNs"]) / 1e6}
    # where the device waits for the host: the idle time between consecutive kernels of the traced run, by (kernel before, kernel
    # after) -- what `host_ms_outside_kernels` of the bench line is made of
    try:
        tr = db_rows(os.path.join(out, "stats"), "select name, start, end from kernels order by start")
        if tr:
            gaps = {}
            total_idle = 0.0
            busy = sum(e - b for _, b, e in tr)
            for (n0, b0, e0), (n1, b1, e1) in zip(tr, tr[1:]):
                g = b1 - e0
                if g > 5000:      # > 5 us: not back-to-back
                    k = short(n0) + " -> " + short(n1)
                    x = gaps.setdefault(k, {"count": 0, "total_us": 0.0, "max_us": 0.0})
                    x["count"] += 1; x["total_us"] += g / 1e3; x["max_us"] = max(x["max_us"], g / 1e3)
                    if g < 50e6:      # (the pauses between warm-up, steps and the writer are not part of a step)
                        total_idle += g / 1e3
            top = sorted(gaps.items(), key=lambda kv: -kv[1]["total_us"])[:40]
            json.dump({"note": "idle time of the device between consecutive kernels of the traced bench run (warm-up + steps + one XMFA write), gaps > 5 us, by (kernel before -> kernel after); gaps >= 50 ms (between steps, around the writer) are listed but not summed",
                       "kernel_busy_ms": busy / 1e6, "idle_ms_in_gaps_below_50ms": total_idle / 1e3, "gaps": [dict(pair=k, **v) for k, v in top]},
                      open(os.path.join(summ, "idle_gaps.json"), "w"), indent=1)
    except Exception as e:   # noqa: BLE001
        print("no gap analysis:", e)
    fetch, write = pmc(os.path.join(out, "fetch")), pmc(os.path.join(out, "write"))
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB
    per = {}
    for k in sorted(set(fetch) | set(write)):
        per[k] = {"dispatches": fetch.get(k, write.get(k))["dispatches"],
                  "fetch_bytes": fetch[k]["sum"] * 1024 if k in fetch else None,
                  "write_bytes": write[k]["sum"] * 1024 if k in write else None}
    json.dump(per, open(os.path.join(summ, "pmc_per_kernel.json"), "w"), indent=1)
    # the same per device phase of a bench step (the PMC passes profile ONE step with no warm-up): what bench.py sets beside the byte
    # model of every phase.  raw = FETCH_SIZE + WRITE_SIZE as counted; upper = 2 x FETCH_SIZE + WRITE_SIZE (if every read were a
    # wide coalesced 16 B/lane stream, which the counter tallies at half: calibration.json) -- the truth lies between the two
    ph = {}
    for name, kernels in PHASES.items():
        f = sum(per[k]["fetch_bytes"] or 0 for k in kernels if k in per)
        w = sum(per[k]["write_bytes"] or 0 for k in kernels if k in per)
        if f or w:
            ph[name] = {"kernels": [k for k in kernels if k in per], "fetch_bytes_per_step": f, "write_bytes_per_step": w, "raw_bytes_per_step": f + w, "upper_bytes_per_step": 2 * f + w,
                        "rocprof_ms_per_step": round(sum(stats[k]["total_ms"] for k in kernels if k in stats) / max(1, traced_steps(out)), 4)}
    named = {k for ks_ in PHASES.values() for k in ks_}
    json.dump({"note": "fabric-side counter bytes per device phase of ONE bench step (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of `bench.py --steps 1 --warmup 0`; Infinity-Cache hits included); "
                       "raw = FETCH_SIZE + WRITE_SIZE, upper = 2 x FETCH_SIZE + WRITE_SIZE (wide coalesced reads are tallied at half: calibration.json); rocprof_ms_per_step from the --stats pass",
               "so_sha256": lib_sha(), "engine_src_sha256": engine_src_sha(), "phases": ph,
               "kernels_in_no_phase": {k: v for k, v in per.items() if k not in named and ((v["fetch_bytes"] or 0) + (v["write_bytes"] or 0)) > 1e6}},
              open(os.path.join(summ, "traffic_phases.json"), "w"), indent=1)
    cal = {}
    known = {}
    try:
        known = json.loads(open(os.path.join(summ, "calib_bytes.json")).read().strip().splitlines()[-1])
    except Exception as e:   # noqa: BLE001
        print("no calibration byte counts:", e)
    cf, cw = pmc(os.path.join(out, "calib")), pmc(os.path.join(out, "calibw"))
    for k in ("calib_stream16", "calib_gather16", "calib_gather8"):
        if k in cf and k in known:
            per_launch = cf[k]["sum"] * 1024 / cf[k]["dispatches"]
            cal[k] = {"requested_bytes": known[k], "fetch_size_bytes": per_launch, "counter_over_requested": per_launch / known[k]}
            if k != "calib_stream16":
                cal[k]["counter_bytes_per_lane"] = per_launch / known["gather_lanes"]
            if k in cw:
                cal[k]["write_size_bytes"] = cw[k]["sum"] * 1024 / cw[k]["dispatches"]
    try:      # the int32 VALU issue rate of a SIMD (scripts/valu_calib.hip): what bench.py's `issue_frac` prices SQ_INSTS_VALU with
        cal["valu"] = json.loads(open(os.path.join(summ, "valu_calib.json")).read().strip().splitlines()[-1])
    except Exception as e:   # noqa: BLE001
        print("no VALU calibration:", e)
    json.dump(cal, open(os.path.join(summ, "calibration.json"), "w"), indent=1)
    # the event search of one engine call = SeedExtend (index-seeded samples) + SmallPairEvents (pairs that fit 128 bases,
    # compared in registers): bench.py times them together as the `seed_extend` phase, so they are summed here too
    se, sp, sr, gp = per.get("SeedExtend"), per.get("SmallPairEvents"), per.get("SeedRest"), per.get("GroupedPairEvents")
    if se and se["fetch_bytes"] is not None and se["write_bytes"] is not None:
        n = se["dispatches"]
        # (GroupedPairEvents too: bench.py's `seed_extend` phase time holds it -- its `grouped_events` mark is added to the phase)
        fetch = se["fetch_bytes"] + sum(x["fetch_bytes"] for x in (sp, sr, gp) if x and x["fetch_bytes"])
        write = se["write_bytes"] + sum(x["write_bytes"] for x in (sp, sr, gp) if x and x["write_bytes"])
        st_se, st_sp, st_sr, st_gp = stats.get("SeedExtend", {}), stats.get("SmallPairEvents", {}), stats.get("SeedRest", {}), stats.get("GroupedPairEvents", {})
        calls = st_se.get("calls")
        # The counters sit on the fabric side of the L2 (TCC_EA0_RDREQ x 64 B): Infinity-Cache hits are INCLUDED, so this is
        # fabric traffic, an upper bound of the HBM bytes.  Calibration (calibration.json): scattered 8-16 B probes count 64 B per
        # lane -- exact for the index / filter / next[] / sequence-block requests -- while a wide coalesced stream is tallied at
        # half its bytes (the guide's gfx950 factor, reproduced by calib_stream16).  The one coalesced stream of this kernel that
        # reaches the fabric is the query pieces (m/2 per pair; the reference windows hit the L2), so the missing half of that
        # stream is added back: corrected = FETCH_SIZE + 0.5 x query-stream bytes + WRITE_SIZE.
        # (everything per bench STEP first: the passes profile one step, and a step is several engine calls -- anchor, the
        # batch computed ahead, the leftover regions -- not all of which launch SeedExtend; bench.py divides by its own count)
        qstream = None
        bench_launches = None
        try:
            bj = [l for l in open(os.path.join(summ, "bench_plain.json")).read().splitlines() if l.startswith("{")][-1]
            rl = json.loads(bj)["roofline"]
            bench_launches = rl.get("launches_per_step")
            qstream = rl.get("alg_query_stream_bytes_per_launch") * bench_launches
        except Exception as e:   # noqa: BLE001
            print("no bench line for the stream correction:", e)
        corr = 0.5 * qstream if qstream else 0.0
        n_se = n
        n = bench_launches or n
        corr = corr / n
        t = {"kernel": "seed_extend = SeedExtend + SeedRest + GroupedPairEvents + SmallPairEvents", "workload": "bact200, the event search of every engine call of one bench step (anchor + recursion)",
             "dispatches": n, "fetch_bytes_per_launch": fetch / n, "write_bytes_per_launch": write / n,
             "raw": {"FETCH_SIZE": fetch / n, "WRITE_SIZE": write / n},
             "seed_extend_dispatches_per_step": n_se, "launches_per_step": n,
             "hbm_bytes_per_step": fetch + write + corr * n,
             "query_stream_bytes_per_launch": (qstream / n) if qstream else None,
             "hbm_bytes_per_launch": (fetch + write) / n + corr,
             "correction": "FETCH_SIZE + 0.5 x query-stream bytes (coalesced 16 B/lane streams are tallied at half, calibration.json calib_stream16) + WRITE_SIZE; scattered probes count 64 B per lane and are taken as they are",
             "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, this binary; fabric-side counters: Infinity-Cache hits included (upper bound of HBM bytes); coalesced query stream corrected x2",
             "rocprof_avg_launch_ms": ((st_se.get("total_ms", 0) + st_sp.get("total_ms", 0) + st_sr.get("total_ms", 0) + st_gp.get("total_ms", 0)) / calls) if calls else None,
             "rocprof_grouped_pair_events_avg_ms": st_gp.get("avg_ms"),
             "rocprof_calls": calls,
             "rocprof_seed_extend_avg_ms": st_se.get("avg_ms"), "rocprof_seed_extend_max_ms": st_se.get("max_ms"), "rocprof_seed_rest_avg_ms": st_sr.get("avg_ms"), "rocprof_seed_rest_max_ms": st_sr.get("max_ms"),
             "rocprof_small_pair_events_avg_ms": st_sp.get("avg_ms"),
             "so_sha256": lib_sha(), "engine_src_sha256": engine_src_sha()}
        json.dump(t, open(os.path.join(summ, "traffic_seed_extend.json"), "w"), indent=1)
    print(json.dumps({"stats": {k: v for k, v in stats.items() if v["total_ms"] > 1}, "calibration": cal, "seed_extend": per.get("SeedExtend"), "small_pair_events": per.get("SmallPairEvents")}, indent=1))


if __name__ == "__main__":
    main()
