#!/bin/bash
# SQ counters of the event search's kernels (and the next two device phases) under the default bench workload: two passes of <= 8
# counters (gpurun refuses --pmc together with the trace domains other than --kernel-trace).  Writes gpurun_out/sq/summary.json:
# per kernel the counters of its LONGEST dispatch (the anchor launch), one record per XCD's sampled shader engine, stamped with
# the sha256 of the library.   gpurun --timeout 600 -- 'bash scripts/sqcounters.sh'
REPO=$(pwd); OUT=$REPO/gpurun_out/sq; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
B="python $REPO/bench.py --steps 2 --warmup 1 --cpu-sample 0 --other-configs off"
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_LDS --kernel-trace -d $OUT/a -o a -- $B > $OUT/a.log 2>&1
timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d $OUT/b -o b -- $B > $OUT/b.log 2>&1
cd $REPO
python - <<'PY'
import sqlite3, glob, json, hashlib, os, re
kernels = ["SeedExtend", "SeedRest", "SmallPairEvents", "MasterEP", "FoldCandidates", "WaveScan", "GroupedPairEvents", "IndexInsert"]
def src_sha():
    h = hashlib.sha256()
    for f in ("kernels.h", "store_kernels.h", "engine_core.h", "engine_hip.hip", "abi_glue.h"):
        h.update(open("parsnp_amd/csrc/engine/" + f, "rb").read())
    return h.hexdigest()
out = {"so_sha256": hashlib.sha256(open("parsnp_amd/lib/libparsnp_hip.so", "rb").read()).hexdigest(), "engine_src_sha256": src_sha(), "workload": "bench.py default (200 x 5 Mb), 2 steps + 1 warm-up",
       "note": "per kernel: counters of its dispatches; `anchor` = the dispatch records of the longest launch (the anchor call); a record = one XCD's sampled shader engine", "kernels": {}}
for d in ("a", "b"):
    for db in glob.glob("gpurun_out/sq/%s/**/*_results.db" % d, recursive=True):
        c = sqlite3.connect(db)
        for k in kernels:
            try:
                rows = c.execute("select counter_name, counter_value from pmc_events where name like ?", ("%pm::" + k + ">%",)).fetchall()
            except Exception as e:
                print(d, k, "no pmc table:", e); continue
            by = {}
            for name, v in rows:
                by.setdefault(name, []).append(float(v))
            kk = out["kernels"].setdefault(k, {"counters": {}})
            for name, vs in by.items():
                vs.sort()
                kk["counters"][name] = {"records": len(vs), "max_per_record": vs[-1], "avg_per_record": sum(vs) / len(vs), "sum": sum(vs)}
            durs = sorted(dur for (dur,) in c.execute("select duration from kernels where name like ?", ("%pm::" + k + ">%",)))
            if durs:
                kk["dispatch_ns"] = {"n": len(durs), "max": durs[-1], "avg": sum(durs) / len(durs)}
json.dump(out, open("gpurun_out/sq/summary.json", "w"), indent=1)
for k, v in out["kernels"].items():
    cs = v["counters"]
    g = lambda n: cs.get(n, {}).get("max_per_record", 0.0)
    print(k, v.get("dispatch_ns"))
    if g("SQ_WAVES"):
        print("   per wave (longest launch): VALU %.0f SALU %.0f VMEM_RD %.1f VMEM_WR %.1f SMEM %.1f LDS %.1f" % tuple(g(n) / g("SQ_WAVES") for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_LDS")))
    if g("SQ_BUSY_CYCLES"):
        print("   busy cycles %.0f: VALU active %.2f, scalar active %.2f, VMEM active %.2f of them; wave cycles %.0f, waiting (any) %.2f, waiting on instruction issue %.2f of the wave cycles" % (
            g("SQ_BUSY_CYCLES"), g("SQ_ACTIVE_INST_VALU") / g("SQ_BUSY_CYCLES"), g("SQ_ACTIVE_INST_SCA") / g("SQ_BUSY_CYCLES"), g("SQ_ACTIVE_INST_VMEM") / g("SQ_BUSY_CYCLES"),
            g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY") / max(1.0, g("SQ_WAVE_CYCLES")), g("SQ_WAIT_INST_ANY") / max(1.0, g("SQ_WAVE_CYCLES"))))
PY
tail -n 2 $OUT/a.log $OUT/b.log
rm -rf $OUT/a $OUT/b
