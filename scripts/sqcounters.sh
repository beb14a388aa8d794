#!/bin/bash
# SQ / TCC counters of SeedExtend's anchor launch under the default bench workload (three passes of <= 8 counters).
# Measurement helper; prints per-launch counter values of the launches of SeedExtend (the long ones are the anchor launches).
REPO=$(pwd); OUT=$REPO/gpurun_out/sq; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
B="python $REPO/bench.py --steps 2 --warmup 1 --cpu-sample 0"
timeout 90 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_LDS --kernel-trace -d $OUT/a -o a -- $B > $OUT/a.log 2>&1
timeout 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d $OUT/b -o b -- $B > $OUT/b.log 2>&1
cd $REPO
python - <<'PY'
import sqlite3, glob
for d in ("a", "b", "c"):
    for db in glob.glob("gpurun_out/sq/%s/**/*_results.db" % d, recursive=True):
        c = sqlite3.connect(db)
        try:
            rows = c.execute("select counter_name, max(counter_value), avg(counter_value), count(*) from pmc_events where name like '%SeedExtend%' group by counter_name").fetchall()
        except Exception as e:
            print(d, "no pmc table:", e); continue
        for r in rows: print(d, r[0], "max", r[1], "avg", r[2], "over", r[3], "dispatch records")
        print(d, sorted([dur for (dur,) in c.execute("select duration from kernels where name like '%SeedExtend%'")])[-3:])
PY
tail -n 2 $OUT/a.log $OUT/b.log $OUT/c.log
