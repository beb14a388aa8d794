#!/bin/bash
# SQ counters of SeedExtend on the 40-genome anchor launch of scripts/seedexp.py (two passes of <= 8 counters)
REPO=$(pwd); OUT=$REPO/gpurun_out/sq; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_LDS --kernel-trace -d $OUT/a -o a -- python $REPO/scripts/seedexp.py "" 0 > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d $OUT/b -o b -- python $REPO/scripts/seedexp.py "" 0 > $OUT/b.log 2>&1
cd $REPO
python - <<'PY'
import sqlite3, glob
for d in ("a", "b"):
    for db in glob.glob("gpurun_out/sq/%s/**/*_results.db" % d, recursive=True):
        c = sqlite3.connect(db)
        rows = c.execute("select name, counter_name, sum(counter_value), count(*) from pmc_events where name like '%SeedExtend%' group by counter_name").fetchall()
        for r in rows: print(r[1], r[2] / r[3], "per launch over", r[3], "launches")
        print([ (n, d) for n, d in c.execute("select name, duration from kernels where name like '%SeedExtend%'")][:3])
PY
tail -n 3 $OUT/a.log; tail -n 3 $OUT/b.log
