#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/fillprof; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace -d $OUT/t -o t -- python $REPO/bench.py --steps 2 --warmup 1 --cpu-sample 0 > $OUT/bench.json 2> $OUT/err.log
cd $REPO
python - <<'PY'
import sqlite3, glob
for db in glob.glob("gpurun_out/fillprof/t/**/*_results.db", recursive=True):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    print(cols)
    q = "select name, grid_x, workgroup_x, duration, start from kernels where name like '%pm_fill16%' or name like '%fillBuffer%' or name like '%copyBuffer%' order by start"
    try:
        rows = list(c.execute(q))
    except Exception as e:
        print("query failed", e); rows = []
    # only the last step's window: print all from the last 120 rows
    for name, gx, wx, dur, st in rows[-130:]:
        short = "fill16" if "pm_fill16" in name else ("fillBuf" if "fillBuffer" in name else "copyBuf")
        print(short, gx, wx, dur)
PY
