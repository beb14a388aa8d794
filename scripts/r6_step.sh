#!/bin/bash
# round 6: one GPU call of the build -> measure loop.  ./scripts/r6_step.sh [tests|bench|rearr|all ...]: the resident-route device
# tests, the headline bench line, config 3 with inversions and config 5 (with the host's laps of one step each)
O=gpurun_out/r6; mkdir -p $O
want=" ${*:-all} "
has() { [[ "$want" == *" all "* || "$want" == *" $1 "* ]]; }
if has tests; then
  timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_vs_reference.py -m gpu -x -q -k "resident_route or inversions or clusters_in_another or order_of_reads or fuzz_resident_route_on_gpu or waiting_clusters or tied_mums or test_events" > $O/tests_a.log 2>&1; tail -3 $O/tests_a.log
fi
if has bench; then
  timeout 300 python bench.py --steps 100 --warmup 5 --cpu-sample 0 --other-configs off > $O/bench_bact200.json 2> $O/bench_bact200.err; tail -1 $O/bench_bact200.json | python scripts/benchline.py
fi
if has inv; then
  timeout 300 python bench.py --workload bact200inv --steps 20 --warmup 2 --cpu-sample 0 --other-configs off > $O/bench_inv.json 2> $O/bench_inv.err; tail -1 $O/bench_inv.json | python scripts/benchline.py
  PARSNP_BENCH_LOG=$O/inv_laps.log PARSNP_DEBUG_TIMERS=1 timeout 300 python bench.py --workload bact200inv --steps 2 --warmup 1 --cpu-sample 0 --other-configs off > /dev/null 2>&1
  grep -E "^\[(resident|anchors|extend|lcb)" $O/inv_laps.log | tail -60 > $O/inv_laps.txt
fi
if has rearr; then
  timeout 600 python bench.py --workload rearr500 --steps 4 --warmup 1 --cpu-sample 0 --other-configs off > $O/bench_rearr500.json 2> $O/bench_rearr500.err; tail -1 $O/bench_rearr500.json | python scripts/benchline.py
  PARSNP_BENCH_LOG=$O/rearr_laps.log PARSNP_DEBUG_TIMERS=1 timeout 600 python bench.py --workload rearr500 --steps 1 --warmup 1 --cpu-sample 0 --other-configs off > /dev/null 2>&1
  grep -E "^\[(resident|anchors|extend|lcb)" $O/rearr_laps.log | tail -120 > $O/rearr_laps.txt
fi
if has prof; then      # per-kernel times of config 3 with inversions and of config 5 (rocprofv3 --kernel-trace --stats, a few steps each)
  for wl in ${PROF_WL:-bact200inv rearr500}; do
    P=$GRAFT_REPO_ROOT/$O/prof_$wl; rm -rf $P; mkdir -p $P
    ( cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --stats -d $P -o ks -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 3 --warmup 1 --cpu-sample 0 --other-configs off > $P/bench.json 2> $P/err.log )
    python - "$P" "$wl" <<'PY'
import glob, sqlite3, sys, re, os
P, wl = sys.argv[1], sys.argv[2]
db = sorted(glob.glob(os.path.join(P, "**", "*_results.db"), recursive=True))
rows = []
if db:
    c = sqlite3.connect(db[0])
    tables = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    t = "kernels" if "kernels" in tables else next((x for x in tables if x.startswith("kernels")), None)
    rows = list(c.execute("select name, count(*), sum(duration), max(duration) from %s group by name order by sum(duration) desc" % t)) if t else []
def short(n):
    m = re.search(r"pm_(?:wave_)?kernel<pm::(\w+)>", n)
    return m.group(1) if m else n.split("(")[0][-60:]
with open(os.path.join(P, "..", "kernels_%s.txt" % wl), "w") as f:
    f.write("# %s: bench.py --steps 3 --warmup 1 under rocprofv3 --kernel-trace (4 steps in all); ms per STEP, calls per step, longest dispatch ms\n" % wl)
    for n, calls, tot, mx in rows[:45]:
        f.write("%-34s %8.3f %7.1f %8.3f\n" % (short(n), tot / 4e6, calls / 4.0, mx / 1e6))
print(open(os.path.join(P, "..", "kernels_%s.txt" % wl)).read()[:2500])
PY
    rm -rf $P
  done
fi
