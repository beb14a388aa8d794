#!/bin/bash
# round 5: the order check (store_kernels.h: ForeignBound and the kernels around it) on the GPU box -- its regression tests, the
# resident-route parity tests, 200 x 5 Mb against the reference's golden, a bench line, how many candidates it noted and left to
# the scan, the kernel table
mkdir -p gpurun_out/r5
O=gpurun_out/r5
timeout 600 python -m pytest tests/test_fuzz_vs_reference.py -m gpu -x -q -k "order_of_reads or resident_route_on_gpu or inversions" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "resident_route or twins" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_big.py -m gpu -x -q -k "bact200" 2>&1 | tail -3
PARSNP_BENCH_LOG=$O/order.log timeout 200 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --other-configs off --tune order_debug=1 > /dev/null 2>&1; grep "order check" $O/order.log | tail -1
timeout 300 python bench.py --steps 60 --warmup 5 --cpu-sample 0 --other-configs off > $O/bench_order.json 2> $O/bench_order.err; tail -1 $O/bench_order.json | python scripts/benchline.py | head -2
bash scripts/profile_stats.sh > $O/stats.log 2>&1; python - <<'PY'
import csv,re
rows=list(csv.DictReader(open('gpurun_out/prof_stats/summary/kernel_stats.csv')))
tot=0; out=[]
for r in rows:
    n=r['Name']
    if 'gap_align' in n or 'PackStrand' in n or 'StoreRowsOut' in n: continue
    m=re.search(r'pm::(\w+)>',n); short=m.group(1) if m else n[:50]
    t=int(r['TotalDurationNs'])/4/1e3; tot+=t; out.append((t,short))
out.sort(reverse=True)
print(" | ".join("%s %.0f"%(s,t) for t,s in out[:40])); print("kernel us per step %.0f"%tot)
PY
