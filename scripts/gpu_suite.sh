#!/bin/bash
# the whole GPU suite + a short bench line:  gpurun --timeout 1500 -- 'bash scripts/gpu_suite.sh'
mkdir -p gpurun_out/r5
timeout 120 python scripts/smoke_core.py 2>&1 | tail -2
timeout 1300 python -m pytest tests -m gpu -q 2>&1 | tail -6
timeout 300 python bench.py --steps 60 --warmup 5 --cpu-sample 0 --other-configs off > gpurun_out/r5/bench_suite.json 2> gpurun_out/r5/bench_suite.err; tail -1 gpurun_out/r5/bench_suite.json | python scripts/benchline.py
bash scripts/profile_stats.sh > gpurun_out/r5/stats.log 2>&1; python - <<'PY'
import csv,re
rows=list(csv.DictReader(open('gpurun_out/prof_stats/summary/kernel_stats.csv')))
tot=0; out=[]
for r in rows:
    n=r['Name']
    if 'gap_align' in n or 'PackStrand' in n or 'StoreRowsOut' in n: continue
    m=re.search(r'pm::(\w+)>',n); short=m.group(1) if m else n[:50]
    t=int(r['TotalDurationNs'])/4/1e3; tot+=t; out.append((t,short))
out.sort(reverse=True)
print(" | ".join("%s %.0f"%(s,t) for t,s in out[:32])); print("kernel us per step %.0f"%tot)
PY
