#!/bin/bash
# round 6: the whole GPU suite, then the three bench lines (config 3, config 3 with inversions, config 5) and the kernel table +
# idle gaps of config 3:  gpurun --timeout 2400 -- 'bash scripts/r6_suite.sh'
O=gpurun_out/r6; mkdir -p $O
timeout 120 python scripts/smoke_core.py 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -q > $O/suite.log 2>&1; tail -6 $O/suite.log
bash scripts/r6_step.sh bench inv rearr
bash scripts/profile_stats.sh > $O/stats.log 2>&1; cp gpurun_out/prof_stats/summary/idle_gaps.json gpurun_out/prof_stats/summary/kernel_stats.csv $O/ 2>/dev/null
python - <<'PY'
import json
g = json.load(open('gpurun_out/r6/idle_gaps.json'))
print({k: v for k, v in g.items() if not isinstance(v, (list, dict))})
for row in (g.get("gaps") or [])[:25]: print(row)
PY
