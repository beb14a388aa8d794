#!/bin/bash
# scripts/floor.sh -- the per-step floor of one rank: bench.py --genomes 25 / 50 / 100 / 200 on the same 5 Mb population (ms per
# step, device phases, host time outside kernels), so that what N ranks can divide and what every rank repeats are numbers
# (DESIGN.md section 5).  Writes gpurun_out/floor.json (copied into profiles/rNN/).   gpurun --timeout 600 -- 'bash scripts/floor.sh'
mkdir -p gpurun_out
python - <<'PY'
import json, subprocess, sys
out = {"note": "bench.py --workload bact200 --genomes G --steps 60 --warmup 5 (one rank, one GPU): what a rank of a partition run with G genomes per GPU pays per step", "runs": []}
for G in (25, 50, 100, 200):
    p = subprocess.run([sys.executable, "bench.py", "--genomes", str(G), "--steps", "60", "--warmup", "5", "--cpu-sample", "0", "--other-configs", "off"], capture_output=True, text=True, timeout=400)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode or not lines:
        out["runs"].append({"genomes": G, "error": p.stderr[-300:]}); continue
    d = json.loads(lines[-1])
    out["runs"].append({"genomes": G, "ms_per_step": d["ms_per_step"], "genomes_per_s": d["value"], "device_ms_per_step": d.get("device_ms_per_step"), "host_ms_outside_kernels": d.get("host_ms_outside_kernels"),
                        "engine_ms": d["engine_ms"], "anchors": d["anchors"], "mums": d["mums"], "lcbs": d["lcbs"], "regions": d["regions"]["processed"], "host_cores_busy": d["host_cores_busy"]})
    print(G, d["ms_per_step"], d.get("device_ms_per_step"), d.get("host_ms_outside_kernels"), flush=True)
json.dump(out, open("gpurun_out/floor.json", "w"), indent=1)
PY
