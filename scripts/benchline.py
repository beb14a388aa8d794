"""one-line digest of a bench.py JSON line on stdin: python bench.py ... | python scripts/benchline.py"""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d["split_s"]; h = d["host_split_s"]; e = d["engine_ms"]; a = d["anchor_launch_ms"]
import statistics
print(d["value"], d["ms_per_step"], "median %.2f min %.2f" % (statistics.median(d["step_ms"]), min(d["step_ms"])), d["step_ms"] if len(d["step_ms"]) <= 10 else "", "setup %.1f anchor %.1f extend %.1f lcb %.1f | validate %.1f neighbour %.1f wall %.1f | master_ep %.2f (anchor call %.2f) seed %.2f fold %.2f" % (
    s["setup"] * 1e3, s["anchor"] * 1e3, s["extend"] * 1e3, s["lcb"] * 1e3, h["validate"] * 1e3, h["neighbour"] * 1e3, s["engine_calls_wall"] * 1e3,
    e["master_ep"], a["master_ep"], e["seed_extend"], e["fold"]))
print("host_cores_busy", d.get("host_cores_busy"), "pcie/step", d.get("pcie_bytes_per_step"), "resident", d.get("resident_route"))
print("engine_ms", {k: v for k, v in sorted(e.items(), key=lambda kv: -kv[1])})
