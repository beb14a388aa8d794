import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d["split_s"]; h=d["host_split_s"]
print(d["value"], d["ms_per_step"], d["step_ms"], "setup %.1f anchor %.1f extend %.1f lcb %.1f | validate %.1f neighbour %.1f wall %.1f" % (s["setup"]*1e3, s["anchor"]*1e3, s["extend"]*1e3, s["lcb"]*1e3, h["validate"]*1e3, h["neighbour"]*1e3, s["engine_calls_wall"]*1e3))
