set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --steps 3 --warmup 1 > gpurun_out/bench_gap.json 2> gpurun_out/bench_gap.err; tail -2 gpurun_out/bench_gap.json
python - <<'PY'
import os, time, json, subprocess
from parsnp_amd import synth, driver
r, gs = synth.make("bact200")
rp, qs = synth.write_set("/tmp/b200", r, gs)
env = dict(os.environ, PARSNP_TIMING="/tmp/b200/timing.json")
t = time.time()
rc, _ = driver.run_core("parsnp_amd/bin/parsnp_core", rp, qs, "/tmp/b200/out", env=env, cores=16)
print("rc", rc, "wall %.1fs" % (time.time() - t))
print(open("/tmp/b200/timing.json").read())
print(open("/tmp/b200/out/parsnpAligner.log").read()[-600:])
PY
