"""One-off probe of the full-size configurations on the GPU box: environment, wall time of generating and running
BASELINE configs 4 (8 partitions of 250) and 5 (500 genomes, rearranged).  Prints one line per step; measurement helper."""
import json, os, shutil, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parsnp_amd import driver, synth
from parsnp_amd.paths import CORE_BIN

def sh(c):
    return subprocess.run(c, shell=True, capture_output=True, text=True).stdout.strip()

print("nproc", sh("nproc"), "| mem", sh("free -g | sed -n 2p"), flush=True)
print(sh("df -h /dev/shm /tmp | cat"), flush=True)
print("cpu quota", sh("cat /sys/fs/cgroup/cpu.max 2>/dev/null"), flush=True)
base = "/dev/shm/probe" if shutil.disk_usage("/dev/shm").free > (40 << 30) else "/tmp/probe"
os.makedirs(base, exist_ok=True)
what = sys.argv[1:] or ["rearr500", "bact2000"]
if "rearr500" in what:
    t = time.time(); ref, gs = synth.make("rearr500"); rp, qs = synth.write_set(base + "/r500/in", ref, gs); print("rearr500 generate+write %.1f s" % (time.time() - t), flush=True)
    t = time.time(); rc, _ = driver.run_core(CORE_BIN, rp, qs, base + "/r500/out", threads=24, env=dict(os.environ, PARSNP_DEBUG_TIMERS="1")); print("rearr500 run rc=%d %.1f s" % (rc, time.time() - t), flush=True)
    print(sh("tail -c 3000 %s/r500/out/parsnp-aligner.err" % base)); print(sh("grep -E 'MUM|luster|overage' %s/r500/out/parsnpAligner.log | head -20; ls -la %s/r500/out" % (base, base)), flush=True)
    shutil.rmtree(base + "/r500", ignore_errors=True)
if "bact2000" in what:
    for part in range(8):
        t = time.time(); ref, gs, ids = synth.make_partition(part); rp, qs = synth.write_set(base + "/b2000/in%d" % part, ref, gs, ids); t1 = time.time()
        rc, _ = driver.run_core(CORE_BIN, rp, qs, base + "/b2000/out%d" % part, threads=24); t2 = time.time()
        print("bact2000 partition %d: generate %.1f s, run rc=%d %.2f s, xmfa %d MB" % (part, t1 - t, rc, t2 - t1, os.path.getsize(base + "/b2000/out%d/parsnpAligner.xmfa" % part) >> 20), flush=True)
    shutil.rmtree(base + "/b2000", ignore_errors=True)
