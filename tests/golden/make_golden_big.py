"""BASELINE-size goldens from the REFERENCE binary (oracle/_ref/parsnp_core_ref, built from /root/reference by
oracle/Makefile).  Build container only; each run is tens of minutes to an hour and several GB:

    python tests/golden/make_golden_big.py bact200 bact2000_p0 rearr50 [--work /tmp/big] [--cores 2]

  bact200      BASELINE config 3: 200 x 5 Mb, population model seed 5, --no-partition
  bact2000_p0  BASELINE config 4: partition 0 (250 genomes, Random(42) order) of 2000 x 5 Mb, seed 6
  rearr50      BASELINE config 5 cut to its first 50 genomes: 5 % segregating sites, 10 % of every genome rearranged
  rearr500     BASELINE config 5 in full (500 genomes; 67 minutes of the reference binary on 3 cores, a 2.47 GB XMFA)

Writes xmfa md5, MUM/LCB signature, log counters and the reference's own phase timers into tests/golden/e2e_big.json
(merged with what is there).  The inputs are regenerated from the same seeds by tests/test_gpu_big.py on the GPU box."""
import json
import os
import re
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import xmfa_util  # noqa: E402
from parsnp_amd import driver, synth  # noqa: E402

REFBIN = os.path.join(ROOT, "oracle", "_ref", "parsnp_core_ref")


def inputs(name, work):
    """-> (ref path, [query paths]); same bytes and file names as tests/test_gpu_big.py builds"""
    d = os.path.join(work, name, "in")
    if name == "bact2000_p0":
        ref, gs, ids = synth.make_partition(0)
        return synth.write_set(d, ref, gs, ids)
    ref, gs = synth.make(name)
    return synth.write_set(d, ref, gs)


def main():
    args = sys.argv[1:]
    work, cores = "/tmp/big", 2
    if "--work" in args:
        i = args.index("--work"); work = args[i + 1]; del args[i:i + 2]
    if "--cores" in args:
        i = args.index("--cores"); cores = int(args[i + 1]); del args[i:i + 2]
    path = os.path.join(HERE, "e2e_big.json")
    for name in args:
        t0 = time.time()
        rp, qs = inputs(name, work)
        t1 = time.time()
        out = os.path.join(work, name, "ref_out")
        rc, _ = driver.run_core(REFBIN, rp, qs, out, threads=cores)
        assert rc == 0, (name, rc)
        x = os.path.join(out, "parsnpAligner.xmfa")
        log = os.path.join(out, "parsnpAligner.log")
        timers = dict(re.findall(r"^\s*([A-Za-z\- ]+?) elapsed time:\s+([0-9.]+)s", open(log).read(), re.M))
        entry = dict(xmfa_md5=xmfa_util.md5(x), signature=xmfa_util.mum_lcb_signature(x), log=xmfa_util.log_counters(log),
                     ref_records=[h for h, _ in xmfa_util.records(x)[1] if h.startswith("> 1:")][:50],
                     n_queries=len(qs), xmfa_bytes=os.path.getsize(x),
                     reference_timers_s={k.strip(): float(v) for k, v in timers.items()},
                     reference_wall_s=round(time.time() - t1, 1), generate_s=round(t1 - t0, 1), reference_cores=cores)
        big = json.load(open(path)) if os.path.exists(path) else {}     # re-read: several of these run side by side
        big[name] = entry
        json.dump(big, open(path, "w"), indent=1)
        print(name, entry["xmfa_md5"], entry["reference_wall_s"], flush=True)


if __name__ == "__main__":
    main()
