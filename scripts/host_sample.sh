#!/bin/bash
mkdir -p gpurun_out
gcc -O2 -shared -fPIC scripts/sigprof.c -o /tmp/sigprof.so
LD_PRELOAD=/tmp/sigprof.so SIGPROF_OUT=gpurun_out/sigprof timeout 300 python bench.py --steps 400 --warmup 5 --cpu-sample 0 --other-configs off > gpurun_out/r4_sigprof.json 2> gpurun_out/r4_sigprof.err
tail -1 gpurun_out/r4_sigprof.json | python scripts/benchline.py | head -1
for f in gpurun_out/sigprof.*; do python scripts/sigprof_report.py $f 2>/dev/null | head -60; done
