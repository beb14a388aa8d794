#!/bin/bash
# scripts/sanitize_host.sh -- the host code (parsnp_amd/csrc/host) + the engine's orchestration over the kernel emulation
# (tests/emu/engine_emu.cpp) under AddressSanitizer + UBSan and under ThreadSanitizer, several steps per process
# (tests/emu/steps_main.cpp), on two small sets and every route of the layout / seed-region plumbing.  CPU only, ~5 minutes.
#   bash scripts/sanitize_host.sh [workdir]
# ThreadSanitizer runs with OMP_THREAD_LIMIT=1: libgomp's barriers are invisible to it (hundreds of false reports with the
# OpenMP teams on), while the helper threads of a step -- the batch computed ahead, the layout image request and its
# corrections, the put-off marks, the chaining verdicts -- are ordinary threads and are checked as they run.
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
W=${1:-/tmp/parsnp_sanitize}
mkdir -p $W
cd $REPO
HOST=$(ls parsnp_amd/csrc/host/*.cpp | grep -v "main.cpp\|capi.cpp")
FLAGS="-O1 -g -fno-omit-frame-pointer -mavx2 -std=c++17 -fopenmp -w -DPARSNP_TEST_HOOKS -DPM_WAVE_EVENTS=5"
g++ $FLAGS -fsanitize=address,undefined tests/emu/steps_main.cpp tests/emu/engine_emu.cpp $HOST -o $W/steps_asan &
g++ $FLAGS -fsanitize=thread tests/emu/steps_main.cpp tests/emu/engine_emu.cpp $HOST -o $W/steps_tsan &
wait
python - <<EOF
import os, sys
sys.path.insert(0, "$REPO")
from parsnp_amd import synth, driver
for name in ("pop6x200k", "rearr6x300k"):
    ref, gs = synth.make(name)
    d = "$W/%s" % name
    os.makedirs(d + "/in", exist_ok=True); os.makedirs(d + "/out", exist_ok=True)
    rp, qs = synth.write_set(d + "/in", ref, gs)
    open(d + "/run.ini", "w").write(driver.ini_text(rp, qs, d + "/out", threads=4))
EOF
bad=0
for name in pop6x200k rearr6x300k; do
    want=$(python -c "import json; print(json.load(open('$REPO/tests/golden/e2e.json'))['$name']['xmfa_md5'])")
    for route in "DEFAULT=1" "PM_FLAGGED_DIV=1" "PARSNP_NO_DEVICE_CHAIN=1 PARSNP_SPLIT_SETTLE=1 PARSNP_ONE_STAGE=1" "PM_STAGE_GATE=1 PM_CHAIN_TIE=1" "PM_CLUSTER_UNSURE=1 PARSNP_NO_DEVICE_CHAIN=1" "PARSNP_NO_RESIDENT=1" "PARSNP_NO_RESIDENT=1 PARSNP_SEQUENTIAL_REPLAY=1"; do
        for san in asan tsan; do
            cd $W/$name/out; rm -f parsnpAligner.xmfa
            extra=""; [ $san = tsan ] && extra="OMP_THREAD_LIMIT=1"
            env $route $extra PM_DIRTY_MIN=16 PARSNP_PARALLEL_MIN=16 PARSNP_CHECK_ZERO=1 ASAN_OPTIONS=detect_leaks=0 TSAN_OPTIONS=halt_on_error=0 \
                $W/steps_$san ../run.ini 3 > $W/log.txt 2>&1 || { echo "FAILED rc: $name $route $san"; bad=1; }
            n=$(grep -c "runtime error\|AddressSanitizer\|WARNING: ThreadSanitizer" $W/log.txt || true)
            got=$(md5sum parsnpAligner.xmfa | cut -d' ' -f1)
            echo "$name [$route] $san: reports=$n md5 $([ "$got" = "$want" ] && echo ok || echo DIFFERS)  $(grep steps_main $W/log.txt | cut -c13-)"
            [ "$n" != "0" ] && bad=1
            [ "$got" != "$want" ] && bad=1
        done
    done
done
exit $bad
