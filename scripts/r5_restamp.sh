#!/bin/bash
# after a change of the engine sources: the profiler passes and the default bench line again (their summaries are stamped with the
# sources they were measured on), the other workloads' chatter, and the resident-route parity tests as a guard
mkdir -p gpurun_out/final gpurun_out/profiles_r05 gpurun_out/r5
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_vs_reference.py -m gpu -x -q -k "resident_route or order_of_reads or another_order or twins" 2>&1 | grep -E "passed|failed" | tail -2
bash scripts/round_profile.sh 2>&1 | tail -9
bash scripts/r5_why.sh 2>&1 | tail -6
