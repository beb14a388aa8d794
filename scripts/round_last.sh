#!/bin/bash
# the last call of a round: the whole GPU suite (bounded), then the profiler passes + the default bench line on the same build
mkdir -p gpurun_out/final
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error" > gpurun_out/final/suite.log; cat gpurun_out/final/suite.log      # (no per-test --timeout: the 2 000-genome flow alone takes minutes)
bash scripts/round_profile.sh
