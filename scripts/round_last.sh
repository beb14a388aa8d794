#!/bin/bash
# the last call of a round: the whole GPU suite (bounded), then the profiler passes + the default bench line on the same build
mkdir -p gpurun_out/final
timeout 520 python -m pytest tests -m gpu -q -x --timeout 240 2>&1 | tail -4 > gpurun_out/final/suite.log; cat gpurun_out/final/suite.log
bash scripts/round_profile.sh
