#!/bin/bash
# the driver's round-end sequence on the final tree: smoke(), the GPU suite (summary line kept)
mkdir -p gpurun_out/final gpurun_out/profiles_r05
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/final/suite_last.log 2>&1; grep -E "passed|failed|error" gpurun_out/final/suite_last.log | tail -3 | tee gpurun_out/profiles_r05/gpu_suite.txt
