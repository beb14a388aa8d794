python scripts/why_route.py rearr50 2>&1 | grep "^rc"
timeout 1500 python -m pytest tests/test_gpu_big.py -m gpu -x -q -k "baseline_size" 2>&1 | tail -5
