bash scripts/r6_step.sh tests bench
timeout 600 python -m pytest tests/test_gpu_big.py -m gpu -x -q -k "baseline_size" 2>&1 | tail -2
