timeout 2000 python -m pytest tests/test_gpu_big.py -m gpu -x -q -k "config5_full" 2>&1 | tail -5
