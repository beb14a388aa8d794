timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_events or random_regions or mers_anchor or batched or small_regions" 2>&1 | tail -2
bash scripts/r6_step.sh bench
