python scripts/why_route.py rearr50 2>&1 | tail -30
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_events or random_regions or mers_anchor or small_regions" 2>&1 | tail -3
bash scripts/r6_step.sh bench
