for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 23 24; do python scripts/why_route.py rearr50 2>&1 | grep -E "^rc" | cut -c1-120; done | sort | uniq -c
bash scripts/r6_step.sh tests inv
timeout 600 python -m pytest tests/test_gpu_big.py -m gpu -x -q -k "baseline_size" 2>&1 | tail -2
