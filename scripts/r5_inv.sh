#!/bin/bash
# round 5: the exact disjointness test of a generation's clusters (ClustersCollide) on the GPU box -- its tests, the resident-route
# parity tests, the inverted 200 x 5 Mb population (consistency test + bench line + why a step left the route, if one did)
mkdir -p gpurun_out/r5
O=gpurun_out/r5
timeout 600 python -m pytest tests/test_fuzz_vs_reference.py -m gpu -x -q -k "another_order or order_of_reads or inversions or resident_route_on_gpu" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "resident_route or twins or bact200inv" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_big.py -m gpu -x -q -k "bact200" 2>&1 | tail -3
PARSNP_BENCH_LOG=$O/inv.log PARSNP_DEBUG_TIMERS=1 timeout 300 python bench.py --workload bact200inv --steps 10 --warmup 2 --cpu-sample 0 --other-configs off > $O/bench_inv.json 2> $O/inv.err
grep -E "route left|\[extend\]" $O/inv.log | sort | uniq -c | head -5; tail -1 $O/bench_inv.json | python scripts/benchline.py | head -3
timeout 300 python bench.py --steps 60 --warmup 5 --cpu-sample 0 --other-configs off > $O/bench_after_inv.json 2> /dev/null; tail -1 $O/bench_after_inv.json | python scripts/benchline.py | head -1
