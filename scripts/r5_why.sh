#!/bin/bash
# round 5: the host's chatter of three bench runs (bench.py keeps it when PARSNP_BENCH_LOG names a file): why a step left the
# resident route (bact200inv), the host route's laps on config 5, the order check's counts on the headline workload
mkdir -p gpurun_out/r5 gpurun_out/profiles_r05
O=gpurun_out/r5
B="--steps 2 --warmup 1 --cpu-sample 0 --other-configs off"
PARSNP_BENCH_LOG=$O/inv.log PARSNP_DEBUG_TIMERS=1 timeout 300 python bench.py --workload bact200inv $B > /dev/null 2> $O/inv.err
grep -E "route left|order check|\[extend\]" $O/inv.log | sort | uniq -c | head
PARSNP_BENCH_LOG=$O/rearr.log PARSNP_DEBUG_TIMERS=1 timeout 400 python bench.py --workload rearr500 $B > $O/rearr.json 2> $O/rearr.err
grep -E "^\[(anchors|extend|lcb|replay|sweep|filter|validate_parallel|chain)" $O/rearr.log | tail -60 > gpurun_out/profiles_r05/rearr500_laps.txt; tail -1 $O/rearr.json | python scripts/benchline.py | head -1
PARSNP_BENCH_LOG=$O/order.log timeout 200 python bench.py $B --tune order_debug=1 > /dev/null 2> $O/order.err
grep "order check" $O/order.log | tail -1 | tee gpurun_out/profiles_r05/order_check_counts.txt
