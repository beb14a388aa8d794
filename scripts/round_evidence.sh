#!/bin/bash
# round 4, the evidence behind profiles/r04: the whole GPU suite, the default bench line, the rocprofv3 passes of scripts/profile.sh,
# the SQ counters of the search kernels, the host-thread sweep (the per-rank curve of a node under a CPU quota) and 8 ranks on ONE GPU
mkdir -p gpurun_out/final
O=gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/suite.log; tail -3 $O/suite.log
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json | python scripts/benchline.py | head -2
timeout 300 python bench.py --steps 60 --warmup 5 --cpu-sample 0 --other-configs off --tune group_small=0 > $O/bench_group_small_0.json 2> /dev/null; tail -1 $O/bench_group_small_0.json | python scripts/benchline.py | head -1
bash scripts/profile.sh > $O/profile.log 2>&1; ls gpurun_out/prof/summary
bash scripts/sqcounters.sh > $O/sq.log 2>&1; tail -12 $O/sq.log | cut -c1-300
echo "[" > $O/host_threads.json
for t in 0 2 4 8 16 24; do
  timeout 200 python bench.py --steps 40 --warmup 5 --cpu-sample 0 --other-configs off --host-threads $t 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'host_threads_asked': $t, 'host_threads': d['config']['host_threads'], 'ms_per_step': d['ms_per_step'], 'genomes_per_s': d['value'], 'host_cores_busy': d['host_cores_busy'], 'host_cpus_usable': d['config']['host_cpus_usable'], 'pcie_bytes_per_step': d['pcie_bytes_per_step'], 'resident_route': d['resident_route']}) + ',')" | tee -a $O/host_threads.json
done
echo "null]" >> $O/host_threads.json
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 8 --mode partition --steps 10 --warmup 2 --cpu-sample 0 --other-configs off > $O/eight_ranks_one_gpu.out 2> $O/eight_ranks_one_gpu.err
tail -1 $O/eight_ranks_one_gpu.out > $O/eight_ranks_one_gpu.json
python -c "
import json
d=json.loads(open('$O/eight_ranks_one_gpu.json').read()); print('8 ranks on one GPU, partition mode:', d['value'], 'genomes/s', d['ms_per_step'], 'ms/step'); print(d['per_rank'])" || tail -5 $O/eight_ranks_one_gpu.err
