#!/bin/bash
# round 4: the whole GPU suite, the host-thread sweep (= the per-rank curve of a node under a CPU quota) and the 8-rank shape on one GPU
mkdir -p gpurun_out
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/r4_suite.log; tail -4 $O/r4_suite.log
echo "[" > $O/r4_host_threads.json
for t in 2 4 6 8 12 16 24; do
  timeout 200 python bench.py --steps 40 --warmup 5 --cpu-sample 0 --host-threads $t 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'host_threads': $t, 'ms_per_step': d['ms_per_step'], 'genomes_per_s': d['value'], 'host_cores_busy': d['host_cores_busy'], 'host_cpus_usable': d['config']['host_cpus_usable'], 'pcie_bytes_per_step': d['pcie_bytes_per_step'], 'resident_route': d['resident_route']}) + ',')" | tee -a $O/r4_host_threads.json
done
echo "null]" >> $O/r4_host_threads.json
# 8 ranks sharing the one GPU (gloo): partition mode, a partition of 200 x 5 Mb per rank
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 8 --mode partition --steps 10 --warmup 2 --cpu-sample 0 > $O/r4_8ranks_partition.json 2> $O/r4_8ranks_partition.err
tail -1 $O/r4_8ranks_partition.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('8 ranks on one GPU, partition mode:', d['value'], 'genomes/s', d['ms_per_step'], 'ms/step'); print(d['per_rank'])" || tail -5 $O/r4_8ranks_partition.err
free -g | head -2; df -h /dev/shm | tail -1
