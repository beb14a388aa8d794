#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
bash scripts/profile_stats.sh > $O/r4_stats.log 2>&1; grep -E "FoldCandidates|ChunkScan|ChunkReduce|GroupReduce|MasterEP|Grouped|SmallPair" $O/prof_stats/summary/kernel_stats.csv | cut -c1-200
