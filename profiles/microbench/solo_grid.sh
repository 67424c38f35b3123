#!/bin/bash
# per-shard step time of the T x D rank grids, each shard alone on one GPU (projection of a multi-GPU run)
mkdir -p gpurun_out/r01L
for g in ${@:-1,1,0,0 2,1,0,0 4,1,1,0 8,1,3,0 2,2,0,1 2,4,1,2 1,2,0,1 1,4,0,1 1,8,0,3}; do
  timeout -k 10 120 python bench.py --solo $g --steps 3 --warmup 1 > gpurun_out/r01L/solo_$g.log 2>&1 || exit 1
  tail -1 gpurun_out/r01L/solo_$g.log
done
