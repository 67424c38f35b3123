#!/bin/bash
# per-shard step time of every rank grid T x D of the multi-GPU layouts, each shard alone on ONE MI355X (bench.py --solo):
# the slowest shard of a grid bounds the grid's step time (plus the exchange for T > 1).  C3, 3 steps each.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02/solo
for g in 1,1,0,0 2,1,0,0 2,1,1,0 4,1,0,0 4,1,3,0 8,1,0,0 8,1,3,0 8,1,7,0 1,2,0,0 1,4,0,1 1,8,0,3 2,2,1,1 2,4,0,2 4,2,1,1; do
  timeout -k 10 200 python3 bench.py --solo $g --steps 3 --no-cpu-baseline > gpurun_out/r02/solo/$g.json 2> gpurun_out/r02/solo/$g.err || exit 1
done
