#!/bin/bash
# One shard of a term-sharded join WITH the dense-head block, alone on one MI355X (projection of the multi-GPU layout the
# one-GPU box cannot host): bench.py --solo T,D,i,j decides the block like rank 0 would (hip_head_chooser) and gives shard
# i its tail range + its share of the block's candidate tiles.  Usage: profiles/solo_head_r03.sh <workload> <out.jsonl> T...
set -e
wl=$1; out=$2; shift 2
: > "$out"
for T in "$@"; do
  for i in 0 $((T-1)); do
    python bench.py --workload "$wl" --solo "$T,1,$i,0" --steps 2 --no-cpu-baseline | grep '^{' >> "$out"
  done
done
cat "$out"
