#!/bin/bash
# kernel trace of one T = 8 term shard of C3 alone (merged rounds: k_prenorm_rows, k_probe_even_merged, k_expand_merged, k_shard_prune)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04/shard8
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --solo 8,1,0,0 --steps 3 > $OUT/trace.log 2>&1; echo "trace rc=$?"
f=$(ls $OUT/trace/*/*kernel_stats.csv | tail -1)
[ -n "$f" ] && cut -c1-160 "$f" | sed -n 1,25p
