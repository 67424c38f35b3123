#!/bin/bash
# Round-3 evidence beside collect_r03.sh: the skewed workloads' bench lines, one term shard alone (uniform C3 and power-law
# C5, with rocprofv3 kernel stats), the latency harness.  Run on the GPU box from the repo root.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03/extra
mkdir -p $OUT
python3 bench.py --workload c5z --steps 2 --warmup 1 --no-cpu-baseline --no-exact-row 2> /dev/null | grep '^{' > $OUT/bench_c5z.json; echo "c5z rc=$?"
for s in 2,1,0,0 4,1,0,0 8,1,0,0 8,1,7,0; do python3 bench.py --solo $s --steps 3 --no-cpu-baseline --head-terms -1 2> /dev/null | grep '^{'; done > $OUT/solo_c3.jsonl; echo "solo c3 rc=$?"
for s in 4,1,0,0 8,1,0,0 8,1,7,0; do python3 bench.py --workload c5z --solo $s --steps 2 --no-cpu-baseline 2> /dev/null | grep '^{'; done > $OUT/solo_c5z.jsonl; echo "solo c5z rc=$?"
python3 bench.py --workload c3z1 --solo 2,1,0,0 --steps 2 --no-cpu-baseline 2> /dev/null | grep '^{' > $OUT/solo_c3z1.jsonl
python3 bench.py --workload c3z1 --solo 8,1,0,0 --steps 2 --no-cpu-baseline 2> /dev/null | grep '^{' >> $OUT/solo_c3z1.jsonl; echo "solo c3z1 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_shard8 -- python3 bench.py --solo 8,1,3,0 --steps 5 --no-cpu-baseline --head-terms -1 > $OUT/trace_shard8.log 2>&1; echo "trace shard8 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5z_shard8 -- python3 bench.py --workload c5z --solo 8,1,3,0 --steps 3 --no-cpu-baseline > $OUT/trace_c5z_shard8.log 2>&1; echo "trace c5z shard8 rc=$?"
python3 bench_latency.py --preload 200000 --messages 50 > $OUT/latency.json 2> /dev/null; echo "latency rc=$?"
python3 bench_latency.py --preload 200000 --messages 50 --zipf 1.0 --backend gpu > $OUT/latency_zipf.json 2> /dev/null
python3 bench_latency.py --shape template --theta 0 --preload 100000 --messages 50 --interval-ms 10 > $OUT/latency_theta0.json 2> /dev/null; echo "latency theta0 rc=$?"
