#!/bin/bash
# Round-4 evidence beyond the headline workload (run on the GPU box from the repo root, via gpurun): bench lines of the other
# workloads, the streaming and latency harnesses, the full-size properties, one-shard-alone timings and the group rehearsal.
#   bash profiles/collect_r04_extra.sh part1|part2|part3|part4
set -o pipefail
OUT=gpurun_out/r04/final; mkdir -p $OUT
P=${1:-part1}
line() { grep '^{' | tail -1; }
if [ "$P" = part1 ]; then
  for w in c2 c3z c5s; do python3 bench.py --workload $w --steps 3 --warmup 1 --cpu-seconds 4 2> $OUT/bench_$w.err | line > $OUT/bench_$w.json; echo "$w rc=$?"; done
  for w in c3z1 c5z; do python3 bench.py --workload $w --steps 3 --warmup 1 --cpu-seconds 4 2> $OUT/bench_$w.err | line > $OUT/bench_$w.json; echo "$w rc=$?"
    APSS_DEBUG=head_bf16 python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-two-directional-row 2>/dev/null | line > $OUT/bench_${w}_bf16.json; echo "$w bf16 rc=$?"; done
  python3 bench_stream.py --refcpu-seconds 8 > $OUT/stream.json 2> $OUT/stream.err; echo "stream rc=$?"
  python3 bench_latency.py --shape production --messages 2000 --interval-ms 0 > $OUT/latency_production.json 2>/dev/null; echo "lat1 rc=$?"
  python3 bench_latency.py --shape c3 --messages 2000 --interval-ms 0 > $OUT/latency_c3.json 2>/dev/null; echo "lat2 rc=$?"
  python3 bench_latency.py --shape production --messages 50 --interval-ms 50 > $OUT/latency_production_cadence.json 2>/dev/null; echo "lat3 rc=$?"
  python3 bench_latency.py --shape template --theta 0 --messages 200 --interval-ms 0 > $OUT/latency_template_theta0.json 2>/dev/null; echo "lat4 rc=$?"
fi
if [ "$P" = part2 ]; then
  python3 profiles/fullsize_powerlaw.py > $OUT/c5z_full.json 2> $OUT/c5z_full.err; echo "c5z_full rc=$?"
  APSS_DEBUG=head_bf16 python3 profiles/fullsize_powerlaw.py > $OUT/c5z_full_bf16.json 2> $OUT/c5z_full_bf16.err; echo "c5z_full_bf16 rc=$?"
  python3 profiles/fullsize_stratified.py > $OUT/c5_full.json 2> $OUT/c5_full.err; echo "c5_full rc=$?"
  # one shard / one cell alone on this GPU (projections before the exchange)
  : > $OUT/solo.jsonl
  for s in 1,1,0,0 2,1,0,0 4,1,0,0 8,1,0,0 8,1,7,0 1,8,0,3 2,4,0,1 4,2,0,1 1,2,0,1 1,4,0,2; do python3 bench.py --solo $s --steps 3 2>/dev/null | line >> $OUT/solo.jsonl; done
  for s in 1,8,0,3 2,4,0,1; do python3 bench.py --solo $s --steps 3 --no-symmetric-ranges 2>/dev/null | line | sed 's/"solo_shard": "/"solo_shard": "nosym:/' >> $OUT/solo.jsonl; done
  for s in 8,1,0,0 8,1,7,0 2,4,0,1 1,8,0,3; do python3 bench.py --workload c5z --solo $s --steps 2 2>/dev/null | line | sed 's/"solo_shard": "/"solo_shard": "c5z:/' >> $OUT/solo.jsonl; done
  echo "solo rc=$?"
fi
if [ "$P" = part3 ]; then
  python3 bench.py --gpus 4 --engine group --share-device --steps 3 --warmup 1 --cpu-seconds 2 2> $OUT/group4.err | line > $OUT/group4_c3.json; echo "group4 rc=$?"
  python3 bench.py --gpus 2 --engine group --share-device --workload c3z1 --steps 2 --warmup 1 --cpu-seconds 2 2> $OUT/group2z.err | line > $OUT/group2_c3z1.json; echo "group2z rc=$?"
  python3 bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --cpu-seconds 2 2> $OUT/gloo2.err | line > $OUT/rehearsal_gloo2.json; echo "gloo2 rc=$?"
  bash profiles/collect_r04.sh c3z1 > $OUT/collect_c3z1.log 2>&1; echo "collect c3z1 rc=$?"
  bash profiles/head_counters_r04.sh c5z i8final > $OUT/headctr_i8.log 2>&1; echo "headctr rc=$?"
  bash profiles/head_counters_r04.sh c5z bf16 APSS_DEBUG=head_bf16 > $OUT/headctr_bf16.log 2>&1; echo "headctr bf16 rc=$?"
fi
if [ "$P" = part4 ]; then
  # after the merged rounds of the term shards (k_probe_even_merged): the T x 1 shards alone, the group rehearsals, and the headline
  # line once more now that profiles/r04_probe_traffic.json carries this source tree's hash (bench.py quotes the counters again)
  : > $OUT/solo_merge.jsonl
  for s in 4,1,0,0 8,1,0,0 8,1,7,0; do python3 bench.py --solo $s --steps 3 2>/dev/null | line >> $OUT/solo_merge.jsonl; done
  for s in 8,1,0,0 8,1,7,0; do python3 bench.py --workload c5z --solo $s --steps 2 2>/dev/null | line | sed 's/"solo_shard": "/"solo_shard": "c5z:/' >> $OUT/solo_merge.jsonl; done
  echo "solo rc=$?"
  python3 bench.py --gpus 4 --engine group --share-device --steps 3 --warmup 1 --cpu-seconds 2 2> $OUT/group4.err | line > $OUT/group4_c3.json; echo "group4 rc=$?"
  python3 bench.py --gpus 8 --engine group --share-device --steps 3 --warmup 1 --cpu-seconds 2 2> $OUT/group8.err | line > $OUT/group8_c3.json; echo "group8 rc=$?"
  python3 bench.py --steps 5 --warmup 1 2> $OUT/bench_c3_again.err | line > $OUT/bench_c3_again.json; echo "c3 rc=$?"
fi
