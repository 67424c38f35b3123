#!/bin/bash
# merged rounds of a term shard's thin-round filter (APSS_DEBUG=merge=L: at most 2^L query rows share a round; no_merge_prune: without
# the exact shard-rule test behind the expansion): correctness first, then the T x 1 shards alone, per setting
OUT=gpurun_out/r04/merge
mkdir -p $OUT
cd "$(dirname "$0")/../.."
timeout -k 10 600 python3 -m pytest tests/test_gpu_merged_rounds.py tests/test_gpu_shards.py tests/test_gpu_even.py tests/test_gpu_shard_head.py tests/test_gpu_group.py tests/test_gpu_symmetric.py -x -q -m gpu > $OUT/tests.txt 2>&1
rc=$?
tail -n 3 $OUT/tests.txt
if [ $rc -ne 0 ]; then echo "tests rc=$rc"; exit $rc; fi
: > $OUT/solo2.jsonl
for dbg in merge=0 no_merge_prune merge=1; do
  for s in 8,1,0,0 4,1,0,0; do
    APSS_DEBUG=$dbg timeout -k 10 200 python3 bench.py --solo $s --steps 3 2>$OUT/err.txt | grep '^{' | sed "s/^{/{\"merge\": \"$dbg\", /" >> $OUT/solo2.jsonl || exit 1
  done
done
for dbg in no_merge_prune merge=1; do
  APSS_DEBUG=$dbg timeout -k 10 200 python3 bench.py --workload c5z --solo 8,1,0,0 --steps 2 2>$OUT/err.txt | grep '^{' | sed "s/^{/{\"merge\": \"$dbg\", \"workload\": \"c5z\", /" >> $OUT/solo2.jsonl || exit 1
done
python3 - <<'P'
import json
for l in open("gpurun_out/r04/merge/solo2.jsonl"):
    d = json.loads(l)
    print(d.get("merge"), d.get("workload", "c3"), d.get("solo_shard"), "ms", round(d.get("ms_per_step", 0), 2), "kernel", d.get("probe_kernel"), "probe_ms", round(d.get("probe_kernel_ms"), 2), "qpr", d.get("queries_per_round"), "cands", d.get("candidates"), "alg", round(d.get("algorithmic_frac"), 3))
P
