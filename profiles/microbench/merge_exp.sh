#!/bin/bash
# merged rounds of a term shard's thin-round filter (APSS_DEBUG=merge=L: at most 2^L query rows share a round): correctness first,
# then the T x 1 shards of C3 alone, per setting
OUT=gpurun_out/r04/merge
mkdir -p $OUT
cd "$(dirname "$0")/../.."
timeout -k 10 500 python3 -m pytest tests/test_gpu_merged_rounds.py tests/test_gpu_shards.py tests/test_gpu_even.py -x -q -m gpu > $OUT/tests.txt 2>&1
rc=$?
tail -n 3 $OUT/tests.txt
if [ $rc -ne 0 ]; then echo "tests rc=$rc"; exit $rc; fi
: > $OUT/solo.jsonl
i=0
for dbg in merge=0 merge=1 merge=2,merge_u=7 merge=1,merge_single=4,merge_u=7 merge=2,merge_single=4,merge_u=7; do
  for s in 8,1,0,0 4,1,0,0; do
    APSS_DEBUG=$dbg timeout -k 10 200 python3 bench.py --solo $s --steps 3 2>$OUT/err_$i.txt | grep '^{' | sed "s/^{/{\"merge\": \"$dbg\", /" >> $OUT/solo.jsonl || exit 1
  done
  i=$((i+1))
done
python3 - <<'P'
import json
for l in open("gpurun_out/r04/merge/solo.jsonl"):
    d = json.loads(l)
    print(d.get("merge"), d.get("solo_shard"), "ms", round(d.get("ms_per_step", 0), 2), "kernel", d.get("probe_kernel"), "probe_ms", d.get("probe_kernel_ms"), "qpr", d.get("queries_per_round"), "cands", d.get("candidates"), "alg", d.get("algorithmic_frac"), "surv", d.get("filter_survivors"))
P
