#!/bin/bash
# Counters of the dense-head contraction inside the library (bench.py --workload $1, default c5z = power-law C5 at N = 2M):
# matrix-pipe busy share, VALU per MFMA, L2 hit rate and fabric traffic of k_head_gemm.  Own passes, kernel-trace only.
# usage (on the GPU box, from the repo root): bash profiles/head_counters_r04.sh [workload] [tag] [ENV=1 ...]
set -o pipefail
WL=${1:-c5z}; TAG=${2:-i8}; shift; shift
for kv in "$@"; do export "$kv"; done
OUT=$PWD/gpurun_out/r04/headctr_${WL}_${TAG}; mkdir -p $OUT
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-two-directional-row > $OUT/p$i.log 2>&1 < /dev/null; echo "p$i rc=$?"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
agg, calls = collections.defaultdict(float), collections.defaultdict(int)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_head_gemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            calls[r["Counter_Name"]] += 1
res = {k: {"sum_over_launches": agg[k], "launches": calls[k]} for k in sorted(agg)}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for k in sorted(agg):
    print(k, agg[k], "(", calls[k], "launches )")
PY
rm -rf $OUT/p*/
