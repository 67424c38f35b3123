#!/bin/bash
# SQ counters of the dense-head contraction on the stand-alone microbenchmark (profiles/microbench/head_gemm_bench: random
# rows, N = 262,144, stored batch): where the matrix pipe's idle third goes.  Own passes, kernel-trace only.
# usage: profiles/gemm_counters.sh <outdir> [ENV=1 ...]   (e.g. NOCOUNT=1, M16=1)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$1; shift
for kv in "$@"; do export "$kv"; done
mkdir -p $OUT
B=profiles/microbench/head_gemm_bench
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC" \
           "SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- $B 262144 256 1 2 > $OUT/p$i.log 2>&1; echo "p$i rc=$?"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg, calls = collections.defaultdict(float), collections.defaultdict(int)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_head_gemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            calls[r["Counter_Name"]] += 1
for k in sorted(agg):
    print(k, agg[k] / calls[k], "per launch (", calls[k], "launches )")
PY
