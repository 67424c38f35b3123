#!/bin/bash
# SQ instruction-mix counters of the probe kernel (own passes, kernel-trace only): usage  profiles/sq_counters.sh <outdir> [ENV=1 ...]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$1; shift
for kv in "$@"; do export "$kv"; done
mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/p1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/p1.log 2>&1; echo "p1 rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/p2.log 2>&1; echo "p2 rc=$?"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH --output-format csv -d $OUT/p3 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/p3.log 2>&1; echo "p3 rc=$?"
rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p4 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/p4.log 2>&1; echo "p4 rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(float)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_probe" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:60], r["Counter_Name"])] += float(r["Counter_Value"])
for k in sorted(agg):
    print(k[0], k[1], agg[k])
PY
