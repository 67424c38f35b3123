#!/bin/bash
# LDS counters of the filter kernel on C3 with and without the bank-aware posting order (APSS_DEBUG=bank_order):
# SQ_LDS_BANK_CONFLICT, SQ_ACTIVE_INST_LDS, SQ_INSTS_LDS per launch.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02/bank; mkdir -p $OUT
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-exact-row"
for v in base bank_order; do
  if [ $v = bank_order ]; then export APSS_DEBUG=bank_order; else unset APSS_DEBUG; fi
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/$v -- $B > $OUT/$v.log 2>&1; echo "$v rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
for v in ("base", "bank_order"):
    agg = collections.defaultdict(float)
    for f in glob.glob("gpurun_out/r02/bank/%s/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_probe_coarse" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
    print(v, dict(agg), "conflict share of LDS-active cycles: %.3f" % (agg["SQ_LDS_BANK_CONFLICT"] / (4 * agg["SQ_ACTIVE_INST_LDS"])))
PY
