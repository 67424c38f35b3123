#!/bin/bash
# SQ counters of the term-shard filter kernel (shard 3 of T = 8, C3) for k_probe_even and (APSS_DEBUG=no_even) k_probe_coarse:
# where the cycles of a thin round go.  One --pmc pass per counter group (no trace options alongside).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02/shardctr; mkdir -p $OUT
B="python3 bench.py --solo 8,1,3,0 --steps 1 --warmup 0 --no-cpu-baseline"
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM"
G2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"
G3="SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
G4="SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAVE32_INSTS"
for v in even no_even; do
  if [ $v = no_even ]; then export APSS_DEBUG=no_even; else unset APSS_DEBUG; fi
  i=0
  for G in "$G1" "$G2" "$G3" "$G4"; do
    i=$((i+1))
    rocprofv3 --pmc $G --output-format csv -d $OUT/$v/g$i -- $B > $OUT/$v.g$i.log 2>&1; echo "$v g$i rc=$?"
  done
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for v in ("even", "no_even"):
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob("gpurun_out/r02/shardctr/%s/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_probe_even" in r["Kernel_Name"] or "k_probe_coarse" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    out[v] = {k: agg[k] / max(1, n[k]) for k in agg}
    out[v]["launches_seen"] = dict(n)
print(json.dumps(out, indent=1))
PY
