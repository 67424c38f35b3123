#!/bin/bash
# SQ counters of the merged thin-round filter (k_probe_even_merged) on one T = 8 term shard of C3 alone, and of the unmerged
# kernel (APSS_DEBUG=merge=0) beside it: why 0.60 and not more.  One rocprofv3 --pmc pass per group (never with a trace).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04/shard8ctr
mkdir -p $OUT
for mode in merged unmerged; do
  if [ $mode = unmerged ]; then export APSS_DEBUG=merge=0; else unset APSS_DEBUG; fi
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
    tag=$(echo $grp | cut -d' ' -f1)
    rocprofv3 --pmc $grp --output-format csv -d $OUT/${mode}_$tag -- python3 bench.py --solo 8,1,0,0 --steps 1 > $OUT/${mode}_$tag.log 2>&1 || { echo "$mode $tag failed"; exit 1; }
  done
done
unset APSS_DEBUG
python3 - <<'P'
import csv, glob, json, collections
out = {}
for mode in ("merged", "unmerged"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(int)
    for f in glob.glob("gpurun_out/r04/shard8ctr/%s_*/**/*counter_collection.csv" % mode, recursive=True):
        seen = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_probe_even" not in k:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            seen[k].add(r["Dispatch_Id"])
        for k in seen:
            calls[(k, f)] = len(seen[k])
    for k, c in agg.items():
        n = max(v for (kk, f), v in calls.items() if kk == k)
        e = {c_: v / n for c_, v in c.items()}
        e["launches_per_pass"] = n
        if e.get("SQ_BUSY_CYCLES"):
            # per-SE counters: SQ_BUSY_CYCLES sums over the shader engines; fractions of wave-cycles and of busy cycles as profiles/summarize_r04.py reads them
            e["valu_busy_frac_of_wave_cycles"] = e.get("SQ_ACTIVE_INST_VALU", 0) / max(1.0, e.get("SQ_WAVE_CYCLES", 1))
            e["lds_busy_frac_of_wave_cycles"] = e.get("SQ_ACTIVE_INST_LDS", 0) / max(1.0, e.get("SQ_WAVE_CYCLES", 1))
            e["waiting_frac_of_wave_cycles"] = e.get("SQ_WAIT_INST_ANY", 0) / max(1.0, e.get("SQ_WAVE_CYCLES", 1))
        out["%s: %s" % (mode, k[:70])] = e
json.dump(out, open("gpurun_out/r04/shard8ctr/summary.json", "w"), indent=1)
for k, e in out.items():
    print(k, {c: (round(v, 4) if v < 10 else int(v)) for c, v in e.items()})
P
