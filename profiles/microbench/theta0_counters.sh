# SQ / memory counters of k_probe<2, ..> (theta <= 0) on the template shape; on the GPU box from the repo root
OUT=$PWD/gpurun_out/r04/theta0ctr; mkdir -p $OUT; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/profiles/theta0_r03.py > $OUT/p$i.log 2>&1 < /dev/null; echo "p$i rc=$?"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg, calls = collections.defaultdict(float), collections.defaultdict(int)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_probe<2" in r["Kernel_Name"] or "k_probe<(int)2" in r["Kernel_Name"] or ("k_probe<" in r["Kernel_Name"] and "k_probe_" not in r["Kernel_Name"]):
            agg[(r["Kernel_Name"][:40], r["Counter_Name"])] += float(r["Counter_Value"]); calls[(r["Kernel_Name"][:40], r["Counter_Name"])] += 1
for k in sorted(agg): print(k, agg[k] / calls[k], "per launch (", calls[k], ")")
PY
rm -rf $OUT/p*/
