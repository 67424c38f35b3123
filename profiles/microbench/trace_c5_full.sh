# kernel trace of full-size uniform C5 (profiles/fullsize_stratified.py): where the build's time goes at N = 10M
OUT=$PWD/gpurun_out/r04/c5_full_trace; mkdir -p $OUT; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/profiles/fullsize_stratified.py > $OUT/line.json 2> $OUT/err.log < /dev/null
f=$(ls $OUT/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats.csv; python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1]))):
    if 'apss' in r['Name'] or 'rocclr' in r['Name']:
        print(r['Name'][:64].ljust(64), r['Calls'].rjust(4), '%11.1f us avg %11.1f us total' % (float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3))
PY
fi
rm -rf $OUT/trace
