OUT=$PWD/gpurun_out/r04/c5s_trace; mkdir -p $OUT; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --workload c5s --steps 2 --warmup 1 --no-cpu-baseline --no-exact-row > $OUT/line.json 2> $OUT/err.log < /dev/null
f=$(ls $OUT/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats.csv; python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r['Name'][:64].ljust(64), r['Calls'].rjust(4), '%9.1f us avg' % (float(r['AverageNs'])/1e3))
PY
fi
rm -rf $OUT/trace
