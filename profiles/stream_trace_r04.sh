#!/bin/bash
# rocprofv3 kernel trace of one streamed batch size (bench_stream.py); run from the repo root ON THE GPU BOX:
#   bash profiles/stream_trace_r04.sh 16384 tag
B=${1:-16384}; TAG=${2:-r04}
OUT=$PWD/gpurun_out/r04/stream_${B}_${TAG}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $OLDPWD/bench_stream.py --batches $B --no-refcpu > $OUT/line.json 2> $OUT/err.log < /dev/null
echo "rc=$?"
f=$(ls $OUT/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats.csv; head -40 "$f"; else echo "no kernel_stats.csv"; ls -R $OUT | head -20; fi
rm -rf $OUT/trace
