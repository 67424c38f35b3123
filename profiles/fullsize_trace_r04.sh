#!/bin/bash
# rocprofv3 kernel trace of configs[4] at full size (profiles/fullsize_powerlaw.py); on the GPU box, from the repo root
OUT=$PWD/gpurun_out/r04/c5z_full_trace; mkdir -p $OUT
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/profiles/fullsize_powerlaw.py > $OUT/line.json 2> $OUT/err.log < /dev/null
echo "rc=$?"
f=$(ls $OUT/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats.csv; cut -c1-90,200-320 "$f" | head -30; else echo "no kernel_stats.csv"; fi
rm -rf $OUT/trace
tail -2 $OUT/err.log
