#!/bin/bash
# what does one step of the register window cost?  24576-row tiles (25 chunks per wave: both windows hold them) with the
# 5-step and the 4-step kernel, and the default 32768-row tiles for reference
mkdir -p gpurun_out/r01L
run() { echo -n "$1: "; env $2 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['probe_kernel_ms'], d['ms_per_step'], d['candidate_pairs_per_step'])"; }
run "cb=32768 U=5" "APSS_X=1" || exit 1
run "cb=24576 U=5" "APSS_CX_TILE=24576" || exit 1
run "cb=24576 U=4" "APSS_CX_TILE=24576 APSS_CX_U4=1" || exit 1
run "cb=32768 U=4" "APSS_CX_U4=1" || exit 1
