#!/bin/bash
# query chunks per tile (APSS_CHUNKS) sweep at the default workload
mkdir -p gpurun_out/r01L
for c in 512 768 1024 1536 2048 4096; do
  APSS_CHUNKS=$c timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r01L/chunks_$c.log 2>&1 || exit 1
  echo -n "$c: "; grep "^{" gpurun_out/r01L/chunks_$c.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['probe_kernel_ms'], d['ms_per_step'])"
done
