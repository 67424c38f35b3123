#!/bin/bash
# probe / step time of the secondary workloads
mkdir -p gpurun_out/r01L
for w in ${@:-c3 c2 c3z}; do
  echo -n "$w: "; timeout -k 10 300 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['probe_kernel_ms'], d['ms_per_step'], d['result_pairs_per_step'], d['candidate_pairs_per_step'])" || exit 1
done
