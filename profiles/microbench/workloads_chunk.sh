mkdir -p gpurun_out/r01L
for w in c2 c3z; do
  timeout -k 10 300 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r01L/wl_${w}_16.log 2>&1 || exit 1
  APSS_CX_CHUNK8=1 timeout -k 10 300 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r01L/wl_${w}_8.log 2>&1 || exit 1
done
for f in gpurun_out/r01L/wl_*.log; do echo $f; tail -1 $f | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['probe_kernel_ms'], d['ms_per_step'], d['result_pairs_per_step'], d['candidate_pairs_per_step'])"; done
