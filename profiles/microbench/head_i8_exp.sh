# INT8 vs bf16 rendering of the dense-head block on the skewed workloads (bench.py lines, reduced to the figures that matter)
mkdir -p gpurun_out/r04
for wl in c3z1 c5z; do
  for dbg in "" "head_bf16"; do
    echo "== $wl APSS_DEBUG=$dbg"
    APSS_DEBUG=$dbg timeout -k 10 400 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-two-directional-row 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print({k: d.get(k) for k in ('ms_per_step','head_kernel_ms','probe_kernel_ms','head_terms','head_survivors','filter_survivors','result_pairs_per_step','rescore_ms','build_ms')})
r=[d['roofline']]+[d[k] for k in ('roofline_dense_head','roofline_sparse_filter') if k in d]
print([(x['bound'], round(x['frac'],3), x['kernel'][:40]) for x in r])"
  done
done
