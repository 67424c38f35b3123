# single-vector message latency, host-pointer path (bench_latency.py), with and without the chained exact pass
mkdir -p gpurun_out/r04
for dbg in "" "no_chain"; do
  for shape in production c3; do
    echo "== APSS_DEBUG=$dbg shape=$shape"
    APSS_DEBUG=$dbg timeout -k 10 200 python bench_latency.py --backend gpu --shape $shape --messages 3000 --interval-ms 0 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)['gpu']; print({k: round(v,4) if isinstance(v,float) else v for k,v in d.items()})"
  done
done
