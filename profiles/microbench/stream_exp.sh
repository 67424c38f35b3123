mkdir -p gpurun_out/r04
for dbg in "" "chunks=256" "chunks=128" "chunks=512" "build_lds" "build_lds,chunks=256"; do
  echo "== APSS_DEBUG=$dbg"
  APSS_DEBUG=$dbg timeout -k 10 200 python bench_stream.py --batches 16384,1024 --no-refcpu 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for b,s in d['batch_size'].items():
    print(b, 'wall %.1f ms probe %.1f build %.1f rest %.1f' % (s['wall_s']*1e3, s['probe_kernels_ms_sum'], s['build_ms_sum'], s['rest_ms_sum']))
"
done
