#!/bin/bash
# k_probe_even vs k_probe_coarse (APSS_DEBUG=no_even) on a T = 8 / T = 2 term shard of C3, on C3 itself and on C5's shape:
# step time and result-set size of each (run on the GPU box from the repo root).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ev
for dbg in "" "no_even"; do
  for g in 8,1,3,0 2,1,1,0 1,1,0,0; do
    APSS_DEBUG=$dbg timeout -k 10 200 python3 bench.py --solo $g --steps 3 --no-cpu-baseline > "gpurun_out/ev/solo_${g}_${dbg:-even}.json" 2> "gpurun_out/ev/solo_${g}_${dbg:-even}.err" || exit 1
  done
  APSS_DEBUG=$dbg timeout -k 10 300 python3 bench.py --workload c5s --steps 3 --warmup 1 --no-cpu-baseline --no-exact-row > "gpurun_out/ev/c5s_${dbg:-even}.json" 2> "gpurun_out/ev/c5s_${dbg:-even}.err" || exit 1
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ev/*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f,'ERR',e); continue
    r=d.get('roofline',{})
    print(f.split('/')[-1], 'ms_per_step',round(d['ms_per_step'],2),'probe_ms',r.get('probe_kernel_ms',r.get('kernel_ms')),'frac',r.get('frac'), 'res',d.get('result_pairs_per_step'))
PY
