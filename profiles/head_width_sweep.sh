#!/bin/bash
# bench.py on the skewed workloads for several head widths (0 = the library's policy): which fold depth pays.
# Usage: profiles/head_width_sweep.sh <outdir> <workload> <steps> K...
out=$1; wl=$2; steps=$3; shift 3
for k in "$@"; do
  python bench.py --workload "$wl" --head-terms "$k" --steps "$steps" --warmup 1 --no-cpu-baseline --no-exact-row | grep "^{" > "$out/${wl}_k$k.json"
done
python - "$out" "$wl" "$@" <<PY
import json, sys
out, wl = sys.argv[1], sys.argv[2]
for k in sys.argv[3:]:
    d = json.load(open("%s/%s_k%s.json" % (out, wl, k)))
    print(wl, "K", k, "terms", d.get("head_terms"), "ms/step %.1f" % d["ms_per_step"], "probe %.1f" % d["probe_kernel_ms"], "head %.1f" % d.get("head_kernel_ms", 0),
          "build %.1f" % d["build_ms"], "rescore %.1f" % d["rescore_ms"], "visits %.3g" % d["posting_visits_per_step"], "res", d["result_pairs_per_step"],
          "surv", d["filter_survivors"], d.get("head_survivors"), (d.get("roofline_sparse_filter") or d["roofline"])["kernel"][:60])
PY
