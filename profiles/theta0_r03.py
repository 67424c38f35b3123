"""similarityThreshold = 0 -- what the reference's server template ships (conf/app_server_template.conf:17; the threshold test
is `sim >= similarityThreshold`, IndexingWorkerActor.scala:93) -- on the general kernel k_probe<2, ..>: every pair sharing a
term is a result, so the join is OUTPUT-bound: 12 B per emitted pair (int32 query row, int32 candidate slot, fp32 score)
against the HBM write bandwidth.  Shapes: the template's (vectorDim 1024, conf/app_server_template.conf:6) and
BASELINE.json configs[1]'s (dim 10k, nnz 50, Zipf(1)), batch self-joins at sizes whose output fits comfortably; the same
batches at theta = 0.7 / 0.5 on the two-pass join for contrast.
Usage: python profiles/theta0_r03.py > profiles/r03_theta0.json"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "all-pairs-similarity_amd"))
from apss import synth  # noqa: E402
from apss.engine import ApssIndex  # noqa: E402

out = {}
for name, n, dim, nnz, zipf, theta_pos in (("template_dim1024", 60_000, 1024, 40, 0.0, 0.7), ("c2_shape", 30_000, 10_000, 50, 1.0, 0.5)):
    rp, idx, val = synth.make_vectors(n, dim, nnz, zipf, seed=11)
    row = {"n": n, "dim": dim, "nnz": nnz, "zipf_s": zipf}
    for theta in (0.0, theta_pos):
        with ApssIndex(dim, theta) as ix:
            ix.insert(np.arange(n), rp, idx, val)
            ix.self_join(fetch=False)  # warm: sizes the result buffers
            t0 = time.perf_counter()
            cnt = ix.self_join(fetch=False)
            wall = time.perf_counter() - t0
            st = ix.stats()
        row["theta_%g" % theta] = {
            "result_pairs": int(cnt), "candidate_pairs": st["candidate_pairs"], "posting_visits": st["posting_visits"],
            "wall_ms": wall * 1e3, "probe_kernel_ms": st["probe_ms"], "probe_launches": st["probe_launches"], "probe_kernel": st["probe_kernel"],
            "pairs_per_s": st["candidate_pairs"] / wall,
            "output_GBps": 12.0 * cnt / (st["probe_ms"] * 1e-3) / 1e9 if st["probe_ms"] > 0 else None,
            "output_frac_of_8TBps": 12.0 * cnt / (st["probe_ms"] * 1e-3) / 8e12 if st["probe_ms"] > 0 else None,
            "algorithmic_frac": 8.0 * st["posting_visits"] / (st["probe_ms"] * 1e-3) / 8e12 if st["probe_ms"] > 0 else None}
    out[name] = row
print(json.dumps(out, indent=1))
