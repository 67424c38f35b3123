"""theta = 0 on the template shape (profiles/theta0_r03.py) with smaller tiles of the general kernel k_probe<2, 512>: more workgroups
per CU (the kernel's rounds are a chain of barriers: accumulate, judge + scan, reserve, write, clear)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "all-pairs-similarity_amd"))
from apss import synth  # noqa: E402
from apss.engine import ApssIndex  # noqa: E402

n, dim, nnz = 60_000, 1024, 40
rp, idx, val = synth.make_vectors(n, dim, nnz, 0.0, seed=11)
out = {}
for tile_rows in (0, 8192, 4096, 2048, 1024):
    with ApssIndex(dim, 0.0, tile_rows=tile_rows) as ix:
        ix.insert(np.arange(n), rp, idx, val)
        ix.self_join(fetch=False)
        t0 = time.perf_counter()
        cnt = ix.self_join(fetch=False)
        wall = time.perf_counter() - t0
        st = ix.stats()
    out[str(tile_rows)] = {"result_pairs": int(cnt), "wall_ms": wall * 1e3, "probe_kernel_ms": st["probe_ms"], "probe_kernel": st["probe_kernel"], "tiles": st["tiles"],
                           "output_frac_of_8TBps": 12.0 * cnt / (st["probe_ms"] * 1e-3) / 8e12}
print(json.dumps(out, indent=1))
