"""C3 (N=1M, dim=100k, nnz=100, theta=0.8) with a sign per term (a third negative): the two-pass join's signed
instantiation against the general kernel (APSS_FORCE_GENERAL)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "all-pairs-similarity_amd"))
from apss import synth  # noqa: E402
from apss.engine import ApssIndex  # noqa: E402

c, rp, idx, val = synth.make_config("c3")
sign = np.random.default_rng(1).choice([-1.0, 1.0, 1.0], size=c["dim"])
sval = (val * sign[idx]).astype(np.float32)
dev = torch.device("cuda", 0)
d = [torch.from_numpy(x).to(dev) for x in (rp, idx, sval)]
ids = torch.arange(c["n"], dtype=torch.int64, device=dev)
for env in ({}, {"APSS_FORCE_GENERAL": "1"}):
    os.environ.pop("APSS_FORCE_GENERAL", None)
    os.environ.update(env)
    with ApssIndex(c["dim"], c["theta"]) as ix:
        ix.insert_and_query_dev(ids, *d)
        ix.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = ix.insert_and_query_dev(ids, *d)
        torch.cuda.synchronize()
        st = ix.stats()
        print(env, "pairs", n, "step ms %.1f" % ((time.perf_counter() - t0) * 1e3), "probe ms %.1f" % st["probe_ms"],
              "survivors", st["filter_survivors"], "candidate pairs", st["candidate_pairs"], flush=True)
