"""Probe-kernel time of the same C3 self-join under different allocation histories of the handle's buffers
(investigation of a 1.7x spread; see DESIGN.md 'allocation order')."""
import sys
sys.path[:0] = ['.', 'all-pairs-similarity_amd']
import numpy as np, torch
from apss import synth
from apss.engine import ApssIndex
cfg = synth.CONFIGS['c3']
n = 1_000_000
rp, idx, val = synth.make_vectors(n, cfg['dim'], cfg['nnz'], 0.0, cfg['seed'])
dev = torch.device('cuda', 0)
d_rp = torch.from_numpy(rp).to(dev); d_idx = torch.from_numpy(idx).to(dev); d_val = torch.from_numpy(val.astype(np.float32)).to(dev)
d_ids = torch.arange(n, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
for name, kw in (("no hints", {}), ("rows+nnz hints", dict(capacity_rows=n, capacity_nnz=idx.size)),
                 ("nnz hint only", dict(capacity_nnz=idx.size)), ("rows hint only", dict(capacity_rows=n))):
    ix = ApssIndex(cfg['dim'], cfg['theta'], **kw)
    for it in range(3):
        ix.clear()
        ix.insert_and_query_dev(d_ids, d_rp, d_idx, d_val)
        print(name, "iter", it, "probe ms %.1f" % ix.stats()["probe_ms"], "build %.1f" % ix.stats()["build_ms"])
    ix.close()
