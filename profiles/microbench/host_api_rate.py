import sys, time, json
sys.path[:0] = ['.', 'all-pairs-similarity_amd']
import numpy as np
from apss import synth
from apss.engine import ApssIndex
cfg = synth.CONFIGS['c3']
rp, idx, val = synth.make_vectors(cfg['n'], cfg['dim'], cfg['nnz'], cfg['zipf_s'], cfg['seed'])
ids = np.arange(cfg['n'], dtype=np.int64)
ix = ApssIndex(cfg['dim'], cfg['theta'], capacity_rows=cfg['n'], capacity_nnz=idx.size)
ts = []
for it in range(3):
    ix.clear()
    t0 = time.perf_counter()
    q, c, s = ix.insert_and_query(ids, rp, idx, val)
    ts.append(time.perf_counter() - t0)
st = ix.stats()
print(json.dumps({"host_api_s": ts, "pairs": int(q.size), "cand": st["candidate_pairs"], "pairs_per_s_pcie_inclusive": st["candidate_pairs"]/min(ts)}))
