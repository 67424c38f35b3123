import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.getcwd(), "all-pairs-similarity_amd"))
from apss.dist import HipShardEngine, join_shards_local, term_ranges
from apss import synth
n, dim, nnz, theta = 200000, 100000, 100, 0.8
rp, idx, val = synth.make_vectors(n, dim, nnz, 0.0, seed=3, dup_frac=0.05)
rng = np.random.default_rng(1)
val = val * np.where(rng.random(val.size) < 0.3, -1.0, 1.0)   # a third of the weights negative
dev = torch.device("cuda", 0)
ranges = term_ranges(np.bincount(idx, minlength=dim), 4)
engines = [HipShardEngine(dim, theta, tr, dev) for tr in ranges]
for e in engines: e.load(rp, idx, val)
q, c, s, nc = join_shards_local(engines, n, theta)
print(os.environ.get("APSS_DEBUG", ""), "pairs", len(q), "probe ms per shard", [round(e.stats["probe_ms"], 2) for e in engines])
