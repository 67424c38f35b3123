import os, sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'all-pairs-similarity_amd'))
from apss import synth
from apss.engine import ApssIndex
n, dim, nnz = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (60000, 20000, 20)
theta = 0.7
rp, idx, val = synth.make_vectors(n, dim, nnz, 0.0, seed=3, dup_frac=0.05)
X = sp.csr_matrix((np.ones_like(val, dtype=np.float32), idx, rp), shape=(n, dim))
P = (X @ X.T).tocsr()
truth = P.nnz - n
print("truth", truth)
for env in [{}, {"APSS_CX_CHUNK8": "1"}, {"APSS_CX_U3": "1"}]:
    for k in ("APSS_CX_CHUNK8", "APSS_CX_U3"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with ApssIndex(dim, theta) as ix:
        q, c, s = ix.insert_and_query(np.arange(n), rp, idx, val)
        st = ix.stats()
    print(env, st["candidate_pairs"], st["candidate_pairs"] - truth, len(q), st["filter_survivors"], st["tiles"])
