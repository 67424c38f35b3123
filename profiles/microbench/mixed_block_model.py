"""Model of the dense head's selectivity for different block geometries, in numpy/torch (no library call): E head terms with a
column each + the other head terms folded into F columns, ONE test on the E + F wide row (w = x_mixed |x| / |x_H|), bf16
rounded.  Counts the pairs of a sample that pass  w_q . w_c >= theta - 0.008  (chance pairs + the planted near-duplicates).
Usage: python profiles/microbench/mixed_block_model.py  (prints a table; what DESIGN.md 5b quotes)"""
import sys
import numpy as np
import torch
sys.path.insert(0, "all-pairs-similarity_amd")
from apss import synth

for name, n, dim, nnz, theta, seed in (("c3z1-like", 30000, 100000, 100, 0.8, 20243), ("c5z-like", 30000, 1000000, 200, 0.9, 20245)):
    rp, idx, val = synth.make_vectors_zipf_dev(n, dim, nnz, 1.0, seed, "cuda")
    idx, val = idx.reshape(-1).cpu().numpy(), val.reshape(-1).double().cpu().numpy()
    df = np.bincount(idx, minlength=dim)
    order = np.lexsort((np.arange(dim), -df))
    rows = np.repeat(np.arange(n), nnz)
    full = np.sqrt(np.bincount(rows, weights=val ** 2, minlength=n))
    for K in (8192, 2048):
        terms = order[:K]
        for E, F in ((256, 0), (128, 128), (64, 192), (192, 64), (64, 64), (32, 96), (96, 32), (0, 128), (0, 256)):
            col = np.full(dim, -1)
            kk = K if F else min(K, E)
            for i, t in enumerate(terms[:kk]):
                col[t] = i if i < E else E + (i - E) % F
            hp = col[idx]
            m = hp >= 0
            W = np.zeros((n, max(E + F, 1)))
            np.add.at(W, (rows[m], hp[m]), val[m])
            hn = np.sqrt(np.bincount(rows[m], weights=val[m] ** 2, minlength=n))
            W *= (full / np.where(hn > 0, hn, 1))[:, None]
            Wt = torch.from_numpy(W).cuda().to(torch.bfloat16).float()
            D = Wt @ Wt.T
            D.fill_diagonal_(0)
            print(name, "K", kk, "exact", E, "folded cols", F, "pairs passing", int((D >= theta - 0.008 - 1e-5).sum().item()), flush=True)
