"""A shard engine backed by the CPU oracle, so that apss.dist's host logic (term ranges, candidate exchange, union,
all-reduce of partial scores) can run on CPU under gloo.  Test infrastructure: lives in tests/, never shipped."""
import numpy as np
import torch


class OracleShardEngine:
    def __init__(self, dim, theta, term_range):
        self.dim, self.theta, self.lo, self.hi = dim, theta, term_range[0], term_range[1]
        self.stats = {}

    def load(self, rp, idx, val, row_range=None):
        import scipy.sparse as sp
        n = len(rp) - 1
        x = sp.csr_matrix((val, idx, rp), shape=(n, self.dim), dtype=np.float64)
        self.xs = x[:, self.lo:self.hi].tocsr()  # the shard's slice of every vector
        self.sub = np.sqrt(np.asarray(self.xs.multiply(self.xs).sum(axis=1)).ravel())
        self.n = n
        self.r0, self.r1 = (0, n) if row_range is None else row_range

    def candidates(self):
        g = (self.xs @ self.xs[self.r0:self.r1].T).tocoo()  # queries: all rows; candidates: this shard's rows
        col = g.col + self.r0
        off = g.row != col
        r, c, v = g.row[off], col[off], g.data[off]
        keep = v >= self.theta * self.sub[r] * self.sub[c] * 0.999999
        self.stats = {"posting_visits": 0, "candidate_pairs": int(off.sum())}
        return torch.from_numpy(r[keep].astype(np.int64)), torch.from_numpy(c[keep].astype(np.int64))

    def partial(self, q, c):
        q, c = q.numpy(), c.numpy()
        if q.size == 0:
            return torch.zeros(0, dtype=torch.float32)
        v = np.asarray(self.xs[q].multiply(self.xs[c]).sum(axis=1)).ravel()
        return torch.from_numpy(v.astype(np.float32))
