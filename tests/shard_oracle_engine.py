"""A shard engine backed by the CPU oracle, so that apss.dist's host logic (term ranges, candidate exchange, union,
all-reduce of partial scores) can run on CPU under gloo.  Test infrastructure: lives in tests/, never shipped."""
import numpy as np
import torch


class OracleShardEngine:
    """head = (terms, part, n_parts): the join's dense-head block (apss.dist module docstring) -- its terms are in no
    shard's tail test; this shard runs the head test on the 64-row candidate tiles t with t % n_parts == part."""

    def __init__(self, dim, theta, term_range, head=None):
        self.dim, self.theta, self.lo, self.hi = dim, theta, term_range[0], term_range[1]
        self.head = head
        self.stats = {}

    def load(self, rp, idx, val, row_range=None, query_spans=None):
        import scipy.sparse as sp
        n = len(rp) - 1
        x = sp.csr_matrix((val, idx, rp), shape=(n, self.dim), dtype=np.float64)
        self.xs = x[:, self.lo:self.hi].tocsr()  # the shard's slice of every vector (exact partial scores: head terms included)
        self.full = np.sqrt(np.asarray(x.multiply(x).sum(axis=1)).ravel())
        self.xt = self.xs
        self.w = None
        if self.head is not None:
            is_head = np.zeros(self.dim, bool)
            is_head[np.asarray(self.head[0])] = True
            mask = sp.diags((~is_head[self.lo:self.hi]).astype(np.float64))
            self.xt = (self.xs @ mask).tocsr()  # the tail test's slice: the range without the block's terms
            xh = x[:, np.asarray(self.head[0])].tocsr()
            hn = np.sqrt(np.asarray(xh.multiply(xh).sum(axis=1)).ravel())
            self.w = sp.diags(np.where(hn > 0, self.full / np.where(hn > 0, hn, 1.0), 0.0)) @ xh  # w_x = x_H |x| / |x_H|
        self.sub = np.sqrt(np.asarray(self.xt.multiply(self.xt).sum(axis=1)).ravel())
        self.n = n
        self.r0, self.r1 = (0, n) if row_range is None else row_range
        # the symmetric scheme across row ranges (apss.dist.HipShardEngine.load): own rows x own rows, then only the rows of
        # `query_spans` as an outside batch whose pairs the caller mirrors
        self.sym = row_range is not None and query_spans is not None
        self.o_rows = np.concatenate([np.arange(a, b) for a, b in query_spans] + [np.zeros(0, np.int64)]).astype(np.int64) if self.sym else None

    def phases(self):
        return ("own", "outside") if self.sym else ("all",)

    def run_phase(self, phase):
        rows = {"own": np.arange(self.r0, self.r1), "outside": self.o_rows, "all": np.arange(self.n)}[phase]
        first = phase in ("own", "all")
        prev = dict(self.stats) if not first else {"posting_visits": 0, "candidate_pairs": 0}
        self._pairs = self.candidates(rows)
        self.stats = {"posting_visits": 0, "candidate_pairs": prev["candidate_pairs"] + (2 if phase == "outside" else 1) * self.stats["candidate_pairs"]}
        return int(self._pairs[0].numel())

    def phase_pairs(self, n):
        return self._pairs

    def phase_partial(self, q, c):
        return self.partial(q, c)

    def candidates(self, query_rows=None):
        qr = np.arange(self.n) if query_rows is None else np.asarray(query_rows, np.int64)
        if qr.size == 0:
            self.stats = {"posting_visits": 0, "candidate_pairs": 0}
            return torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64)
        g = (self.xt[qr] @ self.xt[self.r0:self.r1].T).tocoo()  # queries: the given rows (default all); candidates: this shard's rows
        col = g.col + self.r0
        grow = qr[g.row]
        off = grow != col
        r, c, v = grow[off], col[off], g.data[off]
        # p_g >= theta |q_g||c_g| / (|q||c|)
        keep = v * self.full[r] * self.full[c] >= self.theta * self.sub[r] * self.sub[c] * 0.999999
        self.stats = {"posting_visits": 0, "candidate_pairs": int(off.sum())}
        r, c = r[keep], c[keep]
        if self.w is not None:
            rows = np.arange(self.r0, self.r1)
            mine = rows[((rows - self.r0) // 64) % self.head[2] == self.head[1]]  # this shard's candidate tiles (by slot)
            gh = (self.w[qr] @ self.w[mine].T).tocoo()
            hc = mine[gh.col]
            hr = qr[gh.row]
            bound = float(self.full.max()) ** 2
            hk = (hr != hc) & (gh.data >= self.theta - 0.008 * bound)
            r, c = np.concatenate([r, hr[hk]]), np.concatenate([c, hc[hk]])
        return torch.from_numpy(r.astype(np.int64)), torch.from_numpy(c.astype(np.int64))

    def partial(self, q, c):
        q, c = q.numpy(), c.numpy()
        if q.size == 0:
            return torch.zeros(0, dtype=torch.float32)
        v = np.asarray(self.xs[q].multiply(self.xs[c]).sum(axis=1)).ravel()
        return torch.from_numpy(v.astype(np.float32))
