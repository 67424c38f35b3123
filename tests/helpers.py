"""Shared helpers for the parity tests."""
import numpy as np

BAND = 1e-5  # |score - theta| <= BAND: membership may differ between fp32 device and double oracle (Q12)
TOL = 1e-5   # every reported score within TOL of the oracle's double value (BASELINE.json north_star)


def to_map(q, c, s):
    return {(int(a), int(b)): float(v) for a, b, v in zip(q, c, s)}


def assert_same_pairs(got, want, theta, band=BAND, tol=TOL):
    """got / want: dict (q, c) -> score.  Same set outside the threshold band, scores within tol."""
    missing = [k for k in want if k not in got and abs(want[k] - theta) > band]
    extra = [k for k in got if k not in want and abs(got[k] - theta) > band]
    assert not missing, "missing %d pairs, e.g. %s" % (len(missing), [(k, want[k]) for k in missing[:5]])
    assert not extra, "extra %d pairs, e.g. %s" % (len(extra), [(k, got[k]) for k in extra[:5]])
    worst = 0.0
    for k, v in got.items():
        if k in want:
            worst = max(worst, abs(v - want[k]))
    assert worst <= tol, "score error %g > %g" % (worst, tol)


def topk(m, k):
    """first k of the set sorted by (-score, q, c) (SURVEY Q3 definition of top-k)"""
    return sorted(m.items(), key=lambda kv: (-kv[1], kv[0][0], kv[0][1]))[:k]


def scipy_pairs(dim, theta, rowptr, indices, values):
    """Independent float64 cross-check: threshold X @ X.T, drop the diagonal, keep only structural non-zeros
    (pairs sharing a term), like the intended semantics."""
    import scipy.sparse as sp
    n = len(rowptr) - 1
    x = sp.csr_matrix((values, indices, rowptr), shape=(n, dim), dtype=np.float64)
    b = x.copy()
    b.data[:] = 1.0
    shared = (b @ b.T).tocoo()          # structural pattern: shares >= 1 term
    g = (x @ x.T).tocsr()
    off = shared.row != shared.col
    r, c = shared.row[off], shared.col[off]
    v = np.asarray(g[r, c]).ravel()
    keep = v >= theta
    return {(int(i), int(j)): float(s) for i, j, s in zip(r[keep], c[keep], v[keep])}
