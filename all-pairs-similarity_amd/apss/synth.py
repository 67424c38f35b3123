"""Seeded synthetic sparse-vector workloads (the reference ships no generator).

Per vector: `nnz` distinct dims from the term distribution (uniform, or Zipf p_t ~ 1/rank^s with term ids
randomly permuted), values |N(0,1)| + 0.05, L2-normalised in double the way the reference's client does
(benchmark/LoadGenerator.scala:34-37).  Random sparse vectors almost never reach cosine 0.8, so a fraction
`dup_frac` of the rows is replaced by a perturbed copy of an earlier base row (10 % of the terms replaced,
values scaled by U(0.9, 1.1), re-normalised) which gives a non-empty, checkable result set.
"""
import numpy as np

# BASELINE.json configs -> (N, dim, nnz, zipf_s, theta); seed = 20240 + config index
CONFIGS = {
    "c2": dict(n=100_000, dim=10_000, nnz=50, zipf_s=1.0, theta=0.5, seed=20241),
    "c3": dict(n=1_000_000, dim=100_000, nnz=100, zipf_s=0.0, theta=0.8, seed=20242),
    "c3z": dict(n=1_000_000, dim=100_000, nnz=100, zipf_s=0.5, theta=0.8, seed=20242),
    "c5": dict(n=10_000_000, dim=1_000_000, nnz=200, zipf_s=0.0, theta=0.9, seed=20244),
    "c5s": dict(n=2_000_000, dim=1_000_000, nnz=200, zipf_s=0.0, theta=0.9, seed=20244),  # C5 shape at one fifth of N
    # skewed variants drawn by make_vectors_zipf_dev (device-side generator; the per-row host loop of make_vectors
    # needs minutes per million rows): C3 with Zipf(1) terms, and BASELINE.json configs[4] "power-law" at one fifth of N
    "c3z1": dict(n=1_000_000, dim=100_000, nnz=100, zipf_s=1.0, theta=0.8, seed=20243, gen="zipf_dev"),
    "c5z": dict(n=2_000_000, dim=1_000_000, nnz=200, zipf_s=1.0, theta=0.9, seed=20245, gen="zipf_dev"),
}


def _draw_terms_uniform(rng, rows, dim, nnz):
    # sorted draw from [0, dim - nnz] plus 0..nnz-1 is strictly increasing and < dim
    a = rng.integers(0, dim - nnz + 1, size=(rows, nnz), dtype=np.int64)
    a.sort(axis=1)
    a += np.arange(nnz, dtype=np.int64)[None, :]
    return a.astype(np.int32)


def _draw_terms_zipf(rng, rows, dim, nnz, cdf, perm):
    out = np.empty((rows, nnz), np.int32)
    over = 3
    for r0 in range(0, rows, 4096):
        r1 = min(rows, r0 + 4096)
        m = r1 - r0
        draws = np.searchsorted(cdf, rng.random((m, nnz * over)), side="right").astype(np.int64)
        np.minimum(draws, dim - 1, out=draws)
        for i in range(m):
            u, first = np.unique(draws[i], return_index=True)
            u = u[np.argsort(first)][:nnz]
            if u.size < nnz:  # top up with unused terms
                extra = np.setdiff1d(rng.permutation(dim)[: 4 * nnz], u)[: nnz - u.size]
                u = np.concatenate([u, extra])
            out[r0 + i] = np.sort(perm[u])
    return out


def make_vectors(n, dim, nnz, zipf_s=0.0, seed=0, dup_frac=0.05):
    """Returns CSR (rowptr int64[n+1], indices int32[n*nnz], values float64[n*nnz]); rows have exactly
    `nnz` strictly increasing indices and unit L2 norm."""
    assert 0 < nnz <= dim
    rng = np.random.Generator(np.random.PCG64(seed))
    if zipf_s > 0:
        p = 1.0 / np.arange(1, dim + 1, dtype=np.float64) ** zipf_s
        cdf = np.cumsum(p / p.sum())
        perm = rng.permutation(dim).astype(np.int64)
        idx = _draw_terms_zipf(rng, n, dim, nnz, cdf, perm)
    else:
        idx = _draw_terms_uniform(rng, n, dim, nnz)
    val = np.abs(rng.standard_normal((n, nnz))) + 0.05

    # planted near-duplicates (sources are BASE rows, so no chains)
    if dup_frac > 0 and n > 1:
        is_dup = rng.random(n) < dup_frac
        is_dup[0] = False
        rows = np.nonzero(is_dup)[0]
        src = (rng.random(rows.size) * rows).astype(np.int64)  # uniform in [0, row)
        base_idx = idx[src].copy()
        base_val = val[src] * rng.uniform(0.9, 1.1, size=(rows.size, nnz))
        n_rep = max(1, nnz // 10)
        for j in range(rows.size):
            pos = rng.choice(nnz, size=n_rep, replace=False)
            new_terms = rng.integers(0, dim, size=n_rep)
            row = base_idx[j]
            for p_, t_ in zip(pos, new_terms):
                if t_ not in row:
                    row[p_] = t_
            order = np.argsort(row, kind="stable")
            base_idx[j] = row[order]
            base_val[j] = base_val[j][order]
        idx[rows] = base_idx
        val[rows] = base_val

    val /= np.sqrt((val * val).sum(axis=1, keepdims=True))
    rowptr = np.arange(0, (n + 1) * nnz, nnz, dtype=np.int64)
    return rowptr, idx.reshape(-1).astype(np.int32), val.reshape(-1).astype(np.float64)


def make_config(name, n=None, device=None):
    """One of BASELINE.json's synthetic configs (optionally with n rows instead of the config's) as host CSR."""
    c = dict(CONFIGS[name])
    if n is not None:
        c["n"] = n
    if c.get("gen") == "zipf_dev":
        import torch
        dev = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
        rp, idx, val = make_vectors_zipf_dev(c["n"], c["dim"], c["nnz"], c["zipf_s"], c["seed"], dev)
        return c, rp.cpu().numpy(), idx.reshape(-1).cpu().numpy(), val.reshape(-1).double().cpu().numpy()
    rowptr, idx, val = make_vectors(c["n"], c["dim"], c["nnz"], c["zipf_s"], c["seed"])
    return c, rowptr, idx, val


def workload_counts(dim, rowptr, indices):
    """Analytic work of a full self-join: (postings, posting visits = sum_t df_t^2)."""
    df = np.bincount(indices, minlength=dim).astype(np.float64)
    return int(indices.size), float((df * df).sum())


def make_vectors_stratified_dev(n, dim, nnz, seed, device, dup_frac=0.05, block=1 << 20):
    """Device-side generator for the configs too large to draw on the host in reasonable time (C5: 2e9 entries).

    Row i draws its j-th term uniformly from stratum j = [j*w, (j+1)*w), w = dim // nnz: strictly increasing by
    construction, df uniform over the terms, and entry j of any two rows can only collide with entry j -- so the exact
    dot product of two rows is an ELEMENTWISE sum, which gives a full-size parity check that needs no oracle
    (`stratified_dot`).  Planted near-duplicates as in make_vectors (10 % of the terms redrawn inside their stratum,
    values x U(0.9, 1.1), re-normalised); sources are base rows.  Returns torch tensors
    (rowptr int64[n+1], idx int32[n, nnz], val float32[n, nnz], src int64[n] (-1 = base row))."""
    import torch
    w = dim // nnz
    assert w >= 2
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    idx = torch.empty((n, nnz), dtype=torch.int32, device=device)
    val = torch.empty((n, nnz), dtype=torch.float32, device=device)
    base = (torch.arange(nnz, device=device, dtype=torch.int32) * w)[None, :]
    for r0 in range(0, n, block):
        r1 = min(n, r0 + block)
        idx[r0:r1] = torch.randint(0, w, (r1 - r0, nnz), generator=g, device=device, dtype=torch.int32) + base
        v = torch.randn((r1 - r0, nnz), generator=g, device=device).abs_() + 0.05
        val[r0:r1] = v / v.norm(dim=1, keepdim=True)
    src = torch.full((n,), -1, dtype=torch.int64, device=device)
    if dup_frac > 0 and n > 1:
        is_dup = torch.rand(n, generator=g, device=device) < dup_frac
        is_dup[0] = False
        rows = is_dup.nonzero().flatten()
        # source = a base (non-duplicate) row before it: draw, then walk back to the nearest base row
        cand = (torch.rand(rows.numel(), generator=g, device=device, dtype=torch.float64) * rows.to(torch.float64)).to(torch.int64)
        last_base = torch.where(~is_dup, torch.arange(n, device=device), torch.zeros((), dtype=torch.int64, device=device))
        last_base = torch.cummax(last_base, 0).values  # nearest base row at or before each position (row 0 is base)
        s = last_base[cand]
        src[rows] = s
        for b0 in range(0, rows.numel(), block):
            rr, ss = rows[b0:b0 + block], s[b0:b0 + block]
            m = rr.numel()
            redraw = torch.rand((m, nnz), generator=g, device=device) < 0.10
            new_t = torch.randint(0, w, (m, nnz), generator=g, device=device, dtype=torch.int32) + base
            idx[rr] = torch.where(redraw, new_t, idx[ss])
            v = val[ss] * (0.9 + 0.2 * torch.rand((m, nnz), generator=g, device=device))
            val[rr] = v / v.norm(dim=1, keepdim=True)
    rowptr = torch.arange(0, (n + 1) * nnz, nnz, dtype=torch.int64, device=device)
    return rowptr, idx, val, src


def stratified_dot(idx, val, a, b, block=1 << 20):
    """exact fp32 dot products of rows a[k], b[k] of a stratified batch (entry j only meets entry j), in float64"""
    import torch
    out = torch.empty(a.numel(), dtype=torch.float64, device=idx.device)
    for k0 in range(0, a.numel(), block):
        x, y = a[k0:k0 + block], b[k0:k0 + block]
        out[k0:k0 + block] = ((idx[x] == idx[y]) * (val[x].double() * val[y].double())).sum(dim=1)
    return out


def rows_dot(idx, val, a, b, block=2048):
    """exact dot products, in float64, of rows a[k], b[k] of a batch with a fixed number of entries per row (idx [n, nnz]
    ascending per row, val [n, nnz]): every entry of one row against every entry of the other"""
    import torch
    out = torch.empty(a.numel(), dtype=torch.float64, device=idx.device)
    for k0 in range(0, a.numel(), block):
        x, y = a[k0:k0 + block], b[k0:k0 + block]
        eq = idx[x][:, :, None] == idx[y][:, None, :]
        out[k0:k0 + block] = (eq * (val[x].double()[:, :, None] * val[y].double()[:, None, :])).sum(dim=(1, 2))
    return out


def make_vectors_zipf_dev(n, dim, nnz, zipf_s, seed, device, dup_frac=0.05, block=1 << 16, return_src=False):
    """Zipf(s) workload drawn with torch on `device` (the GPU for the million-row configs; "cpu" works for tests).

    Per row: 3 * nnz draws with replacement from p_t ~ 1 / rank^s (term ids randomly permuted), the first nnz DISTINCT
    ones in draw order are the row's terms (rows short of nnz distinct draws -- rare -- draw again, 16x as many);
    values |N(0,1)| + 0.05; planted near-duplicates as in make_vectors (10 % of the terms replaced by uniformly drawn
    ones that the row does not hold yet, values x U(0.9, 1.1)); rows L2-normalised (benchmark/LoadGenerator.scala:34-37).
    Returns torch tensors (rowptr int64[n+1], idx int32[n, nnz] ascending per row, val float32[n, nnz]); with return_src
    also src int64[n]: the base row a planted near-duplicate was copied from (-1: a base row)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    p = 1.0 / torch.arange(1, dim + 1, dtype=torch.float64, device=device) ** zipf_s
    cdf = torch.cumsum(p / p.sum(), 0)
    perm = torch.randperm(dim, generator=g, device=device).to(torch.int32)

    def draw_rows(m, over):
        L = nnz * over
        d = torch.searchsorted(cdf, torch.rand((m, L), generator=g, device=device, dtype=torch.float64)).clamp_(max=dim - 1)
        order = torch.argsort(d, dim=1, stable=True)
        ds = torch.gather(d, 1, order)
        first = torch.ones_like(ds, dtype=torch.bool)
        first[:, 1:] = ds[:, 1:] != ds[:, :-1]
        isfirst = torch.zeros_like(first).scatter_(1, order, first)  # first occurrence, in draw order
        keep = isfirst & (torch.cumsum(isfirst, 1) <= nnz)
        return d, keep, keep.sum(1)

    idx = torch.empty((n, nnz), dtype=torch.int32, device=device)
    for r0 in range(0, n, block):
        m = min(n, r0 + block) - r0
        d, keep, cnt = draw_rows(m, 3)
        ok = cnt == nnz
        out = torch.empty((m, nnz), dtype=torch.int64, device=device)
        out[ok] = d[ok][keep[ok]].view(-1, nnz)
        bad = (~ok).nonzero().flatten()
        while bad.numel():
            d2, keep2, cnt2 = draw_rows(bad.numel(), 48)
            ok2 = cnt2 == nnz
            out[bad[ok2]] = d2[ok2][keep2[ok2]].view(-1, nnz)
            bad = bad[~ok2]
        idx[r0:r0 + m] = torch.sort(perm[out], dim=1).values
    val = torch.randn((n, nnz), generator=g, device=device).abs_() + 0.05
    src_all = torch.full((n,), -1, dtype=torch.int64, device=device)

    if dup_frac > 0 and n > 1:
        is_dup = torch.rand(n, generator=g, device=device) < dup_frac
        is_dup[0] = False
        rows = is_dup.nonzero().flatten()
        cand = (torch.rand(rows.numel(), generator=g, device=device, dtype=torch.float64) * rows.to(torch.float64)).to(torch.int64)
        last_base = torch.where(~is_dup, torch.arange(n, device=device), torch.zeros((), dtype=torch.int64, device=device))
        src = torch.cummax(last_base, 0).values[cand]  # nearest base (non-duplicate) row at or before the drawn one
        src_all[rows] = src
        n_rep = max(1, nnz // 10)
        for b0 in range(0, rows.numel(), block):
            rr, ss = rows[b0:b0 + block], src[b0:b0 + block]
            m = rr.numel()
            t = idx[ss].clone()
            v = val[ss] * (0.9 + 0.2 * torch.rand((m, nnz), generator=g, device=device))
            pos = torch.argsort(torch.rand((m, nnz), generator=g, device=device), dim=1)[:, :n_rep]
            new_t = torch.randint(0, dim, (m, n_rep), generator=g, device=device, dtype=torch.int32)
            # a replacement that the row already holds (or that repeats inside the draw) is skipped
            clash = (new_t[:, :, None] == t[:, None, :]).any(2)
            clash |= torch.triu(new_t[:, :, None] == new_t[:, None, :], diagonal=1).any(1)
            cur = torch.gather(t, 1, pos)
            t.scatter_(1, pos, torch.where(clash, cur, new_t))
            order = torch.argsort(t, dim=1, stable=True)
            idx[rr] = torch.gather(t, 1, order)
            val[rr] = torch.gather(v, 1, order)
    val /= val.norm(dim=1, keepdim=True)
    rowptr = torch.arange(0, (n + 1) * nnz, nnz, dtype=torch.int64, device=device)
    return (rowptr, idx, val, src_all) if return_src else (rowptr, idx, val)
