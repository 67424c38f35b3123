"""GPU test of the term-range shard path (SHARD kernels, candidate rule, partial scores): several shard handles on
one GPU, combined in-process exactly as apss.dist combines them across GPUs."""
import numpy as np
import pytest

from apss import synth
from helpers import assert_same_pairs, to_map

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shard_kernel", ["coarse_filter", "single_pass"])
@pytest.mark.parametrize("world,zipf,theta", [(2, 0.0, 0.5), (4, 1.0, 0.6), (8, 0.0, 0.8)])
def test_shards_on_one_gpu_match_oracle(oracle, monkeypatch, world, zipf, theta, shard_kernel):
    import torch
    if shard_kernel == "single_pass":
        monkeypatch.setenv("APSS_DEBUG", "shard_exact")  # read when the handle is created
    from apss.dist import HipShardEngine, join_shards_local, term_ranges
    n, dim, nnz = 4000, 2000, 30
    rp, idx, val = synth.make_vectors(n, dim, nnz, zipf, seed=91, dup_frac=0.1)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    assert len(want) > 100
    dev = torch.device("cuda", 0)
    ranges = term_ranges(np.bincount(idx, minlength=dim), world)
    engines = [HipShardEngine(dim, theta, tr, dev, tile_rows=1024) for tr in ranges]
    for e in engines:
        e.load(rp, idx, val)
    q, c, s, n_cand = join_shards_local(engines, n, theta)
    assert_same_pairs(to_map(q, c, s), want, theta)
    # on uniform data (many terms of every vector in every shard) the candidate rule keeps the exchange tiny
    # compared with the pairs each shard touched; a Zipf head term isolated in its own shard defeats it (every pair
    # sharing that term has within-shard cosine 1) -- still exact, just more candidates
    touched = sum(e.stats["candidate_pairs"] for e in engines)
    if zipf == 0.0:
        assert sum(n_cand) < 0.05 * touched + 20 * len(want)
    # posting visits add up over the shards
    assert sum(e.stats["posting_visits"] for e in engines) == int(synth.workload_counts(dim, rp, idx)[1])


def test_candidate_range_shard_on_one_gpu(oracle):
    """a shard that indexes only a row range and is queried with the whole batch (the D axis of apss.dist)"""
    import torch
    from apss.dist import HipShardEngine, join_shards_local
    n, dim, nnz, theta = 4000, 2000, 30, 0.5
    rp, idx, val = synth.make_vectors(n, dim, nnz, 0.0, seed=92, dup_frac=0.1)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    dev = torch.device("cuda", 0)
    got = {}
    for r0, r1 in ((0, 1500), (1500, 4000)):
        engines = [HipShardEngine(dim, theta, tr, dev, tile_rows=1024) for tr in ((0, 900), (900, 2000))]
        for e in engines:
            e.load(rp, idx, val, (r0, r1))
        q, c, s, _ = join_shards_local(engines, n, theta)
        assert all(r0 <= x < r1 for x in c)
        got.update(to_map(q, c, s))
    assert_same_pairs(got, want, theta)


def test_sparse_term_shards_keep_the_shard_threshold(oracle):
    """few postings per (tile, term) segment: an unsharded handle would switch to 65536-row tiles and the 1024-thread
    filter kernel, which has no shard variant (it once ran with the unsharded threshold and found nothing)"""
    import torch
    from apss.dist import HipShardEngine, join_shards_local, term_ranges
    n, dim, nnz, theta = 70_000, 40_000, 16, 0.6
    rp, idx, val = synth.make_vectors(n, dim, nnz, 0.0, seed=55, dup_frac=0.05)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    assert len(want) > 1000
    dev = torch.device("cuda", 0)
    engines = [HipShardEngine(dim, theta, tr, dev) for tr in term_ranges(np.bincount(idx, minlength=dim), 4)]
    for e in engines:
        e.load(rp, idx, val)
    q, c, s, n_cand = join_shards_local(engines, n, theta)
    assert min(n_cand) > 0
    assert_same_pairs(to_map(q, c, s), want, theta)


@pytest.mark.parametrize("world", [2, 3])
def test_term_shards_with_norms_above_one_lose_nothing(oracle, world):
    """the candidate rule divides by the FULL row norms (p_g >= theta |q_g||c_g| / (|q||c|)), so un-normalised rows go
    through term shards like through one handle: rows scaled to norms between 0.5 and 2.5 (|q||c| up to 6), among them
    the case of two shards with p_g = 0.4, |q_g||c_g| = 2 each that a rule without the division would drop"""
    import torch
    from apss.dist import HipShardEngine, join_shards_local, term_ranges
    n, dim, nnz, theta = 3000, 1200, 20, 0.7
    rp, idx, val = synth.make_vectors(n, dim, nnz, 0.0, seed=93, dup_frac=0.15)
    rng = np.random.default_rng(4)
    val = val * np.repeat(rng.uniform(0.5, 2.5, size=n), nnz)
    # the two-shard example: unit-free rows [2 over terms of shard A | 2 over terms of shard B] x small overlap
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    assert len(want) > 300
    dev = torch.device("cuda", 0)
    engines = [HipShardEngine(dim, theta, tr, dev, tile_rows=512) for tr in term_ranges(np.bincount(idx, minlength=dim), world)]
    for e in engines:
        e.load(rp, idx, val)
    q, c, s, _ = join_shards_local(engines, n, theta)
    assert_same_pairs(to_map(q, c, s), want, theta, band=2e-5, tol=2e-5)  # scores up to 6: fp32 sums


def test_two_shard_counterexample_of_the_unit_norm_rule(oracle):
    """|q||c| = 4, two shards with p_g = 0.4 and |q_g||c_g| = 2 each: total 0.8 >= theta = 0.8 while p_g / (|q_g||c_g|)
    = 0.2 on both shards -- found only because the rule compares with theta / (|q||c|) = 0.2"""
    import torch
    from apss.dist import HipShardEngine, join_shards_local
    dim, theta = 8, 0.79
    # q = (a, a, 0, 0 | a, a, 0, 0), c = (b, 0, b', 0 | b, 0, b', 0): per shard q_g.c_g = a b, |q_g| = a sqrt2, |c_g| = sqrt(b^2 + b'^2)
    a, b = 1.0, 0.4
    bp = np.sqrt(2.0 - b * b)
    rp = np.array([0, 4, 8], np.int64)
    idx = np.array([0, 1, 4, 5, 0, 2, 4, 6], np.int32)
    val = np.array([a, a, a, a, b, bp, b, bp], np.float64)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    assert set(want) == {(0, 1), (1, 0)} and abs(want[(0, 1)] - 0.8) < 1e-12
    dev = torch.device("cuda", 0)
    engines = [HipShardEngine(dim, theta, tr, dev, tile_rows=64) for tr in ((0, 4), (4, 8))]
    for e in engines:
        e.load(rp, idx, val)
    q, c, s, _ = join_shards_local(engines, 2, theta)
    assert_same_pairs(to_map(q, c, s), want, theta)


def test_c4_full_size_eight_term_shards_on_one_gpu():
    """BASELINE.json configs[3] (C3's N = 1M, dim = 100k, nnz = 100, theta = 0.8, index term-sharded 8 ways) at FULL size on
    one GPU: the eight term-range shard handles are built one after another, joined with join_shards_local (the
    in-process twin of the RCCL exchange), and the result is checked by the oracle-free stratified property: every
    planted pair, nothing else, every score exact, posting visits of the shards adding up to sum df^2"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "fullsize_stratified", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "fullsize_stratified.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.run_term_shards(1_000_000, 100_000, 100, 0.8, 8, seed=20242)
    assert out["missing"] == 0 and out["unexpected"] == 0 and out["planted_pairs_required"] > 50_000
    assert len(out["probe_ms_per_shard"]) == 8 and out["max_abs_score_error"] <= 1e-5
    # the exchange stays small: a shard's candidate list is of the order of the true pairs, not of the touched pairs
    assert max(out["candidates_per_shard"]) < 4 * out["result_pairs"]


@pytest.mark.parametrize("world,hook", [(4, None), (8, None), (8, "no_acc8"), (3, "window=2")])
def test_term_shards_8bit_filter_matches_oracle(oracle, monkeypatch, world, hook):
    """default tile size: a term shard whose norms and row lengths leave room runs the 8-bit filter over 65536-row tiles
    (k_probe_coarse<.., ACC8>); same pairs as the oracle, as with the hook that keeps 16-bit accumulators"""
    import torch
    if hook:
        monkeypatch.setenv("APSS_DEBUG", hook)
    from apss.dist import HipShardEngine, join_shards_local, term_ranges
    n, dim, nnz, theta = 70_000, 6000, 32, 0.7
    rp, idx, val = synth.make_vectors(n, dim, nnz, 0.0, seed=17, dup_frac=0.05)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    assert len(want) > 3000
    dev = torch.device("cuda", 0)
    engines = [HipShardEngine(dim, theta, tr, dev) for tr in term_ranges(np.bincount(idx, minlength=dim), world)]
    for e in engines:
        e.load(rp, idx, val)
    q, c, s, n_cand = join_shards_local(engines, n, theta)
    assert_same_pairs(to_map(q, c, s), want, theta)
    assert engines[0].stats["tiles"] == (2 if hook != "no_acc8" else 3)  # 65536-row tiles unless the hook forbids them
    assert sum(e.stats["posting_visits"] for e in engines) == int(synth.workload_counts(dim, rp, idx)[1])
    # the rule's own false candidates (two rows whose dominant term inside a shard coincides) dwarf the filter's: a small
    # fraction of the pairs a shard touches either way
    assert max(n_cand) < 0.05 * min(e.stats["candidate_pairs"] for e in engines)


def test_8bit_shard_filter_falls_back_when_a_query_batch_does_not_fit(oracle):
    """the store's rows fit the 8-bit sums, a later query batch does not (norm 12: even at 2^5 units per 1.0, S |q||c| would
    carry into the neighbour's byte): the handle rebuilds its index for 16-bit accumulators and answers; exchange done by
    hand as apss.dist does it"""
    import torch
    from apss.engine import ApssIndex
    n, nq, dim, nnz, theta = 40_000, 300, 5000, 24, 9.0
    rp, idx, val = synth.make_vectors(n + nq, dim, nnz, 0.0, seed=23, dup_frac=0.0)
    # queries: scaled copies of stored rows (score = 12 x cosine)
    qrows = np.arange(0, nq) * 7
    q_rp = np.arange(0, (nq + 1) * nnz, nnz, dtype=np.int64)
    q_idx = np.concatenate([idx[rp[r]:rp[r + 1]] for r in qrows])
    q_val = np.concatenate([val[rp[r]:rp[r + 1]] for r in qrows]) * 12.0
    e0 = int(rp[n])
    w = oracle.Worker(dim, theta)
    w.index_data(np.arange(n), rp[:n + 1], idx[:e0], val[:e0], build_only=True)
    want = to_map(*w.index_data(np.arange(n, n + nq), q_rp, q_idx, q_val, query_only=True))
    assert len(want) >= nq
    dev = torch.device("cuda", 0)
    shards = [ApssIndex(dim, theta, term_range=tr) for tr in ((0, 2500), (2500, 5000))]
    keys, tiles_before = [], []
    for ix in shards:
        ix.insert(np.arange(n), rp[:n + 1], idx[:e0], val[:e0])
        tiles_before.append(ix.stats()["tiles"])
        qq, cc, _ = ix.query(np.arange(n, n + nq), q_rp, q_idx, q_val)
        assert ix.stats()["tiles"] == 2  # rebuilt: 32768-row tiles
        keys.append((qq - n) * n + cc)
    assert tiles_before == [1, 1]  # 65536-row tiles while the 8-bit filter was in use
    uniq = np.unique(np.concatenate(keys))
    uq, uc = torch.from_numpy(uniq // n).to(dev).to(torch.int32), torch.from_numpy(uniq % n).to(dev).to(torch.int32)
    total = torch.zeros(uniq.size, dtype=torch.float32, device=dev)
    for ix in shards:
        part = torch.empty_like(total)
        ix.partial_scores_dev(uq, uc, part)
        total += part
    keep = (total >= theta).cpu().numpy()
    got = {(int(a) + n, int(b)): float(v) for a, b, v in zip(uniq[keep] // n, uniq[keep] % n, total.cpu().numpy()[keep])}
    assert_same_pairs(got, want, theta, band=1e-4, tol=1e-4)
    for ix in shards:
        ix.close()


def test_signed_weights_under_the_shard_rule(oracle):
    """weights of either sign on term-range shards: the two-pass filter (positive products only, shard-rule normalisation)
    instead of the general kernel; candidates + exact partial scores give the oracle's set"""
    import torch
    from apss.dist import HipShardEngine, join_shards_local, term_ranges
    rng = np.random.default_rng(733)
    n, dim, nnz, theta = 6000, 3000, 30, 0.6
    rows = []
    for i in range(n):
        if i > 20 and rng.random() < 0.15:  # a noisy copy of an earlier row, a few signs flipped
            t, v = rows[rng.integers(0, i)]
            w = v * (1 + 0.02 * rng.standard_normal(v.size))
            w[rng.random(v.size) < 0.05] *= -1
            rows.append((t, w / np.linalg.norm(w)))
            continue
        t = np.sort(rng.choice(dim, size=nnz, replace=False)).astype(np.int32)
        v = rng.standard_normal(nnz)
        rows.append((t, v / np.linalg.norm(v)))
    rp = np.arange(0, (n + 1) * nnz, nnz, dtype=np.int64)
    idx = np.concatenate([t for t, _ in rows]).astype(np.int32)
    val = np.concatenate([v for _, v in rows]).astype(np.float64)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    assert len(want) > 300
    dev = torch.device("cuda", 0)
    ranges = term_ranges(np.bincount(idx, minlength=dim), 3)
    engines = [HipShardEngine(dim, theta, tr, dev) for tr in ranges]
    for e in engines:
        e.load(rp, idx, val)
    q, c, s, _ = join_shards_local(engines, n, theta)
    assert_same_pairs(to_map(q, c, s), want, theta)
    # (a shard's survivors are its candidates; what shows the filter ran instead of the general kernel is its time:
    # profiles/r02_summary.md, "signed weights under the shard rule")
    assert sum(e.stats["posting_visits"] for e in engines) == int((np.bincount(idx, minlength=dim).astype(np.int64) ** 2).sum())


@pytest.mark.parametrize("T,D,head", [(1, 2, None), (1, 3, None), (1, 4, None), (2, 4, 64), (2, 3, None)])
def test_symmetry_across_row_ranges(oracle, T, D, head):
    """the T x D grid with the symmetry used ACROSS the row ranges (apss.dist.ShardedJoin.half_spans, round 4): a cell joins its
    own rows with themselves and meets only the rows of the ranges it owns the cross pairs with; the pairs it finds there are
    reported in both directions.  Union over the grid == the oracle's list, no pair from two cells, and the devices make about
    half of the cross-range posting visits.  Cells built one after another on one GPU, exchange in-process as apss.dist does it"""
    import torch
    from apss.dist import HipShardEngine, half_spans, term_ranges
    n, dim, nnz, theta = 6000, 3000, 30, 0.55
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0 if head else 0.0, seed=140 + 7 * D + T, dup_frac=0.1)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    assert len(want) > 200
    dev = torch.device("cuda", 0)
    df = np.bincount(idx, minlength=dim)
    head_terms = None
    if head:
        head_terms = np.lexsort((np.arange(dim), -df))[:head].astype(np.int32)
        df = df.copy()
        df[head_terms] = 0
    ranges = term_ranges(df, T)
    got, dev_visits, ref_visits = {}, 0, 0
    for dj in range(D):
        spans = half_spans(n, D, dj)
        r0, r1 = n * dj // D, n * (dj + 1) // D
        engines = [HipShardEngine(dim, theta, ranges[i], dev, tile_rows=512, head=None if head_terms is None else (head_terms, i, T, 0))
                   for i in range(T)]
        for e in engines:
            e.load(rp, idx, val, (r0, r1), query_spans=spans)
        for ph in engines[0].phases():
            keys = []
            for e in engines:
                q, c = e.phase_pairs(e.run_phase(ph))
                keys.append(q * n + c)
            uniq = torch.unique(torch.cat(keys))
            uq, uc = torch.div(uniq, n, rounding_mode="floor"), uniq % n
            total = sum(e.phase_partial(uq, uc) for e in engines)
            keep = total >= theta
            cell = to_map(uq[keep].cpu().numpy(), uc[keep].cpu().numpy(), total[keep].cpu().numpy())
            if ph == "outside":
                assert all(r0 <= c < r1 and not (r0 <= q < r1) for (q, c) in cell)
                cell.update({(c, q): v for (q, c), v in list(cell.items())})
            assert not (set(cell) & set(got)), "a pair from two cells"
            got.update(cell)
        dev_visits += sum(e.stats["device_posting_visits"] for e in engines)
        ref_visits += sum(e.stats["posting_visits"] for e in engines)
    assert_same_pairs(got, want, theta)
    full = int((np.bincount(idx, minlength=dim).astype(np.int64) ** 2).sum()) if head_terms is None else None
    if full is not None:
        assert ref_visits == full  # reference-equivalent count: every (query, term, posting) of the two-directional join
        assert dev_visits < 0.75 * full  # the devices made about half of them (small tiles: the diagonal's share is visible)
