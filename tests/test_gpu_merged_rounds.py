"""Merged rounds of a term shard's thin-round filter (apss_even.hpp, k_probe_even_merged): M = 2 | 4 neighbouring query rows
share a round and a candidate's accumulator holds the SUM of their filter sums -- an upper bound of each, so the filter
stays sound; a crossing becomes M survivors and the exact partial scores prune as ever.  Checked here: the joined result is
the oracle's list whether or not rounds are merged, on whole-store joins (the diagonal workgroups run their rows one per
round: a row meets itself there), on outside batches (symmetry across row ranges) and with 16-bit and 8-bit accumulators;
neighbouring rows that are near-duplicates of EACH OTHER (same round) and of their neighbours' neighbours are found."""
import numpy as np
import pytest

from apss import synth
from helpers import assert_same_pairs, to_map

pytestmark = pytest.mark.gpu


def _data(n, dim, nnz, seed):
    rp, idx, val = synth.make_vectors(n, dim, nnz, 0.0, seed=seed, dup_frac=0.05)
    idx, val = idx.reshape(n, nnz).copy(), val.reshape(n, nnz).copy()
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    # near-duplicates INSIDE a round (rows 4k, 4k + 1), across two rounds of two (4k + 1, 4k + 2) and across rounds of four (4k + 3, 4k + 4)
    for k in rng.choice(n // 4 - 2, size=n // 40, replace=False):
        a, b = 4 * k + int(rng.integers(0, 4)), 0
        b = a + 1
        idx[b], val[b] = idx[a], val[a] * rng.uniform(0.95, 1.05, size=nnz)
        val[b] /= np.linalg.norm(val[b])
    return rp, idx.reshape(-1), val.reshape(-1)


@pytest.mark.parametrize("world,tile_rows,merge", [(4, 1024, None), (4, 1024, "merge=1"), (4, 1024, "merge=0"), (8, 0, None), (2, 2048, "merge=2,merge_single=7,merge_u=7"),
                                                      (4, 1024, "res_cap=256"), (4, 1024, "res_cap=256,no_merge_prune")])
def test_merged_rounds_match_oracle(oracle, monkeypatch, world, tile_rows, merge):
    import torch
    if merge:
        monkeypatch.setenv("APSS_DEBUG", merge)
    from apss.dist import HipShardEngine, join_shards_local, term_ranges
    # rows of >= 8 terms per shard over thousands of terms per range: where the library merges (apss_hip.hip, probe())
    n, dim, nnz, theta = (20_000, 60_000, 64, 0.7) if tile_rows else (70_000, 300_000, 96, 0.75)
    rp, idx, val = _data(n, dim, nnz, seed=311 + world)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    assert len(want) > 1000
    dev = torch.device("cuda", 0)
    engines = [HipShardEngine(dim, theta, tr, dev, tile_rows=tile_rows) for tr in term_ranges(np.bincount(idx, minlength=dim), world)]
    for e in engines:
        e.load(rp, idx, val)
    q, c, s, n_cand = join_shards_local(engines, n, theta)
    assert_same_pairs(to_map(q, c, s), want, theta)
    qpr = [e.stats["queries_per_round"] for e in engines]
    if merge == "merge=0":
        assert qpr == [1] * world and all("merged" not in e.stats["probe_kernel"] for e in engines)
    else:
        assert all(m in (2, 4) for m in qpr), qpr
        assert all(e.stats["probe_kernel"].startswith("k_probe_even_merged<") for e in engines), engines[0].stats["probe_kernel"]
        if merge == "merge=1":
            assert qpr == [2] * world
    assert engines[0].stats["symmetric"] == 1 and engines[0].stats["tiles"] > 1
    # the reference-equivalent posting visits do not depend on how rounds are formed
    assert sum(e.stats["posting_visits"] for e in engines) == int((np.bincount(idx, minlength=dim).astype(np.int64) ** 2).sum())
    # a crossing stands for every row of its round: at most M times the unmerged candidates, still far below the touched pairs
    assert max(n_cand) < 0.05 * min(e.stats["candidate_pairs"] for e in engines)


def test_merged_rounds_on_outside_batches(oracle):
    """T x D = 4 x 2 with the symmetry across row ranges: the `outside` phase is a query batch that is NOT stored in the handle
    (no diagonal: every workgroup merges; self-exclusion by external id happens in k_expand_merged)"""
    import torch
    from apss.dist import HipShardEngine, half_spans, term_ranges
    n, dim, nnz, theta, T, D = 16_000, 60_000, 64, 0.7, 4, 2
    rp, idx, val = _data(n, dim, nnz, seed=977)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    assert len(want) > 800
    dev = torch.device("cuda", 0)
    ranges = term_ranges(np.bincount(idx, minlength=dim), T)
    got, merged_outside = {}, 0
    for dj in range(D):
        r0, r1 = n * dj // D, n * (dj + 1) // D
        engines = [HipShardEngine(dim, theta, ranges[i], dev, tile_rows=1024) for i in range(T)]
        for e in engines:
            e.load(rp, idx, val, (r0, r1), query_spans=half_spans(n, D, dj))
        for ph in engines[0].phases():
            keys = []
            for e in engines:
                q, c = e.phase_pairs(e.run_phase(ph))
                keys.append(q * n + c)
                if ph == "outside" and e.o_rows.numel():
                    merged_outside += int(e.ix.stats()["queries_per_round"] > 1)
            uniq = torch.unique(torch.cat(keys))
            uq, uc = torch.div(uniq, n, rounding_mode="floor"), uniq % n
            total = sum(e.phase_partial(uq, uc) for e in engines)
            keep = total >= theta
            cell = to_map(uq[keep].cpu().numpy(), uc[keep].cpu().numpy(), total[keep].cpu().numpy())
            if ph == "outside":
                cell.update({(c, q): v for (q, c), v in list(cell.items())})
            assert not (set(cell) & set(got)), "a pair from two cells"
            got.update(cell)
    assert_same_pairs(got, want, theta)
    assert merged_outside == T  # (D = 2: one of the two cells owns the cross pairs)


def test_rows_of_few_terms_are_not_merged(oracle):
    """three or four terms per row and shard over a few hundred terms: chance hits from different rows of a round would cross the
    threshold together (measured: 8 x the candidates) -- the library does not merge there"""
    import torch
    from apss.dist import HipShardEngine, join_shards_local, term_ranges
    n, dim, nnz, theta = 4000, 2000, 30, 0.8
    rp, idx, val = synth.make_vectors(n, dim, nnz, 0.0, seed=91, dup_frac=0.1)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    dev = torch.device("cuda", 0)
    engines = [HipShardEngine(dim, theta, tr, dev, tile_rows=1024) for tr in term_ranges(np.bincount(idx, minlength=dim), 8)]
    for e in engines:
        e.load(rp, idx, val)
    q, c, s, _ = join_shards_local(engines, n, theta)
    assert_same_pairs(to_map(q, c, s), want, theta)
    assert [e.stats["queries_per_round"] for e in engines] == [1] * 8


@pytest.mark.parametrize("T,zipf", [(2, 0.0), (4, 0.0), (4, 1.0)])
def test_merged_rounds_on_term_shards_with_a_dense_head_block(oracle, T, zipf):
    """term shards that share a dense-head block (64 terms): the shards' TAIL rounds are the thin ones (power-law C5's 8 x 1
    shard at N = 2M: tail filter 39.5 -> 24.0 ms with two rows per round); same list as the oracle.  On the small Zipf(1)
    case the tail keeps long segments and the all-wave kernel: nothing is merged there, the list is the same"""
    import torch
    from apss.dist import HipShardEngine, join_shards_local, term_ranges
    n, dim, nnz, theta = 24_000, 80_000, 80, 0.7
    rp, idx, val = synth.make_vectors(n, dim, nnz, zipf, seed=4242 + T, dup_frac=0.08)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    assert len(want) > 1000
    dev = torch.device("cuda", 0)
    df = np.bincount(idx, minlength=dim)
    head_terms = np.lexsort((np.arange(dim), -df))[:64].astype(np.int32)
    df = df.copy()
    df[head_terms] = 0
    engines = [HipShardEngine(dim, theta, tr, dev, tile_rows=2048, head=(head_terms, i, T, 0)) for i, tr in enumerate(term_ranges(df, T))]
    for e in engines:
        e.load(rp, idx, val)
    q, c, s, _ = join_shards_local(engines, n, theta)
    assert_same_pairs(to_map(q, c, s), want, theta)
    assert all(e.stats["head_terms"] == 64 for e in engines)
    assert [e.stats["queries_per_round"] for e in engines] == [2 if zipf == 0.0 else 1] * T, [e.stats["probe_kernel"] for e in engines]


def test_merged_rounds_hand_over_no_more_candidates_than_unmerged(oracle, monkeypatch):
    """a crossing of a merged round stands for both of its rows; k_shard_prune keeps the pairs that pass the shard rule on their EXACT
    partial score, so the list a shard hands to the exchange is no longer than the unmerged filter's (whose sums are upper
    bounds) -- and without that test (no_merge_prune) it is up to twice as long; the joined result is the same in all three"""
    import torch
    from apss.dist import HipShardEngine, join_shards_local, term_ranges
    n, dim, nnz, theta, world = 20_000, 60_000, 64, 0.7, 4
    rp, idx, val = _data(n, dim, nnz, seed=515)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    dev = torch.device("cuda", 0)
    ranges = term_ranges(np.bincount(idx, minlength=dim), world)
    cands = {}
    for hook in ("merge=0", "merge=1", "merge=1,no_merge_prune"):
        monkeypatch.setenv("APSS_DEBUG", hook)  # read when a handle is created
        engines = [HipShardEngine(dim, theta, tr, dev, tile_rows=1024) for tr in ranges]
        for e in engines:
            e.load(rp, idx, val)
        q, c, s, n_cand = join_shards_local(engines, n, theta)
        assert_same_pairs(to_map(q, c, s), want, theta)
        assert [e.stats["queries_per_round"] for e in engines] == [1 if hook == "merge=0" else 2] * world
        cands[hook] = sum(n_cand)
    assert cands["merge=1"] <= cands["merge=0"] < cands["merge=1,no_merge_prune"] <= 2 * cands["merge=0"] + 100, cands
