"""The symmetric whole-store join (ProbeArgs::tri, k_mirror_survivors; include/apss.h: APSS_FLAG_NO_SYMMETRY): when the query
batch IS the indexed store, the filter kernels run only the (query tile, candidate tile) pairs on or below the diagonal and
mirror the survivors of the pairs below it.  The contract pinned here: the SAME result list as the two-directional probe
(pairs and scores, bit for bit: both directions are still re-scored exactly), the SAME statistics as the reference's
two-directional probe (posting visits = sum of df^2, candidate pairs = off-diagonal non-zeros of X X^T), and about half of
the visits on the device.  Small tiles (tile_rows) so that a few thousand rows span a dozen tiles."""
import numpy as np
import pytest

from helpers import assert_same_pairs, to_map
from test_gpu_even import _mixed_batch

pytestmark = pytest.mark.gpu


def _counts(dim, rp, idx, val):
    import scipy.sparse as sp
    n = rp.size - 1
    df = np.bincount(idx, minlength=dim).astype(np.int64)
    X = sp.csr_matrix((np.ones_like(val), idx, rp), shape=(n, dim))
    return int((df ** 2).sum()), (X @ X.T).nnz - n


@pytest.mark.parametrize("debug", ["", "no_even"])
@pytest.mark.parametrize("n,tile_rows", [(6000, 256), (5001, 512), (3000, 64)])
def test_symmetric_join_equals_the_two_directional_probe(oracle, monkeypatch, debug, n, tile_rows):
    from apss import _lib
    from apss.engine import ApssIndex
    monkeypatch.setenv("APSS_DEBUG", debug)
    theta = 0.7
    dim, rp, idx, val = _mixed_batch(n, seed=500 + n)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    assert len(want) > 300
    visits, cands = _counts(dim, rp, idx, val)
    ids = np.arange(n, dtype=np.int64)
    runs = {}
    for name, flags in (("sym", 0), ("both", _lib.FLAG_NO_SYMMETRY)):
        with ApssIndex(dim, theta, head_terms=-1, tile_rows=tile_rows, flags=flags) as ix:
            q, c, s = ix.insert_and_query(ids, rp, idx, val)
            runs[name] = (to_map(q, c, s), ix.stats())
    got, st = runs["sym"]
    both, st2 = runs["both"]
    assert st["symmetric"] == 1 and st2["symmetric"] == 0, (st, st2)
    assert_same_pairs(got, want, theta)
    assert got == both  # the same pairs with the same fp32 scores: each direction is re-scored by the same exact pass
    for a in (st, st2):
        assert a["posting_visits"] == visits and a["candidate_pairs"] == cands, (a, visits, cands)
    assert st2["device_posting_visits"] == visits
    tiles = -(-n // (2 * tile_rows))  # (the filter's tiles hold 2 x tile_rows rows)
    assert tiles >= 5
    # on and below the diagonal: (tiles + 1) / (2 tiles) of the tile pairs, give or take the tiles' unequal content
    assert 0.45 * visits < st["device_posting_visits"] < ((tiles + 1) / (2 * tiles) + 0.08) * visits, (st["device_posting_visits"], visits, tiles)
    assert st["probe_kernel"] == st2["probe_kernel"]


def test_symmetric_join_with_signed_weights(oracle):
    """signed weights take the filter's positive-products sums (an upper bound of the score in either direction): a true
    pair survives in the direction that is run, so the mirrored half must agree with the oracle too"""
    from apss.engine import ApssIndex
    n, theta = 4000, 0.6
    dim, rp, idx, val = _mixed_batch(n, seed=77)
    rng = np.random.default_rng(5)
    val = val * np.where(rng.random(val.size) < 0.15, -1.0, 1.0)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    assert len(want) > 100
    with ApssIndex(dim, theta, head_terms=-1, tile_rows=128) as ix:
        got = to_map(*ix.insert_and_query(np.arange(n, dtype=np.int64), rp, idx, val))
        st = ix.stats()
    assert st["symmetric"] == 1 and st["probe_kernel"].startswith("k_probe_coarse<"), st
    assert_same_pairs(got, want, theta)


def test_symmetry_is_for_whole_store_joins_only(oracle):
    """a second batch probes the store it extends: its rows against the OLDER rows are found in one direction only (the
    reference reports a pair once, by the later vector), so that call must not mirror anything; apss_self_join afterwards is
    a whole-store join again"""
    from apss.engine import ApssIndex
    n, theta = 6000, 0.7
    dim, rp, idx, val = _mixed_batch(n, seed=91)
    half = 3000
    e = int(rp[half])
    ids = np.arange(n, dtype=np.int64)
    w = oracle.Worker(dim, theta)
    want = to_map(*w.index_data(ids[:half], rp[:half + 1], idx[:e], val[:e]))
    want.update(to_map(*w.index_data(ids[half:], rp[half:] - e, idx[e:], val[e:])))
    with ApssIndex(dim, theta, head_terms=-1, tile_rows=128) as ix:
        q1, c1, s1 = ix.insert_and_query(ids[:half], rp[:half + 1], idx[:e], val[:e])
        assert ix.stats()["symmetric"] == 1
        q2, c2, s2 = ix.insert_and_query(ids[half:], rp[half:] - e, idx[e:], val[e:])
        assert ix.stats()["symmetric"] == 0
        got = to_map(np.concatenate([q1, q2]), np.concatenate([c1, c2]), np.concatenate([s1, s2]))
        assert_same_pairs(got, want, theta)
        qs, cs, ss = ix.self_join()
        st = ix.stats()
    assert st["symmetric"] == 1, st
    assert_same_pairs(to_map(qs, cs, ss), to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val)), theta)


@pytest.mark.parametrize("head", [False, True])
def test_symmetric_join_on_term_shards(oracle, head):
    """shard handles (the shard rule's per-row ratios are symmetric in the pair) with and without a dense-head block: the
    mirrored survivors are candidates like any other and phase 2 scores them exactly"""
    import torch
    from apss.dist import HipShardEngine, join_shards_local, term_ranges
    n, theta = 6000, 0.7
    dim, rp, idx, val = _mixed_batch(n, seed=123)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    dev = torch.device("cuda", 0)
    df = np.bincount(idx, minlength=dim).astype(np.int64)
    terms = np.argsort(-df, kind="stable")[:64].astype(np.int32) if head else np.zeros(0, np.int32)
    df_tail = df.copy()
    df_tail[terms] = 0
    ranges = term_ranges(df_tail, 3)
    engines = [HipShardEngine(dim, theta, tr, dev, tile_rows=128, head=(terms, i, 3, 0) if head else None) for i, tr in enumerate(ranges)]
    for e in engines:
        e.load(rp, idx, val)
    q, c, s, _ = join_shards_local(engines, n, theta)
    assert_same_pairs(to_map(q, c, s), want, theta)
    assert all(e.stats["symmetric"] == 1 for e in engines), [e.stats["symmetric"] for e in engines]
    assert sum(e.stats["posting_visits"] for e in engines) == int((df_tail ** 2).sum())
    assert sum(e.stats["device_posting_visits"] for e in engines) < 0.62 * int((df_tail ** 2).sum())


def test_symmetric_join_with_a_dense_head_block(oracle):
    """plain handle with a forced head block: the head contraction mirrors its own triangle, the sparse tail's survivors are
    mirrored by k_mirror_survivors, and a pair found by both appears once"""
    from apss.engine import ApssIndex
    n, theta = 6000, 0.7
    dim, rp, idx, val = _mixed_batch(n, seed=321)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    with ApssIndex(dim, theta, head_terms=64, tile_rows=128) as ix:
        q, c, s = ix.insert_and_query(np.arange(n, dtype=np.int64), rp, idx, val)
        st = ix.stats()
    assert st["head_terms"] == 64 and st["symmetric"] == 1, st
    assert q.size == len(to_map(q, c, s))  # no pair twice
    assert_same_pairs(to_map(q, c, s), want, theta)


@pytest.mark.parametrize("head", [-1, 64])
def test_candidate_list_overflow_on_a_symmetric_join_regrows_once(oracle, monkeypatch, head):
    """the candidate list overflows in the FILTER stage of a symmetric run (APSS_DEBUG=res_cap=500: small tests never exceed the
    library's 2^20 entries): the mirror kernel has then copied only what the list held and the counter under-reports -- the
    regrowth is sized from the filter's own count (twice over: the mirrored half) plus the dense half's survivors, so the run
    is repeated ONCE, not once per stage; same list as a run that never overflowed, with and without a dense-head block"""
    from apss.engine import ApssIndex
    theta, n = 0.7, 6000
    dim, rp, idx, val = _mixed_batch(n, seed=811)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    assert len(want) > 1500
    ids = np.arange(n, dtype=np.int64)
    with ApssIndex(dim, theta, head_terms=head, tile_rows=256) as ix:
        ref = to_map(*ix.insert_and_query(ids, rp, idx, val))
        st0 = ix.stats()
    monkeypatch.setenv("APSS_DEBUG", "res_cap=500")
    with ApssIndex(dim, theta, head_terms=head, tile_rows=256) as ix:
        got = to_map(*ix.insert_and_query(ids, rp, idx, val))
        st = ix.stats()
    assert st["symmetric"] == 1 and st0["symmetric"] == 1 and st["head_terms"] == max(head, 0)
    assert got == ref and st["filter_survivors"] == st0["filter_survivors"] > 500
    assert st["probe_launches"] == 2 * st0["probe_launches"]  # one overflowing attempt + one that fits
    assert_same_pairs(got, want, theta)
