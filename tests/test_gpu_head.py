"""GPU parity tests of the dense-head path (apss_head.hpp): the most frequent terms of a skewed (Zipfian) term
distribution are scored by a bf16 MFMA contraction instead of their posting lists, the rest by the sparse filter under
the shard rule; the union is re-scored exactly.  Checked against the CPU oracle (CommonUtils.scala:98-117 +
IndexingWorkerActor.scala:74-111 restated in oracle/): same pair set outside |score - theta| <= 1e-5, scores within 1e-5."""
import numpy as np
import pytest

from apss import synth
from helpers import assert_same_pairs, to_map

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from apss import _lib, engine
    _lib.lib()
    return engine


def _join(engine, dim, theta, rp, idx, val, **kw):
    n = len(rp) - 1
    with engine.ApssIndex(dim, theta, **kw) as ix:
        q, c, s = ix.insert_and_query(np.arange(n), rp, idx, val)
        st = ix.stats()
    return to_map(q, c, s), st


@pytest.mark.parametrize("kh", [64, 128, 256])
@pytest.mark.parametrize("n,dim,nnz,theta", [(3000, 2048, 24, 0.5), (5000, 10000, 50, 0.6), (1500, 400, 12, 0.8)])
def test_forced_head_block_matches_oracle(engine, oracle, kh, n, dim, nnz, theta):
    """small Zipf(1) batches with the block forced on: every block width, partially filled query blocks and tiles"""
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=31 + kh, dup_frac=0.1)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    assert len(want) > 100
    got, st = _join(engine, dim, theta, rp, idx, val, head_terms=kh, tile_rows=1024)
    assert st["head_terms"] == min(kh, dim) and st["head_pairs"] > 0 and st["head_flops"] > 0
    assert_same_pairs(got, want, theta)
    # the plain path on the same input: identical set, and it visits every posting the hybrid path leaves to the MFMA
    ref, st0 = _join(engine, dim, theta, rp, idx, val, head_terms=-1, tile_rows=1024)
    assert st0["head_terms"] == 0 and ref.keys() == got.keys()
    assert st["posting_visits"] < st0["posting_visits"]
    # pairs scored: between max(head, tail) and their sum; the plain path's count is exact
    assert st["candidate_pairs"] <= st0["candidate_pairs"]
    assert st0["candidate_pairs"] <= st["candidate_pairs"] + st["head_pairs"] + st0["candidate_pairs"] // 2


@pytest.mark.parametrize("kh", [512, 1024, 4096])
@pytest.mark.parametrize("n,dim,nnz,theta", [(3000, 2048, 24, 0.5), (5000, 10000, 50, 0.6)])
def test_wide_head_with_a_folded_block_matches_oracle(engine, oracle, kh, n, dim, nnz, theta):
    """heads of more than 256 terms: ONE block of 256 columns -- the 128 most frequent terms keep a column each, the others FOLD
    into the other 128 (term i adds into column 128 + i mod 128) -- whose dot product bounds the head's true partial score from
    above (non-negative weights): the cost of a 256-term head whatever the number of terms.  Same pairs as the oracle and as the plain path, fewer
    posting visits than a 256-term head; dim = 2048 with 4096 wanted: EVERY term is in the head and the join is two GEMMs"""
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=61 + kh, dup_frac=0.1)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    assert len(want) > 100
    got, st = _join(engine, dim, theta, rp, idx, val, head_terms=kh, tile_rows=1024)
    n_used = int((np.bincount(idx, minlength=dim) > 0).sum())
    assert st["head_terms"] == min(kh, n_used) and st["head_pairs"] > 0
    assert_same_pairs(got, want, theta)
    ref, st1 = _join(engine, dim, theta, rp, idx, val, head_terms=256, tile_rows=1024)
    assert ref.keys() == got.keys() and st["posting_visits"] < st1["posting_visits"]
    if kh >= dim:
        assert st["posting_visits"] == 0
    assert st["head_pairs"] >= st1["head_pairs"]  # (pairs with a positive dot in the block: sharing a head term, or colliding in a folded column)
    assert st["head_columns"] == 256 and abs(st["head_flops"] / st1["head_flops"] - 1.0) < 1e-9  # the same ONE contraction of width 256, whatever kh
    # an outside query batch (square form) and a handful of queries (GEMV form) over both blocks
    with engine.ApssIndex(dim, theta, head_terms=kh, tile_rows=1024) as ix:
        ix.insert(np.arange(n), rp, idx, val)
        for b0, b1 in ((7, 12), (100, 700)):
            sl = slice(rp[b0], rp[b1])
            gq = to_map(*ix.query(np.arange(b0, b1), rp[b0:b1 + 1] - rp[b0], idx[sl], val[sl]))
            assert_same_pairs(gq, {k: v for k, v in want.items() if b0 <= k[0] < b1}, theta)


def test_head_pairs_count_is_exact(engine):
    """`head_pairs` against an independent count: pairs sharing at least one of the block's terms (scipy boolean X X^T
    restricted to the most frequent terms), self pairs excluded"""
    import scipy.sparse as sp
    n, dim, nnz, theta, kh = 4000, 3000, 20, 0.6, 64
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=5, dup_frac=0.05)
    _, st = _join(engine, dim, theta, rp, idx, val, head_terms=kh)
    df = np.bincount(idx, minlength=dim)
    order = np.lexsort((np.arange(dim), -df))  # the library's order: df descending, term id ascending
    head = np.zeros(dim, bool)
    head[order[:kh]] = True
    X = sp.csr_matrix((np.ones(idx.size, np.float32), idx, rp), shape=(n, dim))[:, head]
    truth = (X @ X.T).nnz - int((X.getnnz(axis=1) > 0).sum())
    assert st["head_pairs"] == truth


def test_c2_size_zipf1_policy_and_forced_block(engine, oracle):
    """BASELINE.json configs[1] at full size (N=100k, dim=10k, nnz=50, Zipf(1), theta=0.5).  A head of the 256 most frequent
    terms ALONE is unselective at theta = 0.5 (it passes a quarter of a percent of all pairs on to re-scoring: the policy's
    sample declined it in round 2); the wide head's single test over 8192 terms is selective again and the policy takes it.
    Whatever the head -- none, the policy's, 256 terms forced -- the result set is the same; against the oracle on a query
    sample (the oracle needs seconds per thousand queries)"""
    cfg, rp, idx, val = synth.make_config("c2")
    n, dim, theta = cfg["n"], cfg["dim"], cfg["theta"]
    ref, st0 = _join(engine, dim, theta, rp, idx, val, head_terms=-1)
    assert st0["head_terms"] == 0, st0
    auto, st_a = _join(engine, dim, theta, rp, idx, val)
    assert st_a["head_terms"] in (0, 512, 1024, 2048, 4096, 8192, 16384, 32768) and auto.keys() == ref.keys()
    got, st = _join(engine, dim, theta, rp, idx, val, head_terms=256)
    assert st["head_terms"] == 256 and st["head_survivors"] > 100 * len(got)
    assert ref.keys() == got.keys() and len(got) > 1000
    assert max(abs(got[k] - ref[k]) for k in got) <= 2e-6
    sample = 1500
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val, 0, sample))
    assert len(want) > 10
    assert_same_pairs({k: v for k, v in got.items() if k[0] < sample}, want, theta)


def test_c3_shape_zipf1_auto_policy(engine, oracle):
    """C3's shape with Zipf(1) terms at N = 60k (theta = 0.8): here the library takes the block by itself; result set
    against the plain path on the whole batch and against the oracle on a query sample"""
    cfg, rp, idx, val = synth.make_config("c3z1", n=60_000, device="cuda")
    dim, theta = cfg["dim"], cfg["theta"]
    got, st = _join(engine, dim, theta, rp, idx, val)
    assert st["head_terms"] in (64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768) and st["head_pairs"] > 0, st
    ref, st0 = _join(engine, dim, theta, rp, idx, val, head_terms=-1)
    assert st0["head_terms"] == 0 and ref.keys() == got.keys() and len(got) > 1000
    assert st["posting_visits"] < st0["posting_visits"] // 10
    sample = 1000
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val, 0, sample))
    assert len(want) > 10
    assert_same_pairs({k: v for k, v in got.items() if k[0] < sample}, want, theta)


def test_c5_power_law_reduced(engine, oracle):
    """BASELINE.json configs[4] "power-law" at reduced N: dim = 1M (global-atomic index build, 31 term ranges), nnz = 200,
    Zipf(1), theta = 0.9, N = 40k.  The library takes the dense-head block by itself; result set against the plain path on
    the whole batch and against the oracle on a query sample; a frozen-index query batch afterwards (outside batch:
    the square form of the contraction)"""
    cfg, rp, idx, val = synth.make_config("c5z", n=40_000, device="cuda")
    dim, theta, n = cfg["dim"], cfg["theta"], cfg["n"]
    assert dim == 1_000_000 and cfg["nnz"] == 200
    with engine.ApssIndex(dim, theta) as ix:
        got = to_map(*ix.insert_and_query(np.arange(n), rp, idx, val))
        st = ix.stats()
        assert st["head_terms"] in (64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768) and st["head_pairs"] > 0.9 * n * (n - 1), st
        b0, b1 = 5000, 5600
        sl = slice(rp[b0], rp[b1])
        gq = to_map(*ix.query(np.arange(b0, b1), rp[b0:b1 + 1] - rp[b0], idx[sl], val[sl]))
    ref, st0 = _join(engine, dim, theta, rp, idx, val, head_terms=-1)
    assert st0["head_terms"] == 0 and ref.keys() == got.keys() and len(got) > 1000
    assert max(abs(got[k] - ref[k]) for k in got) <= 2e-6
    assert {k for k in gq} == {k for k in got if b0 <= k[0] < b1}
    sample = 400
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val, 0, sample))
    assert len(want) > 5
    assert_same_pairs({k: v for k, v in got.items() if k[0] < sample}, want, theta)


def test_streaming_batches_and_frozen_queries(engine, oracle):
    """the IndexData handler batch after batch (IndexingWorkerActor.scala:123-137) on a handle whose block is chosen at
    the first batch, then re-evaluated when the store has doubled; then frozen-index queries of a few vectors (the
    GEMV form of the filter) and of a few hundred (partially filled MFMA query blocks)"""
    n, dim, nnz, theta = 6000, 3000, 24, 0.55
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=77, dup_frac=0.1)
    w = oracle.Worker(dim, theta)
    cuts = [0, 2000, 2300, 4100, 6000]
    with engine.ApssIndex(dim, theta, head_terms=128, tile_rows=512) as ix:
        for b0, b1 in zip(cuts[:-1], cuts[1:]):
            sl = slice(rp[b0], rp[b1])
            args = (np.arange(b0, b1), rp[b0:b1 + 1] - rp[b0], idx[sl], val[sl])
            want = to_map(*w.index_data(*args))
            got = to_map(*ix.insert_and_query(*args))
            assert_same_pairs(got, want, theta)
            assert ix.stats()["head_terms"] == 128
        full = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
        for b0, b1 in [(10, 13), (100, 116), (1000, 1400)]:
            sl = slice(rp[b0], rp[b1])
            got = to_map(*ix.query(np.arange(b0, b1), rp[b0:b1 + 1] - rp[b0], idx[sl], val[sl]))
            assert_same_pairs(got, {k: v for k, v in full.items() if b0 <= k[0] < b1}, theta)


def test_unnormalised_rows_and_fallback(engine, oracle):
    """rows of norm != 1 (the rule divides by the full norms: no pair is lost), then a batch with a negative weight:
    the handle gives the block up for good and answers through the plain path"""
    n, dim, nnz, theta = 3000, 1500, 16, 0.4
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=9, dup_frac=0.1)
    rng = np.random.default_rng(3)
    scale = np.repeat(rng.uniform(0.4, 1.3, size=n), nnz)
    val2 = val * scale
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val2))
    got, st = _join(engine, dim, theta, rp, idx, val2, head_terms=64)
    assert st["head_terms"] == 64 and len(want) > 100
    assert_same_pairs(got, want, theta)
    val3 = val.copy()
    val3[5::7] *= -1.0
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val3))
    got, st = _join(engine, dim, theta, rp, idx, val3, head_terms=64)
    assert st["head_terms"] == 0
    assert_same_pairs(got, want, theta)
    # a non-negative store with the block, then a query with a negative weight
    with engine.ApssIndex(dim, theta, head_terms=64) as ix:
        ix.insert(np.arange(n), rp, idx, val)
        assert ix.stats()["rows"] == n
        q = to_map(*ix.query(np.arange(50), rp[:51], idx[:rp[50]], val3[:rp[50]]))
        w = oracle.Worker(dim, theta)
        w.index_data(np.arange(n), rp, idx, val)
        want = to_map(*w.index_data(np.arange(50), rp[:51], idx[:rp[50]], val3[:rp[50]], query_only=True))
        assert_same_pairs(q, want, theta)
        assert ix.stats()["head_terms"] == 0


def test_maildir_tfidf_with_a_forced_head_block(engine):
    """BASELINE config 1 -- TF-IDF vectors of the reference's own mail corpus (HashingTF 2^20; real, Zipfian term
    distribution; rows of up to 2,247 terms) -- through the dense-head path: the frequent terms in the bf16 block, the tail
    under the shard rule with queries cut into parts of 512 terms.  Same pairs as the committed fixture (oracle output)"""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "maildir_small_tfidf.npz"))
    dim, theta = int(z["dim"]), float(z["theta"])
    want = to_map(z["out_q"], z["out_c"], z["out_sim"])
    for kh in (64, 256):
        got, st = _join(engine, dim, theta, z["rowptr"], z["indices"], z["values"], head_terms=kh)
        assert st["head_terms"] == kh and st["head_pairs"] > 0
        assert_same_pairs(got, want, theta)


def test_power_law_property_without_the_oracle_reduced():
    """the oracle-free check of profiles/fullsize_powerlaw.py (BASELINE.json configs[4] at N = 10M: every reported pair's
    score against a direct float64 dot, every planted near-duplicate pair reported) at N = 150k, where it takes seconds:
    configs[4]'s dim, nnz, distribution and threshold, the library's own head policy"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "fullsize_powerlaw", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "fullsize_powerlaw.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.run(150_000, 1_000_000, 200, 0.9)
    assert out["missing"] == 0 and out["planted_pairs_required"] > 5000 and out["max_abs_score_error"] <= 1e-5
    assert out["head_terms"] >= 256 and out["head_pairs"] > 0.9 * out["pairs_n_squared"]
    assert out["posting_visits"] < 0.01 * out["posting_visits_all_sparse"]  # the head took the long posting lists


def test_int8_and_bf16_renderings_of_the_block_give_the_same_pairs(engine, oracle, monkeypatch):
    """the block's rows as INT8 rounded UP (the default: v_mfma_i32_32x32x32_i8, integer sums, no rounding bound) and as bf16
    (APSS_DEBUG=head_bf16): the same result list, on a folded head too; the INT8 filter passes hardly more than the bf16 one"""
    n, dim, nnz, theta = 6000, 4000, 40, 0.55
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=77, dup_frac=0.1)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    for kh in (100, 256, 2000):
        got8, st8 = _join(engine, dim, theta, rp, idx, val, head_terms=kh)
        monkeypatch.setenv("APSS_DEBUG", "head_bf16")
        got16, st16 = _join(engine, dim, theta, rp, idx, val, head_terms=kh)
        monkeypatch.delenv("APSS_DEBUG")
        assert st8["head_int8"] == 1 and st16["head_int8"] == 0 and st8["head_columns"] == (128 if kh <= 128 else 256)
        assert_same_pairs(got8, want, theta)
        assert got8.keys() == got16.keys()
        assert st8["head_survivors"] <= 1.5 * st16["head_survivors"] + 100
        assert st8["head_pairs"] == st16["head_pairs"]  # pairs sharing a head term: a positive integer sum iff a positive product


def test_int8_block_follows_growing_row_norms(engine, oracle):
    """un-normalised rows streamed with norms that grow from batch to batch (1, 1.6, 2.4, 3.5) and an outside query batch larger
    still: the INT8 scale S = 127 / (largest norm) shrinks for the WHOLE block and the rows packed so far are re-quantised in
    place (k_head_rescale: still upper bounds); every batch's answer equals the oracle worker's"""
    n, dim, nnz = 4000, 1500, 16
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=19, dup_frac=0.15)
    theta = 0.6
    cuts = [0, 1000, 1800, 2500, 3200]
    grow = [1.0, 1.6, 2.4, 3.5]
    w = oracle.Worker(dim, theta)
    with engine.ApssIndex(dim, theta, head_terms=300) as ix:
        for (b0, b1), f in zip(zip(cuts[:-1], cuts[1:]), grow):
            sl = slice(rp[b0], rp[b1])
            args = (np.arange(b0, b1), rp[b0:b1 + 1] - rp[b0], idx[sl], val[sl] * f)
            want = to_map(*w.index_data(*args))
            got = to_map(*ix.insert_and_query(*args))
            assert ix.stats()["head_int8"] == 1 and ix.stats()["head_terms"] == 300
            assert_same_pairs(got, want, theta, band=5e-5, tol=5e-5)  # scores up to 12: fp32 sums
        b0, b1 = 3200, 4000
        sl = slice(rp[b0], rp[b1])
        args = (np.arange(b0, b1), rp[b0:b1 + 1] - rp[b0], idx[sl], val[sl] * 4.5)
        want = to_map(*w.index_data(*args, query_only=True))
        got = to_map(*ix.query(*args))
        assert_same_pairs(got, want, theta, band=1e-4, tol=1e-4)
        from apss import _lib
        assert not (ix.stats()["downgrades"] & _lib.DOWNGRADE_HEAD) and ix.stats()["head_int8"] == 1  # (the block stayed, in INT8)
