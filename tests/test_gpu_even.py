"""The thin-round filter kernel (k_probe_even, csrc/apss_even.hpp) against the oracle, with batches built so that every
kind of round it distinguishes occurs: rounds inside the register window, rounds whose chunks spill past the window
(read straight from the strip, whole-tile clear), rounds flagged at staging because they do not fit the strips (swept
straight from the index) and rounds with long segments (swept by the whole workgroup).  `thin_launches` in the stats says
which kernel ran; APSS_DEBUG=no_even is the same join through k_probe_coarse."""
import numpy as np
import pytest

from helpers import assert_same_pairs, to_map

pytestmark = pytest.mark.gpu


def _mixed_batch(n, seed):
    """mostly 4-term rows over a large vocabulary (thin rounds), plus a few per cent of 30-term rows over small popular
    vocabularies whose (tile, term) segments run to 1, 4 and 10 chunks, plus one term long enough to be a 'long segment';
    a tenth of the rows are noisy copies of earlier rows, so that every class has pairs above the threshold"""
    rng = np.random.default_rng(seed)
    dim = 6000
    vocab = {"rare": (0, 5000), "one": (5150, 5400), "four": (5400, 5990), "ten": (5000, 5150)}
    rows = []
    for i in range(n):
        u = rng.random()
        if i > 50 and u < 0.10:
            src = rows[rng.integers(0, i)]
            t, w = src[0].copy(), src[1] * (1.0 + 0.05 * rng.standard_normal(src[1].size))
            rows.append((t, np.abs(w) + 1e-3))
            continue
        if u < 0.86:
            lo, hi, k = *vocab["rare"], 4
        elif u < 0.90:
            lo, hi, k = *vocab["ten"], 30
        elif u < 0.95:
            lo, hi, k = *vocab["four"], 30
        else:
            lo, hi, k = *vocab["one"], 12
        t = np.sort(rng.choice(np.arange(lo, hi), size=k, replace=False)).astype(np.int32)
        w = np.abs(rng.standard_normal(k)) + 0.1
        if rng.random() < 0.03:  # the long term: df ~ n / 33
            t = np.append(t, np.int32(5999))
            w = np.append(w, 0.5)
        rows.append((t, w))
    rp = np.zeros(n + 1, np.int64)
    for i, (t, _) in enumerate(rows):
        rp[i + 1] = rp[i] + t.size
    idx = np.concatenate([t for t, _ in rows]).astype(np.int32)
    val = np.concatenate([w / np.linalg.norm(w) for _, w in rows]).astype(np.float64)
    return dim, rp, idx, val


@pytest.mark.parametrize("debug,thin", [("", True), ("no_even", False)])
def test_every_kind_of_round_of_the_thin_kernel(oracle, monkeypatch, debug, thin):
    from apss.engine import ApssIndex
    monkeypatch.setenv("APSS_DEBUG", debug)
    n, theta = 20000, 0.7
    dim, rp, idx, val = _mixed_batch(n, seed=404)
    # the classes the kernel distinguishes are really there: chunks of a 30-term row of each popular vocabulary
    df = np.bincount(idx, minlength=dim)
    assert 280 > df[5000:5150].mean() > 120 and 80 > df[5400:5990].mean() > 30 and df[5999] > 256
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    assert len(want) > 1000
    with ApssIndex(dim, theta, head_terms=-1) as ix:  # (no dense-head block: the sparse filter takes every term)
        q, c, s = ix.insert_and_query(np.arange(n, dtype=np.int64), rp, idx, val)
        st = ix.stats()
    assert_same_pairs(to_map(q, c, s), want, theta)
    assert (st["thin_launches"] > 0) == thin, sorted(st.items())
    assert st["posting_visits"] == int((df.astype(np.int64) ** 2).sum())
    # the benchmark's unit against an independent count: pairs sharing a term = structural non-zeros of X X^T off the diagonal
    import scipy.sparse as sp
    X = sp.csr_matrix((np.ones_like(val), idx, rp), shape=(n, dim))
    assert st["candidate_pairs"] == (X @ X.T).nnz - n


def test_thin_kernel_under_the_shard_rule(oracle):
    """the same batch cut into 4 term ranges (shard rule, 8-bit accumulators where the norms allow): the thin kernel on every
    shard, joined as apss.dist joins them"""
    import torch
    from apss.dist import HipShardEngine, join_shards_local, term_ranges
    n, theta = 20000, 0.7
    dim, rp, idx, val = _mixed_batch(n, seed=405)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    dev = torch.device("cuda", 0)
    ranges = term_ranges(np.bincount(idx, minlength=dim), 4)
    engines = [HipShardEngine(dim, theta, tr, dev) for tr in ranges]
    for e in engines:
        e.load(rp, idx, val)
    q, c, s, _ = join_shards_local(engines, n, theta)
    assert_same_pairs(to_map(q, c, s), want, theta)
    assert sum(e.stats["thin_launches"] for e in engines) > 0, [e.stats["thin_launches"] for e in engines]
    assert sum(e.stats["posting_visits"] for e in engines) == int((np.bincount(idx, minlength=dim).astype(np.int64) ** 2).sum())


@pytest.mark.parametrize("n,chunks", [(2047, 2047), (2047, 1024), (2048, 683), (1000, 334), (777, 1), (2, 2), (1, 1)])
def test_chunk_shapes_pin_the_barrier_pairing(oracle, monkeypatch, n, chunks):
    """k_probe_even's staging and adding waves run DIFFERENT loops around the same two barriers per round (apss_even.hpp): the
    counts match by construction -- one iteration per query of the workgroup's chunk on both sides, the adding waves' loop
    unrolled by two with an exit after an odd last round.  Pinned here on the shapes where a mismatch would hang or corrupt:
    chunks of ONE query (v1 - v0 = 1: the adders break after their first round), of two and three (even / odd), a short
    last chunk, one chunk holding everything, and batches of one and two rows"""
    from apss import synth
    from apss.engine import ApssIndex
    monkeypatch.setenv("APSS_DEBUG", "chunks=%d" % chunks)
    dim, nnz, theta = 900, 10, 0.6
    rp, idx, val = synth.make_vectors(max(n, 2), dim, nnz, 0.0, seed=31 + n, dup_frac=0.2)
    rp, idx, val = rp[:n + 1], idx[:rp[n]], val[:rp[n]]
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    with ApssIndex(dim, theta, head_terms=-1) as ix:
        got = to_map(*ix.insert_and_query(np.arange(n, dtype=np.int64), rp, idx, val))
        st = ix.stats()
    assert_same_pairs(got, want, theta)
    if n > 100:
        assert st["probe_kernel"].startswith("k_probe_even<"), st["probe_kernel"]
        assert len(want) > 20
