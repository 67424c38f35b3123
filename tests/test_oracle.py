"""Pins the CPU oracle (oracle/apss_oracle.c): hand-derived KATs of the reference semantics, an
independent scipy float64 cross-check, and properties (partition invariance, streaming subset, symmetry).
The reference has no tests or golden vectors of its own -- parity is otherwise unpinned."""
import json
import os

import numpy as np
import pytest

from apss import synth
from helpers import assert_same_pairs, scipy_pairs, to_map

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _kat():
    with open(os.path.join(GOLDEN, "kat_section33.json")) as f:
        return json.load(f)


def _csr_of(vecs, names):
    rowptr, idx, val, ids = [0], [], [], []
    for nm in names:
        v = vecs[nm]
        ks = sorted(int(k) for k in v)
        idx += ks
        val += [v[str(k)] for k in ks]
        rowptr.append(len(idx))
        ids.append(int(nm[1:]))
    return np.array(ids), np.array(rowptr), np.array(idx, np.int32), np.array(val)


def _local_even(rowptr, idx):
    lptr, ld = [0], []
    for i in range(len(rowptr) - 1):
        ld += [d for d in idx[rowptr[i]:rowptr[i + 1]] if d % 2 == 0]
        lptr.append(len(ld))
    return np.array(lptr), np.array(ld, np.int32)


def _expect(d):
    return {(int(q[1:]), int(c[1:])): s for q, m in d.items() for c, s in m.items()}


@pytest.mark.parametrize("mode_name", ["as_written", "intended"])
def test_kat_single_batch_even_dim_worker(oracle, mode_name):
    k = _kat()
    mode = oracle.MODE_AS_WRITTEN if mode_name == "as_written" else oracle.MODE_INTENDED
    names = ["v1", "v2", "v3", "v5"]  # v4 lives on the odd-dim worker
    ids, rp, idx, val = _csr_of(k["vectors"], names)
    w = oracle.Worker(k["dim"], k["theta"], mode)
    q, c, s = w.index_data(ids, rp, idx, val, local=_local_even(rp, idx))
    got = to_map(q, c, s)
    want = _expect(k["single_batch"][mode_name])
    assert set(got) == set(want)
    for key in want:
        assert got[key] == pytest.approx(want[key], abs=1e-12)


def test_kat_streaming_intended(oracle):
    k = _kat()
    w = oracle.Worker(k["dim"], k["theta"], oracle.MODE_INTENDED)
    for names, key in ((["v1", "v3"], "batch1"), (["v2", "v5"], "batch2")):
        ids, rp, idx, val = _csr_of(k["vectors"], names)
        q, c, s = w.index_data(ids, rp, idx, val, local=_local_even(rp, idx))
        got = to_map(q, c, s)
        want = _expect(k["streaming_intended"][key])
        assert set(got) == set(want), key
        for kk in want:
            assert got[kk] == pytest.approx(want[kk], abs=1e-12)


def test_as_written_single_dim_query_returns_nothing(oracle):
    # Q1: a query with one local dim never scores anything
    w = oracle.Worker(4, 0.1, oracle.MODE_AS_WRITTEN)
    q, c, s = w.index_data([1, 2], [0, 1, 2], [3, 3], [1.0, 1.0])
    assert q.size == 0
    w2 = oracle.Worker(4, 0.1, oracle.MODE_INTENDED)
    q, c, s = w2.index_data([1, 2], [0, 1, 2], [3, 3], [1.0, 1.0])
    assert to_map(q, c, s) == {(1, 2): 1.0, (2, 1): 1.0}


def test_query_only_unseen_dim(oracle):
    # Q8: frozen index, unseen dim -> NoSuchElementException as written, empty posting list intended
    w = oracle.Worker(8, 0.5, oracle.MODE_AS_WRITTEN)
    w.index_data([1], [0, 2], [0, 1], [0.6, 0.8])
    with pytest.raises(KeyError):
        w.index_data([2], [0, 2], [1, 5], [0.6, 0.8], query_only=True)
    w = oracle.Worker(8, 0.5, oracle.MODE_INTENDED)
    w.index_data([1], [0, 2], [0, 1], [0.6, 0.8])
    q, c, s = w.index_data([2], [0, 2], [1, 5], [1.0, 0.0], query_only=True)
    assert to_map(q, c, s) == {(2, 1): 0.8}
    assert len(w) == 1  # not indexed


def test_self_exclusion_and_dedup_are_by_id(oracle):
    # Q5: the same id stored twice is excluded as "self" both times; two ids with equal vectors are not
    w = oracle.Worker(4, 0.5, oracle.MODE_INTENDED)
    q, c, s = w.index_data([7, 7, 8], [0, 1, 2, 3], [2, 2, 2], [1.0, 1.0, 1.0])
    assert to_map(q, c, s) == {(7, 8): 1.0, (8, 7): 1.0}


def test_threshold_is_inclusive(oracle):
    w = oracle.Worker(4, 0.5, oracle.MODE_INTENDED)
    q, c, s = w.index_data([1, 2], [0, 1, 2], [0, 0], [1.0, 0.5])
    assert to_map(q, c, s) == {(1, 2): 0.5, (2, 1): 0.5}  # sim == theta kept (IWA:93)


def test_calculate_similarity(oracle):
    assert oracle.calculate_similarity(8, [0, 2, 5], [1.0, 2.0, 3.0], 8, [2, 5, 7], [0.5, 2.0, 9.0]) == 7.0
    assert np.isnan(oracle.calculate_similarity(8, [0], [1.0], 9, [0], [1.0]))  # require(size1 == size2)
    assert oracle.calculate_similarity(8, [], [], 8, [1], [1.0]) == 0.0


def test_normalize_prune_admission(oracle):
    rp = np.array([0, 2, 5])
    v = np.array([3.0, 4.0, 1.0, 2.0, 2.0])
    nv = oracle.l2_normalize(rp, v)
    assert np.allclose(nv, [0.6, 0.8, 1 / 3, 2 / 3, 2 / 3], atol=1e-15)
    orp, oi, ov = oracle.value_prune(rp, [0, 3, 1, 2, 9], nv, 0.6)  # strict >
    assert orp.tolist() == [0, 1, 3] and oi.tolist() == [3, 2, 9]
    assert np.allclose(ov, [0.8, 2 / 3, 2 / 3])
    keep = oracle.admission(rp, nv, 1.5)
    assert keep.tolist() == [False, True]  # sums 1.4, 1.666


def test_text_format_roundtrip(oracle):
    k = _kat()["text_format"]
    size, idx, val = oracle.parse_sparse_vector(k["text"])
    assert size == k["size"] and idx.tolist() == k["indices"] and val.tolist() == k["values"]
    txt = oracle.print_sparse_vector(size, idx, val)
    s2, i2, v2 = oracle.parse_sparse_vector(txt)
    assert (s2, i2.tolist(), v2.tolist()) == (size, idx.tolist(), val.tolist())
    rng = np.random.default_rng(5)
    v = rng.standard_normal(7)
    s3, i3, v3 = oracle.parse_sparse_vector(oracle.print_sparse_vector(100, np.arange(7) * 3, v))
    assert s3 == 100 and (v3 == v).all()  # bit-exact round trip
    with pytest.raises(ValueError):
        oracle.parse_sparse_vector("(3,[0,1])")
    with pytest.raises(ValueError):
        oracle.parse_sparse_vector("(3,[0,1],[1.0])")


@pytest.mark.parametrize("zipf,theta", [(0.0, 0.3), (1.0, 0.5), (1.0, 0.8)])
def test_intended_matches_scipy(oracle, zipf, theta):
    n, dim, nnz = 1200, 512, 16
    rp, idx, val = synth.make_vectors(n, dim, nnz, zipf, seed=11, dup_frac=0.1)
    w = oracle.Worker(dim, theta, oracle.MODE_INTENDED)
    q, c, s = w.index_data(np.arange(n), rp, idx, val)
    got = to_map(q, c, s)
    want = scipy_pairs(dim, theta, rp, idx, val)
    assert len(want) > 50
    assert_same_pairs(got, want, theta, band=1e-12, tol=1e-12)
    # and the dense-accumulator variant used for bigger parity cases agrees too
    q2, c2, s2 = oracle.selfjoin_pairs(dim, theta, rp, idx, val)
    assert_same_pairs(to_map(q2, c2, s2), got, theta, band=1e-12, tol=1e-12)
    # symmetry within one batch (Q4)
    assert all((c_, q_) in got for (q_, c_) in got)


@pytest.mark.parametrize("shards,entries,workers", [(1, 1, 1), (3, 2, 2), (10, 10, 10), (4, 1, 7)])
def test_partition_invariance_intended(oracle, shards, entries, workers):
    n, dim, nnz, theta = 400, 128, 8, 0.4
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=3, dup_frac=0.1)
    single = oracle.Worker(dim, theta, oracle.MODE_INTENDED)
    want = to_map(*single.index_data(np.arange(n), rp, idx, val))
    cl = oracle.Cluster(dim, theta, oracle.MODE_INTENDED, shards, entries, workers)
    got = to_map(*cl.flush(np.arange(n), rp, idx, val))
    assert_same_pairs(got, want, theta, band=0, tol=1e-15)
    if shards * workers > 1:
        # Q6: the reference scores the same pair on every worker that shares a dim
        assert cl.sim_calls() > 0


def test_streaming_is_subset_of_single_batch(oracle):
    n, dim, nnz, theta = 600, 128, 8, 0.4
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=4, dup_frac=0.1)
    whole = to_map(*oracle.Worker(dim, theta).index_data(np.arange(n), rp, idx, val))
    w = oracle.Worker(dim, theta)
    stream = {}
    for b0 in range(0, n, 150):
        b1 = b0 + 150
        sl = slice(rp[b0], rp[b1])
        stream.update(to_map(*w.index_data(np.arange(b0, b1), rp[b0:b1 + 1] - rp[b0], idx[sl], val[sl])))
    assert set(stream) <= set(whole)
    # a later batch's vector finds every earlier one; earlier ones are never re-queried (Q4)
    assert all(q >= (c // 150) * 150 for (q, c) in stream)
    assert {k for k in whole if k[0] >= (k[1] // 150) * 150} == set(stream)


def test_cpu_baseline_variants_agree(oracle):
    n, dim, nnz, theta = 3000, 1000, 20, 0.5
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=9)
    a = oracle.selfjoin_sample(0, dim, theta, rp, idx, val, 0, 500, 2)
    b = oracle.selfjoin_sample(1, dim, theta, rp, idx, val, 0, 500, 3)
    assert a["pairs"] == b["pairs"] and a["cand_pairs"] == b["cand_pairs"] and a["visits"] == b["visits"]
    want = [k for k in to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val, 0, 500))]
    assert a["pairs"] == len(want)


def test_golden_fixtures_match_oracle(oracle):
    """The committed fixtures (tests/golden/*.npz, made by make_golden.py) are what the GPU tests compare to."""
    for name in sorted(os.listdir(GOLDEN)):
        if not name.endswith(".npz"):
            continue
        z = np.load(os.path.join(GOLDEN, name))
        if "values" not in z.files:
            continue  # (maildir_full_counts.npz holds term counts: tests/test_formats.py checks it against the oracle)
        dim, theta = int(z["dim"]), float(z["theta"])
        n = len(z["rowptr"]) - 1
        w = oracle.Worker(dim, theta, oracle.MODE_INTENDED)
        got = to_map(*w.index_data(np.arange(n), z["rowptr"], z["indices"], z["values"]))
        want = to_map(z["out_q"], z["out_c"], z["out_sim"])
        assert_same_pairs(got, want, theta, band=0, tol=1e-15)


def test_cpu_baseline_term_modulo_workers_variant():
    """variant 2 of the CPU baseline = the reference's own parallelism (T workers owning dim % T, every worker sees every
    query and re-scores every pair it reaches, EntryProxyActor.scala:41-46 + IWA:92): the same posting visits as the
    query-range variant, at least as many scorings (a pair sharing dims of k workers is scored k times), one worker = no
    duplicates"""
    from oracle import oracle
    from apss import synth
    n, dim = 3000, 800
    rp, idx, val = synth.make_vectors(n, dim, 12, 0.0, seed=4)
    a = oracle.selfjoin_sample(0, dim, 0.5, rp, idx, val, 0, 500, 3)
    b = oracle.selfjoin_sample(2, dim, 0.5, rp, idx, val, 0, 500, 3)
    c = oracle.selfjoin_sample(2, dim, 0.5, rp, idx, val, 0, 500, 1)
    assert a["visits"] == b["visits"] == c["visits"]
    assert b["cand_pairs"] >= a["cand_pairs"] == c["cand_pairs"] and b["pairs"] >= a["pairs"] == c["pairs"]
    assert b["cand_pairs"] < 3 * a["cand_pairs"]
