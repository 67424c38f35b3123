"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against the CPU oracle on the same seeded
inputs, against the committed golden fixtures, and through size-independent properties.
Tolerances (BASELINE.json north_star): same pair set except inside |score - theta| <= 1e-5, scores within 1e-5."""
import json
import os

import numpy as np
import pytest

from apss import synth
from helpers import assert_same_pairs, to_map, topk

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def apss_mod():
    from apss import _lib, engine
    _lib.lib()  # raises if the HIP library is missing: no fallback
    return engine


def _gpu_join(engine, dim, theta, rp, idx, val, ids=None, **kw):
    n = len(rp) - 1
    ids = np.arange(n) if ids is None else ids
    with engine.ApssIndex(dim, theta, **kw) as ix:
        q, c, s = ix.insert_and_query(ids, rp, idx, val)
        st = ix.stats()
    return to_map(q, c, s), st


def test_kat_section33_through_abi(apss_mod):
    with open(os.path.join(GOLDEN, "kat_section33.json")) as f:
        k = json.load(f)
    names = ["v1", "v2", "v3", "v4", "v5"]
    rp, idx, val = [0], [], []
    for nm in names:
        v = k["vectors"][nm]
        ks = sorted(int(x) for x in v)
        idx += ks
        val += [v[str(x)] for x in ks]
        rp.append(len(idx))
    ids = np.array([int(nm[1:]) for nm in names])
    got, st = _gpu_join(apss_mod, k["dim"], k["theta"], np.array(rp), np.array(idx, np.int32), np.array(val), ids,
                        tile_rows=64)
    want = {(int(q[1:]), int(c[1:])): s for q, m in k["single_batch"]["intended"].items() for c, s in m.items()}
    assert_same_pairs(got, want, k["theta"])
    assert st["rows"] == 5 and st["nnz"] == 7


@pytest.mark.parametrize("name", ["mini_uniform_t03", "mini_zipf_t05", "mini_zipf_t08"])
@pytest.mark.parametrize("tile_rows", [256, 1024, 0])
@pytest.mark.parametrize("path", ["two_pass", "exact_wave", "general"])
def test_golden_fixture(apss_mod, name, tile_rows, path):
    """all three probe paths: coarse filter + exact rescoring (default), the single-pass exact speed kernel, and the
    general item-list kernel"""
    from apss import _lib
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    dim, theta = int(z["dim"]), float(z["theta"])
    got, st = _gpu_join(apss_mod, dim, theta, z["rowptr"], z["indices"], z["values"], tile_rows=tile_rows,
                        flags={"two_pass": 0, "exact_wave": _lib.FLAG_EXACT_ACCUM, "general": _lib.FLAG_FORCE_GENERAL}[path])
    assert (st["filter_survivors"] > 0) == (path == "two_pass")
    want = to_map(z["out_q"], z["out_c"], z["out_sim"])
    assert_same_pairs(got, want, theta)
    # top-k (first k by (-score, q, c)): equal sets imply equal top-k up to score ties inside 1e-5
    assert {k for k, _ in topk(got, 20)} <= set(want)
    # work counters: posting visits of a self-join = sum_t df_t^2
    _, visits = synth.workload_counts(dim, z["rowptr"], z["indices"])
    assert st["posting_visits"] == int(visits)


@pytest.mark.parametrize("path", ["two_pass", "exact_wave", "general"])
def test_join_cut_into_several_launches(apss_mod, monkeypatch, path):
    """a long join runs as a sequence of launches over groups of tiles (here: one tile per launch); same pairs and
    the same work counters as the single launch"""
    from apss import _lib
    monkeypatch.setenv("APSS_DEBUG", "tiles_per_launch=1")
    z = np.load(os.path.join(GOLDEN, "mini_zipf_t05.npz"))
    dim, theta = int(z["dim"]), float(z["theta"])
    got, st = _gpu_join(apss_mod, dim, theta, z["rowptr"], z["indices"], z["values"], tile_rows=128,
                        flags={"two_pass": 0, "exact_wave": _lib.FLAG_EXACT_ACCUM, "general": _lib.FLAG_FORCE_GENERAL}[path])
    assert st["probe_launches"] == st["tiles"] > 3
    assert_same_pairs(got, to_map(z["out_q"], z["out_c"], z["out_sim"]), theta)
    assert st["posting_visits"] == int(synth.workload_counts(dim, z["rowptr"], z["indices"])[1])


@pytest.mark.parametrize("flags_name", ["fast", "exact_wave", "force_general", "force_scan"])
def test_scan_path_equals_crossing_path(apss_mod, oracle, flags_name):
    from apss import _lib
    n, dim, nnz, theta = 1500, 300, 12, 0.45
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=21, dup_frac=0.1)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    got, st = _gpu_join(apss_mod, dim, theta, rp, idx, val, tile_rows=512,
                        flags={"fast": 0, "exact_wave": _lib.FLAG_EXACT_ACCUM, "force_general": _lib.FLAG_FORCE_GENERAL,
                               "force_scan": _lib.FLAG_FORCE_SCAN}[flags_name])
    assert len(want) > 100
    assert_same_pairs(got, want, theta)
    # candidate pairs scored == distinct (q, c != q) sharing a term
    cands = oracle.selfjoin_sample(1, dim, theta, rp, idx, val, 0, n, 2)["cand_pairs"]
    assert st["candidate_pairs"] == cands


def test_signed_weights_and_nonpositive_threshold(apss_mod, oracle):
    n, dim, nnz = 800, 200, 10
    rp, idx, val = synth.make_vectors(n, dim, nnz, 0.5, seed=22, dup_frac=0.1)
    rng = np.random.default_rng(1)
    sval = val * rng.choice([-1.0, 1.0], size=val.size)
    for theta, v in ((0.3, sval), (0.0, val), (-0.2, sval)):
        w = oracle.Worker(dim, theta)
        want = to_map(*w.index_data(np.arange(n), rp, idx, v))
        got, st = _gpu_join(apss_mod, dim, theta, rp, idx, v, tile_rows=256)
        assert len(want) > 50
        assert_same_pairs(got, want, theta)
        # signed weights with theta > 0 still take the two-pass join (the filter sums the positive products only);
        # theta <= 0 admits pairs of any score: accumulator scan of the general kernel
        assert (st["filter_survivors"] > 0) == (theta > 0)


def test_signed_weights_two_pass_at_scale(apss_mod, oracle):
    """weights of either sign (a third of the entries flipped, duplicates flip with their source or not) at a density
    where rounds carry real work; long rows included; candidate pairs counted exactly as for non-negative input"""
    import scipy.sparse as sp
    n, dim, nnz, theta = 30_000, 3_000, 24, 0.55
    rp, idx, val = synth.make_vectors(n, dim, nnz, 0.3, seed=77, dup_frac=0.15)
    rng = np.random.default_rng(9)
    sign_of_term = rng.choice([-1.0, 1.0, 1.0], size=dim)  # a term's sign: near-duplicates keep most of their dot product
    sval = val * sign_of_term[idx] * rng.choice([-1.0, 1.0], size=val.size, p=[0.05, 0.95])
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, sval))
    assert len(want) > 2000
    got, st = _gpu_join(apss_mod, dim, theta, rp, idx, sval)
    assert st["filter_survivors"] >= len(want)
    assert_same_pairs(got, want, theta)
    X = sp.csr_matrix((np.ones(idx.size, np.float32), idx, rp), shape=(n, dim))
    assert st["candidate_pairs"] == (X @ X.T).nnz - n


def test_streaming_batches_match_oracle_worker(apss_mod, oracle):
    n, dim, nnz, theta, b = 2400, 400, 14, 0.4, 700
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=23, dup_frac=0.1)
    w = oracle.Worker(dim, theta)
    with apss_mod.ApssIndex(dim, theta, tile_rows=512) as ix:
        for b0 in range(0, n, b):
            b1 = min(n, b0 + b)
            sl = slice(rp[b0], rp[b1])
            args = (np.arange(b0, b1), rp[b0:b1 + 1] - rp[b0], idx[sl], val[sl])
            want = to_map(*w.index_data(*args))
            got = to_map(*ix.insert_and_query(*args))
            assert_same_pairs(got, want, theta)
        assert ix.size() == (n, n * nnz)
        # frozen index (stopUpdateIndex, IWA:125-127): query-only, unseen dims are empty lists
        qrp, qidx, qval = synth.make_vectors(50, dim, nnz, 1.0, seed=24, dup_frac=0)
        want = to_map(*w.index_data(np.arange(50) + 10 ** 6, qrp, qidx, qval, query_only=True))
        got = to_map(*ix.query(np.arange(50) + 10 ** 6, qrp, qidx, qval))
        assert_same_pairs(got, want, theta)
        assert ix.size() == (n, n * nnz)
        # and the whole-store self-join equals the single-batch oracle
        whole = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
        assert_same_pairs(to_map(*ix.self_join()), whole, theta)


def test_ragged_empty_and_long_rows(apss_mod, oracle):
    """empty batch, empty rows, a row with more terms than a workgroup has threads (multi-pass), a term shared by
    every vector (workgroup-wide sweep), > kItemCap short segments in one round (list overflow sub-rounds)"""
    dim, theta = 3000, 0.2
    rng = np.random.default_rng(3)
    rows = []
    for i in range(900):
        k = int(rng.integers(0, 40))
        if i % 97 == 0:
            k = 0
        if i in (5, 450):
            k = 2600  # > 1024 terms and > 2048 work items
        t = np.sort(rng.choice(dim - 1, size=k, replace=False)) + 1
        t = np.concatenate([[0], t]) if i % 3 else t  # dim 0 in 2/3 of the rows -> long posting list
        v = np.abs(rng.standard_normal(t.size)) + 0.05
        rows.append((t.astype(np.int32), v / max(np.sqrt((v * v).sum()), 1e-30)))
    rp = np.concatenate([[0], np.cumsum([r[0].size for r in rows])])
    idx = np.concatenate([r[0] for r in rows])
    val = np.concatenate([r[1] for r in rows])
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    for tr in (64, 512):
        got, _ = _gpu_join(apss_mod, dim, theta, rp, idx, val, tile_rows=tr)
        assert_same_pairs(got, want, theta)
    with apss_mod.ApssIndex(dim, theta) as ix:
        e = np.zeros(0)
        assert ix.insert_and_query(e, [0], e, e)[0].size == 0
        assert ix.query(e, [0], e, e)[0].size == 0
        assert ix.self_join()[0].size == 0
        assert ix.size() == (0, 0)


def test_ext_id_semantics(apss_mod):
    # self-exclusion is by external id (IWA:91): same id stored twice never matches itself; equal vectors
    # under different ids match
    with apss_mod.ApssIndex(4, 0.5, tile_rows=64) as ix:
        q, c, s = ix.insert_and_query([7, 7, 8], [0, 1, 2, 3], [2, 2, 2], [1.0, 1.0, 1.0])
        got = {(int(a), int(b)) for a, b in zip(q, c)}
        assert got == {(7, 8), (8, 7)}
        assert np.allclose(s, 1.0)


def test_error_codes(apss_mod):
    from apss import _lib
    with apss_mod.ApssIndex(8, 0.5) as ix:
        with pytest.raises(apss_mod.ApssError) as e:
            ix.insert([1], [0, 2], [3, 3], [0.5, 0.5])  # not strictly increasing (SV:75)
        assert e.value.code == _lib.E_INVALID
        with pytest.raises(apss_mod.ApssError) as e:
            ix.insert([1], [0, 1], [8], [1.0])  # index >= vectorDim (the require of CU:99 / SV:105)
        assert e.value.code == _lib.E_INVALID
        with pytest.raises(apss_mod.ApssError) as e:
            ix.insert([1], [0, 1], [2], [float("nan")])
        assert e.value.code == _lib.E_INVALID
        assert ix.size() == (0, 0)  # failed batches leave no trace
        # device-pointer entry points validate on the device: a batch whose rowptr does not start at 0 is malformed
        import torch
        dv = lambda a, t: torch.tensor(a, dtype=t, device="cuda")  # noqa: E731
        with pytest.raises(apss_mod.ApssError) as e:
            ix.insert_dev(dv([1], torch.int64), dv([1, 2], torch.int64), dv([2, 3], torch.int32), dv([1.0, 1.0], torch.float32))
        assert e.value.code == _lib.E_INVALID and ix.size() == (0, 0)
        ix.insert([1], [0, 1], [2], [1.0])
        assert ix.size() == (1, 1)
    with pytest.raises(apss_mod.ApssError):
        apss_mod.ApssIndex(0, 0.5)
    with pytest.raises(apss_mod.ApssError):
        apss_mod.ApssIndex(8, 0.5, tile_rows=100)


def test_ingest_prefilters(apss_mod, oracle):
    """value prune (WWA:188-194), admission (EPA:81-93) and L2 normalisation (LG:34-37) on the device"""
    from apss import _lib
    n, dim, nnz, theta = 1200, 256, 12, 0.5
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=31, dup_frac=0.1)
    raw = val * np.repeat(np.random.default_rng(2).uniform(0.5, 3.0, n), nnz)  # un-normalised input
    ids = np.arange(n)
    nv = oracle.l2_normalize(rp, raw)
    # normalise -> prune -> index
    prp, pidx, pval = oracle.value_prune(rp, idx, nv, 0.1)
    assert pidx.size < idx.size
    want = to_map(*oracle.Worker(dim, theta).index_data(ids, prp, pidx, pval))
    assert len(want) > 30
    with apss_mod.ApssIndex(dim, theta, tile_rows=256, index_threshold=0.1,
                            flags=_lib.FLAG_NORMALIZE | _lib.FLAG_VALUE_PRUNE) as ix:
        got = to_map(*ix.insert_and_query(ids, rp, idx, raw))
        assert ix.size() == (n, pidx.size)
    assert_same_pairs(got, want, theta)
    # admission: sum_i v_i >= theta decides whether a vector is stored at all
    theta_a = 2.9
    keep = oracle.admission(rp, nv, theta_a)
    assert 0 < keep.sum() < n
    with apss_mod.ApssIndex(dim, theta_a, tile_rows=256, flags=_lib.FLAG_NORMALIZE | _lib.FLAG_ADMISSION) as ix:
        ix.insert(ids, rp, idx, raw)
        assert ix.size()[0] == int(keep.sum())


def test_c2_shape_reduced(apss_mod, oracle):
    """BASELINE config 2 shape (dim 10k, nnz 50, Zipf(1), theta 0.5) at N = 20k, default 32768-row tiles and 8192"""
    n, dim, nnz, theta = 20000, 10000, 50, 0.5
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=synth.CONFIGS["c2"]["seed"])
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    assert len(want) > 500
    from apss import _lib
    visits = int(synth.workload_counts(dim, rp, idx)[1])
    for tr, fl, head in ((0, 0, -1), (0, 0, 0), (8192, 0, -1), (0, _lib.FLAG_EXACT_ACCUM, 0), (0, _lib.FLAG_FORCE_GENERAL, 0)):
        got, st = _gpu_join(apss_mod, dim, theta, rp, idx, val, tile_rows=tr, flags=fl, head_terms=head)
        assert_same_pairs(got, want, theta)
        if st["head_terms"]:  # the library moved the most frequent terms to the dense block: their postings are not visited
            assert (tr, fl, head) == (0, 0, 0) and st["posting_visits"] < visits // 4 and st["head_pairs"] > 0
        else:
            assert st["posting_visits"] == visits


def test_c3_shape_properties(apss_mod):
    """BASELINE config 3 shape (dim 100k, nnz 100, uniform, theta 0.8) at N = 200k: size-independent properties
    -- symmetric result set inside one batch, every score in [theta - 1e-5, 1 + 1e-5], posting visits equal
    sum_t df_t^2, and the same set from a different tiling."""
    n, dim, nnz, theta = 200_000, 100_000, 100, 0.8
    rp, idx, val = synth.make_vectors(n, dim, nnz, 0.0, seed=synth.CONFIGS["c3"]["seed"])
    got, st = _gpu_join(apss_mod, dim, theta, rp, idx, val)
    assert len(got) > 1000
    assert all((c, q) in got for (q, c) in got)
    s = np.array(list(got.values()))
    assert s.min() >= theta - 1e-5 and s.max() <= 1 + 1e-5
    assert max(abs(got[(q, c)] - got[(c, q)]) for (q, c) in got) <= 2e-6
    assert st["posting_visits"] == int(synth.workload_counts(dim, rp, idx)[1])
    from apss import _lib
    got2, st2 = _gpu_join(apss_mod, dim, theta, rp, idx, val, tile_rows=8192, flags=_lib.FLAG_EXACT_ACCUM)
    assert_same_pairs(got2, got, theta, tol=2e-6)
    assert st2["candidate_pairs"] == st["candidate_pairs"] and st["filter_survivors"] >= len(got)


def test_unnormalised_input_keeps_the_two_pass_join(apss_mod, oracle):
    """row norms of 10 (scores up to 100): the filter runs at 2^9 units per 1.0 instead of 2^15 (its 16-bit sums still
    hold S |q||c| plus one unit per shared term), the exact pass is fp32 as ever; with exact accumulators only
    (APSS_FLAG_EXACT_ACCUM) such norms are beyond the fixed-point kernels and the fp32-atomic kernel takes over"""
    from apss import _lib
    n, dim, nnz = 1500, 300, 12
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=41, dup_frac=0.1)
    big = val * 10.0
    theta = 45.0
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, big))
    assert len(want) > 100
    got, st = _gpu_join(apss_mod, dim, theta, rp, idx, big, tile_rows=512)
    assert st["filter_survivors"] >= len(want)
    assert_same_pairs(got, want, theta, band=1e-3, tol=1e-3)  # fp32 sums of values around 100
    got, st = _gpu_join(apss_mod, dim, theta, rp, idx, big, tile_rows=512, flags=_lib.FLAG_EXACT_ACCUM)
    assert st["filter_survivors"] == 0
    assert_same_pairs(got, want, theta, band=1e-3, tol=1e-3)


def test_weights_beyond_fp16_use_the_float_path(apss_mod, oracle):
    """row norms of 1e5: the coarse postings (fp16 weights) cannot hold them, the fp32-atomic kernel runs the join"""
    n, dim, nnz = 1200, 300, 12
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=43, dup_frac=0.1)
    huge = val * 1.0e5
    theta = 4.5e9
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, huge))
    assert len(want) > 100
    got, st = _gpu_join(apss_mod, dim, theta, rp, idx, huge, tile_rows=512)
    assert st["filter_survivors"] == 0 and set(got) == set(want)
    for k, v in got.items():
        assert abs(v - want[k]) <= 1e-5 * abs(want[k])  # relative: scores of 1e10 in fp32


def test_medium_norms_use_the_coarser_fixed_point_scale(apss_mod, oracle):
    n, dim, nnz = 1500, 300, 12
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=42, dup_frac=0.1)
    v3 = val * 3.0  # |q||c| = 9: scale 2^28
    theta = 4.5
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, v3))
    assert len(want) > 100
    got, st = _gpu_join(apss_mod, dim, theta, rp, idx, v3, tile_rows=512)
    assert st["filter_survivors"] >= len(want)  # two-pass at 2^12 units per 1.0
    assert_same_pairs(got, want, theta, band=1e-4, tol=1e-4)  # inputs are fp32 on the device: 6e-8 * 9 * terms
    from apss import _lib
    got, st = _gpu_join(apss_mod, dim, theta, rp, idx, v3, tile_rows=512, flags=_lib.FLAG_EXACT_ACCUM)
    assert st["filter_survivors"] == 0  # the single-pass kernel at its coarser scale, 2^28
    assert_same_pairs(got, want, theta, band=1e-4, tol=1e-4)


def test_maildir_small_plumbing(apss_mod):
    """BASELINE config 1: TF-IDF vectors of the reference's mail corpus (HashingTF 2^20, rows of up to 2247 terms),
    fixture from tests/golden/make_maildir_fixture.py; exercises vectorDim = 2^20 and the > 512-terms-per-query path"""
    z = np.load(os.path.join(GOLDEN, "maildir_small_tfidf.npz"))
    dim, theta = int(z["dim"]), float(z["theta"])
    want = to_map(z["out_q"], z["out_c"], z["out_sim"])
    assert dim == 1 << 20 and len(want) > 100
    got, st = _gpu_join(apss_mod, dim, theta, z["rowptr"], z["indices"], z["values"])
    assert_same_pairs(got, want, theta)
    assert st["filter_survivors"] >= len(want)  # long rows go through the two-pass join too (in parts of 512 terms)
    # streamed in three batches the union of the answers is the subset where the query arrived no earlier than the hit
    n = len(z["rowptr"]) - 1
    rp, idx, val = z["rowptr"], z["indices"], z["values"]
    with apss_mod.ApssIndex(dim, theta, tile_rows=256) as ix:
        stream = {}
        for b0 in range(0, n, 256):
            b1 = min(n, b0 + 256)
            sl = slice(rp[b0], rp[b1])
            stream.update(to_map(*ix.insert_and_query(np.arange(b0, b1), rp[b0:b1 + 1] - rp[b0], idx[sl], val[sl])))
    assert_same_pairs(stream, {k: v for k, v in want.items() if k[0] >= (k[1] // 256) * 256}, theta)


def test_sparse_regime_with_signed_weights(apss_mod, oracle):
    """65536-row tiles (1024-thread filter kernel) with weights of either sign"""
    n, dim, nnz, theta = 70_000, 40_000, 16, 0.6
    rp, idx, val = synth.make_vectors(n, dim, nnz, 0.0, seed=56, dup_frac=0.05)
    sval = val * np.random.default_rng(3).choice([-1.0, 1.0, 1.0], size=dim)[idx]
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, sval))
    assert len(want) > 500
    got, st = _gpu_join(apss_mod, dim, theta, rp, idx, sval)
    assert st["tiles"] == 2 and st["filter_survivors"] >= len(want)
    assert_same_pairs(got, want, theta)


def test_sparse_regime_uses_65536_row_tiles(apss_mod, oracle):
    """few postings per (tile, term) segment (C5-like density): the handle picks 65536-row coarse tiles and the
    1024-thread filter kernel; more than one tile, the last one partial"""
    n, dim, nnz, theta = 70_000, 40_000, 16, 0.6
    rp, idx, val = synth.make_vectors(n, dim, nnz, 0.0, seed=55, dup_frac=0.05)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    assert len(want) > 1000
    got, st = _gpu_join(apss_mod, dim, theta, rp, idx, val)
    assert st["tiles"] == 2 and st["filter_survivors"] >= len(want)  # 65536 + 4464 rows
    assert_same_pairs(got, want, theta)
    assert st["posting_visits"] == int(synth.workload_counts(dim, rp, idx)[1])


def _long_rows(n, dim, seed, lo=5, hi=1800, long_frac=0.3, dup_frac=0.3):
    rng = np.random.default_rng(seed)
    rows = []
    for i in range(n):
        if i and rng.random() < dup_frac:  # a near-copy of an earlier row: a few terms dropped, weights jittered
            t, v = rows[int(rng.integers(0, i))]
            keep = rng.random(t.size) < 0.97
            if keep.sum() == 0:
                keep[0] = True
            t, v = t[keep], v[keep] * (1 + 0.05 * rng.standard_normal(int(keep.sum())))
            v = np.abs(v) + 1e-3
        else:
            k = int(rng.integers(513, hi)) if rng.random() < long_frac else int(rng.integers(lo, 60))
            t = np.sort(rng.choice(dim, size=min(k, dim), replace=False)).astype(np.int32)
            v = np.abs(rng.standard_normal(t.size)) + 0.01
        rows.append((t, v / np.sqrt((v * v).sum())))
    rp = np.concatenate([[0], np.cumsum([r[0].size for r in rows])]).astype(np.int64)
    return rp, np.concatenate([r[0] for r in rows]).astype(np.int32), np.concatenate([r[1] for r in rows])


@pytest.mark.parametrize("cx_tile", [None, "65536"])
def test_queries_longer_than_a_workgroup_use_the_two_pass_join(apss_mod, oracle, monkeypatch, cx_tile):
    """queries of > 512 terms are cut into parts that share the accumulators; both filter kernels (512 and 1024 threads);
    many near-copies of one long row overflow the survivor list of a part (> 512 crossings in one tile)"""
    if cx_tile:
        monkeypatch.setenv("APSS_DEBUG", "cx_tile=" + cx_tile)
    n, dim, theta = 2500, 6000, 0.7
    rp, idx, val = _long_rows(n, dim, seed=77)
    # 700 more near-copies of row 0, made long: every one of them crosses in the same tile for each of the others
    rng = np.random.default_rng(5)
    t0 = np.sort(rng.choice(dim, size=1500, replace=False)).astype(np.int32)
    v0 = np.abs(rng.standard_normal(1500)) + 0.01
    extra_t, extra_v, extra_rp = [], [], [int(rp[-1])]
    for _ in range(700):
        v = v0 * (1 + 0.02 * rng.standard_normal(1500))
        extra_t.append(t0)
        extra_v.append(v / np.sqrt((v * v).sum()))
        extra_rp.append(extra_rp[-1] + 1500)
    rp = np.concatenate([rp, np.array(extra_rp[1:], np.int64)])
    idx = np.concatenate([idx] + extra_t)
    val = np.concatenate([val] + extra_v)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    assert len(want) > 700 * 699
    got, st = _gpu_join(apss_mod, dim, theta, rp, idx, val)
    assert st["filter_survivors"] >= len(want) > 0
    assert_same_pairs(got, want, theta)
    assert st["posting_visits"] == int(synth.workload_counts(dim, rp, idx)[1])
    # frozen-index queries in small batches (parts of different queries interleave inside a workgroup's chunk)
    with apss_mod.ApssIndex(dim, theta) as ix:
        ix.insert(np.arange(len(rp) - 1), rp, idx, val)
        got2 = {}
        for b0 in range(0, 600, 97):
            b1 = min(600, b0 + 97)
            sl = slice(rp[b0], rp[b1])
            got2.update(to_map(*ix.query(np.arange(b0, b1), rp[b0:b1 + 1] - rp[b0], idx[sl], val[sl])))
    assert_same_pairs(got2, {k: v for k, v in want.items() if k[0] < 600}, theta)


def test_stratified_fullsize_property_reduced():
    """the oracle-free full-size check (profiles/fullsize_stratified.py, run at C5 = 1e7 x 200 for
    profiles/r01_c5_full.json) at a size that takes a second: exact scores elementwise, planted pairs all found,
    nothing else reported, posting visits == sum df^2"""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "fullsize_stratified", os.path.join(os.path.dirname(GOLDEN), "..", "profiles", "fullsize_stratified.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.run(300_000, 100_000, 50, 0.85, seed=7)
    assert out["missing"] == 0 and out["unexpected"] == 0 and out["planted_pairs_required"] > 10_000
    assert out["tiles"] >= 5
    # BASELINE.json configs[2] at full size (the shape bench.py measures), stratified: every planted pair, nothing else
    out = mod.run(1_000_000, 100_000, 100, 0.8, seed=20242)
    assert out["missing"] == 0 and out["unexpected"] == 0 and out["planted_pairs_required"] > 50_000
    assert out["tiles"] == 31 and out["probe_launches"] == 1 and out["max_abs_score_error"] <= 1e-5


@pytest.mark.parametrize("hook", [None, "chunk8", "window=3", "window=2"])
def test_candidate_pair_count_is_exact_when_the_window_overflows(apss_mod, monkeypatch, hook):
    """`candidate_pairs` (the benchmark's unit) against an independent count, scipy's boolean X X^T, at C3's segment
    density (33 postings per (tile, term)); the hooks shrink the register window so that most rounds take the
    overflow path (first touches are then counted per lane, not per wave)"""
    import scipy.sparse as sp
    if hook:
        monkeypatch.setenv("APSS_DEBUG", hook)
    n, dim, nnz, theta = 33_000, 100_000, 100, 0.7
    rp, idx, val = synth.make_vectors(n, dim, nnz, 0.0, seed=3, dup_frac=0.05)
    X = sp.csr_matrix((np.ones(idx.size, np.float32), idx, rp), shape=(n, dim))
    truth = (X @ X.T).nnz - n
    _, st = _gpu_join(apss_mod, dim, theta, rp, idx, val)
    assert st["candidate_pairs"] == truth
    assert st["posting_visits"] == int(synth.workload_counts(dim, rp, idx)[1])
    assert st["filter_survivors"] >= st["result_pairs"] > 1000


@pytest.mark.parametrize("path", ["two_pass", "exact_wave"])
def test_global_atomic_index_build_matches_the_lds_build(apss_mod, monkeypatch, path):
    """dims of more than 8 x 32768 terms build their index with global atomics (maildir: 2^20); the hook forces that
    path on a small dim: same pairs, same counters as the LDS-cursor build the other tests use"""
    from apss import _lib
    z = np.load(os.path.join(GOLDEN, "mini_zipf_t05.npz"))
    dim, theta = int(z["dim"]), float(z["theta"])
    flags = {"two_pass": 0, "exact_wave": _lib.FLAG_EXACT_ACCUM}[path]
    monkeypatch.setenv("APSS_DEBUG", "build_lds")  # (small batches default to the atomic build)
    ref, st_ref = _gpu_join(apss_mod, dim, theta, z["rowptr"], z["indices"], z["values"], tile_rows=256, flags=flags)
    monkeypatch.setenv("APSS_DEBUG", "build_atomic")
    got, st = _gpu_join(apss_mod, dim, theta, z["rowptr"], z["indices"], z["values"], tile_rows=256, flags=flags)
    assert got.keys() == ref.keys() and st["candidate_pairs"] == st_ref["candidate_pairs"]
    assert st["posting_visits"] == st_ref["posting_visits"] and st["nnz"] == st_ref["nnz"]
    assert_same_pairs(got, to_map(z["out_q"], z["out_c"], z["out_sim"]), theta)


@pytest.mark.parametrize("flags_name", ["two_pass", "exact_wave"])
def test_bucketed_index_build_for_large_dims(apss_mod, oracle, monkeypatch, flags_name):
    """dims of more than 16 x 16384 terms (the reference's HashingTF default is 2^20): a build from the first row partitions the
    entries by term range (k_bucket_pass) and runs the LDS build over the buckets instead of two global atomics per posting;
    same pairs and counters as the atomic build (APSS_DEBUG=no_bucket), the oracle's pairs on a query sample, and a later
    batch (appended: atomic kernels) on top"""
    from apss import _lib
    n, dim, nnz, theta = 150_000, 400_000, 24, 0.6
    rp, idx, val = synth.make_vectors(n, dim, nnz, 0.0, seed=404, dup_frac=0.08)
    flags = {"two_pass": 0, "exact_wave": _lib.FLAG_EXACT_ACCUM}[flags_name]
    got, st = _gpu_join(apss_mod, dim, theta, rp, idx, val, flags=flags)
    monkeypatch.setenv("APSS_DEBUG", "no_bucket")
    ref, st_ref = _gpu_join(apss_mod, dim, theta, rp, idx, val, flags=flags)
    monkeypatch.delenv("APSS_DEBUG")
    assert len(got) > 5000 and got.keys() == ref.keys()
    assert st["candidate_pairs"] == st_ref["candidate_pairs"] and st["posting_visits"] == st_ref["posting_visits"] == int(synth.workload_counts(dim, rp, idx)[1])
    sample = 1500
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val, 0, sample))
    assert_same_pairs({k: v for k, v in got.items() if k[0] < sample}, want, theta)
    # a second batch lands on the bucket-built index
    m = 120_000
    with apss_mod.ApssIndex(dim, theta, flags=flags) as ix:
        ix.insert(np.arange(m), rp[:m + 1], idx[:rp[m]], val[:rp[m]])
        q, c, s = ix.insert_and_query(np.arange(m, n), rp[m:] - rp[m], idx[rp[m]:], val[rp[m]:])
    later = to_map(q, c, s)
    assert later == {k: v for k, v in ref.items() if k[0] >= m} or later.keys() == {k for k in ref if k[0] >= m}


def test_handles_are_independent_across_threads(apss_mod, oracle):
    """the actor model of the reference (akka.conf:20-31): different workers run concurrently on a dispatcher's threads and
    a worker's thread may change between messages -- four handles driven from four threads at once (ctypes drops the GIL
    inside the library), then one handle passed from thread to thread"""
    import threading
    jobs = []
    for s in range(4):
        n, dim, nnz, theta = 3000 + 500 * s, 1500 + 100 * s, 20, 0.5
        rp, idx, val = synth.make_vectors(n, dim, nnz, 0.5 * (s % 2), seed=200 + s, dup_frac=0.1)
        jobs.append((dim, theta, rp, idx, val, to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))))
    got, errs = [None] * 4, []

    def work(i):
        try:
            dim, theta, rp, idx, val, _ = jobs[i]
            for _ in range(3):  # several joins per thread so that the calls really overlap
                got[i], _st = _gpu_join(apss_mod, dim, theta, rp, idx, val, tile_rows=512)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for i in range(4):
        assert_same_pairs(got[i], jobs[i][5], jobs[i][1])
    # one handle, a different thread for every message
    dim, theta, rp, idx, val, want = jobs[0]
    n = len(rp) - 1
    stream = {}
    with apss_mod.ApssIndex(dim, theta, tile_rows=256) as ix:
        def msg(b0, b1):
            sl = slice(rp[b0], rp[b1])
            stream.update(to_map(*ix.insert_and_query(np.arange(b0, b1), rp[b0:b1 + 1] - rp[b0], idx[sl], val[sl])))
        for b0 in range(0, n, 1000):
            t = threading.Thread(target=msg, args=(b0, min(n, b0 + 1000)))
            t.start()
            t.join()
    assert_same_pairs(stream, {k: v for k, v in want.items() if k[0] >= (k[1] // 1000) * 1000}, theta)


def test_results_are_invalidated_by_an_insert(apss_mod):
    """the results of a query-type call point into the store; an insert that grows the store frees those arrays, so the
    result calls answer APSS_E_STATE afterwards instead of reading freed HBM"""
    import torch
    from apss import _lib
    n, dim, nnz, theta = 600, 300, 10, 0.4
    rp, idx, val = synth.make_vectors(n, dim, nnz, 0.0, seed=8, dup_frac=0.2)
    with apss_mod.ApssIndex(dim, theta, tile_rows=64) as ix:
        q, c, s = ix.insert_and_query(np.arange(100), rp[:101], idx[:rp[100]], val[:rp[100]])
        assert len(q) > 0 and ix.result_count() == len(q)
        ix.insert(np.arange(100, n), rp[100:] - rp[100], idx[rp[100]:], val[rp[100]:])  # forces reallocation of the store
        for call in (ix.result_count, ix.fetch, ix.results_dev):
            with pytest.raises(apss_mod.ApssError) as e:
                call()
            assert e.value.code == _lib.E_STATE
        with pytest.raises(apss_mod.ApssError) as e:
            ix.partial_scores_dev(torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda"),
                                  torch.zeros(1, dtype=torch.float32, device="cuda"))
        assert e.value.code == _lib.E_STATE
        # and the next query-type call answers as usual
        q2, c2, s2 = ix.self_join()
        assert len(q2) >= len(q)
