"""Attacks on the soundness of the two-pass join's FILTER (k_probe_coarse): it must never lose a pair whose exact score
reaches theta (IndexingWorkerActor.scala:93 `sim >= similarityThreshold`); what it passes on too much is pruned by the
exact pass.  Seeded sweeps against the CPU oracle over the regimes where a 16-bit fixed-point sum of fp16 weights could
go wrong: weights spanning eight decades inside a row (fp16 subnormals), rows of 1 .. 2,247 terms, un-normalised rows at
every accumulator scale 2^4 .. 2^15 with sums a few units below the 16-bit carry, thresholds at the edge where the coarse
threshold is one unit, and signed weights whose products nearly cancel.  Every case asserts ZERO missing pairs."""
import numpy as np
import pytest

from helpers import to_map

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from apss import _lib, engine
    _lib.lib()
    return engine


def _csr(rows):
    rp = np.concatenate([[0], np.cumsum([len(r[0]) for r in rows])]).astype(np.int64)
    return rp, np.concatenate([r[0] for r in rows]).astype(np.int32), np.concatenate([r[1] for r in rows]).astype(np.float64)


def _check(engine, oracle, dim, theta, rows, band=1e-5, tol=1e-5, expect_filter=True, **kw):
    """GPU two-pass join vs oracle: no missing pair outside the band, no extra pair outside it, scores within tol
    (band / tol relative to the score scale for un-normalised rows)"""
    rp, idx, val = _csr(rows)
    n = len(rows)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    with engine.ApssIndex(dim, theta, head_terms=-1, **kw) as ix:
        got = to_map(*ix.insert_and_query(np.arange(n), rp, idx, val))
        st = ix.stats()
    if expect_filter:
        assert st["filter_survivors"] >= st["result_pairs"], "the two-pass path did not run"
    missing = [k for k in want if k not in got and abs(want[k] - theta) > band]
    extra = [k for k in got if k not in want and abs(got[k] - theta) > band]
    assert not missing, "filter lost %d pairs, e.g. %s" % (len(missing), [(k, want[k]) for k in missing[:4]])
    assert not extra, (len(extra), extra[:4])
    for k, v in got.items():
        if k in want:
            assert abs(v - want[k]) <= tol, (k, v, want[k])
    return want, got, st


def _near_copies(rng, base_t, base_v, dim, n_copies, jitter=0.03, replace=0.1):
    out = []
    for _ in range(n_copies):
        t, v = base_t.copy(), base_v * (1 + jitter * rng.standard_normal(base_v.size))
        k = max(0, int(replace * t.size))
        if k:
            pos = rng.choice(t.size, size=k, replace=False)
            new = rng.integers(0, dim, size=k)
            for p_, t_ in zip(pos, new):
                if t_ not in t:
                    t[p_] = t_
        o = np.argsort(t)
        out.append((t[o], np.abs(v[o]) if (base_v >= 0).all() else v[o]))
    return out


@pytest.mark.parametrize("seed", range(6))
def test_log_uniform_weights_inside_a_row(engine, oracle, seed):
    """weights log-uniform over [1e-8, 1] in every row: most of a row's entries are fp16 subnormals or below fp16's range
    (pack_coarse clamps those UP to the smallest subnormal); near-copies of such rows must all be found"""
    rng = np.random.default_rng(100 + seed)
    dim, nnz = 3000, int(rng.choice([8, 40, 200]))
    rows = []
    for _ in range(60):
        t = np.sort(rng.choice(dim, size=nnz, replace=False))
        v = 10.0 ** rng.uniform(-8, 0, size=nnz)
        v /= np.sqrt((v * v).sum())
        rows.append((t, v))
        for c in _near_copies(rng, t, v, dim, 4, jitter=0.05):
            rows.append((c[0], c[1] / np.sqrt((c[1] ** 2).sum())))
    theta = float(rng.choice([0.3, 0.7, 0.95]))
    want, _, _ = _check(engine, oracle, dim, theta, rows, tile_rows=int(rng.choice([64, 256, 0])))
    assert len(want) > 100


@pytest.mark.parametrize("seed", range(4))
def test_rows_of_one_to_2247_terms(engine, oracle, seed):
    """row lengths from 1 to 2,247 terms (the longest TF-IDF row of the mail corpus) in one batch; long rows run as
    virtual rows of <= 512 terms that share the accumulators"""
    rng = np.random.default_rng(200 + seed)
    dim = 6000
    rows = []
    for length in [1, 1, 2, 3, 5, 17, 64, 511, 512, 513, 1025, 2247]:
        t = np.sort(rng.choice(dim, size=length, replace=False))
        v = np.abs(rng.standard_normal(length)) + 0.01
        v /= np.sqrt((v * v).sum())
        rows.append((t, v))
        for c in _near_copies(rng, t, v, dim, 5, jitter=0.02, replace=0.05 if length > 10 else 0.0):
            rows.append((c[0], c[1] / np.sqrt((c[1] ** 2).sum())))
    order = rng.permutation(len(rows))
    rows = [rows[i] for i in order]
    theta = float(rng.choice([0.5, 0.8]))
    want, _, _ = _check(engine, oracle, dim, theta, rows)
    assert len(want) > 200


@pytest.mark.parametrize("k", range(4, 16))
def test_unnormalised_rows_at_the_edge_of_every_accumulator_scale(engine, oracle, k):
    """the filter picks S = 2^k units per 1.0, the largest with  max|q| max|c| * 1.0005 * S < 65535 - shared terms.  Row
    norms are placed so that exactly scale 2^k is chosen with the largest sums a few units below the 16-bit carry:
    duplicates of the largest row (score = its squared norm), all weights equal and sitting just ABOVE an fp16 rounding
    midpoint (every weight rounds up by almost 2^-11) -- the worst case for the carry guard -- plus random rows"""
    rng = np.random.default_rng(300 + k)
    dim, nnz = 2000, 64
    S = 2.0 ** k
    edge = (65535.0 - nnz) / (1.0005 * S)             # bound must stay below this
    norm2_max = (edge - 1e-6) / 1.0001 * (1 - 1e-5)   # squared norm of the largest rows
    v = np.sqrt(norm2_max / nnz)
    # snap v DOWN to just above an fp16 midpoint (mantissa 1 + 2^-11 (2 j + 1) + eps): rounds up by ~2^-11 relative
    e = np.floor(np.log2(v))
    m = v / 2.0 ** e
    j = np.floor((m - 1.0) * 2 ** 11 / 2 - 0.5)
    v_bad = (1.0 + (2 * j + 1) * 2.0 ** -11 + 2.0 ** -20) * 2.0 ** e
    assert 0.995 * v < v_bad <= v
    rows = []
    for _ in range(12):
        t = np.sort(rng.choice(dim, size=nnz, replace=False))
        for _ in range(4):
            rows.append((t, np.full(nnz, v_bad)))   # exact duplicates: score = nnz v_bad^2, the largest possible sum
    for _ in range(200):                             # background rows of smaller norms
        t = np.sort(rng.choice(dim, size=nnz, replace=False))
        w = np.abs(rng.standard_normal(nnz)) + 0.05
        w *= np.sqrt(norm2_max * rng.uniform(0.05, 0.9)) / np.sqrt((w * w).sum())
        rows.append((t, w))
        for c in _near_copies(rng, t, w, dim, 1, jitter=0.02):
            rows.append(c)
    theta = 0.8 * norm2_max * 0.5
    scale = norm2_max
    want, got, st = _check(engine, oracle, dim, theta, rows, band=2e-5 * scale, tol=2e-5 * scale, tile_rows=256)
    dup = [kv for kv in want.items() if abs(kv[1] - nnz * v_bad * v_bad) < 1e-6 * scale]
    assert len(dup) == 12 * 4 * 3 and all(kq in got for kq, _ in dup)


@pytest.mark.parametrize("seed", range(4))
def test_threshold_where_the_coarse_threshold_is_one_unit(engine, oracle, seed):
    """theta so small that floor(theta S (1 - 2^-11 - 1e-6)) - 2 = 1 unit (S = 2^15 for unit rows): the filter passes
    every touched candidate on; and the first theta for which the two-pass path is refused (one unit lower)"""
    rng = np.random.default_rng(400 + seed)
    dim, nnz, n = 500, 6, 400
    rows = []
    for _ in range(n):
        t = np.sort(rng.choice(dim, size=nnz, replace=False))
        v = 10.0 ** rng.uniform(-3, 0, size=nnz)
        rows.append((t, v / np.sqrt((v * v).sum())))
    for theta, two_pass in ((3.002 / 32768 / (1 - 2.0 ** -11 - 1e-6), True), (2.9 / 32768, False)):
        want, got, st = _check(engine, oracle, dim, theta, rows, band=1e-6, tol=1e-6, expect_filter=two_pass, tile_rows=128)
        assert (st["filter_survivors"] > 0) == two_pass
        assert len(want) > 1000


@pytest.mark.parametrize("seed", range(6))
def test_signed_weights_with_near_cancelling_products(engine, oracle, seed):
    """weights of either sign: the filter sums the positive products only (an upper bound of the score).  Pairs whose
    positive and negative products nearly cancel (score just above / just below theta) and pairs of large positive
    mass with a score far below theta"""
    rng = np.random.default_rng(500 + seed)
    dim, nnz = 1500, 40
    rows = []
    for _ in range(80):
        t = np.sort(rng.choice(dim, size=nnz, replace=False))
        v = rng.standard_normal(nnz)
        v /= np.sqrt((v * v).sum())
        rows.append((t, v))
        # a partner with the same support: a fraction f of the entries flipped in sign -> score ~ 1 - 2 f
        for f in (0.0, 0.05, 0.1, 0.25, 0.5):
            w = v * (1 + 0.01 * rng.standard_normal(nnz))
            flip = rng.random(nnz) < f
            w[flip] *= -1
            rows.append((t, w / np.sqrt((w * w).sum())))
    theta = float(rng.choice([0.5, 0.8, 0.9]))
    want, got, st = _check(engine, oracle, dim, theta, rows, tile_rows=int(rng.choice([64, 512])))
    assert len(want) > 100
    assert st["filter_survivors"] > st["result_pairs"]  # the upper bound lets cancelling pairs through; the exact pass drops them


def _bf16_round_down_loss(x):
    """relative amount by which round-to-nearest bf16 lowers x (negative: it raises it)"""
    import struct
    u = struct.unpack("<I", struct.pack("<f", np.float32(x)))[0]
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    y = struct.unpack("<f", struct.pack("<I", r))[0]
    return (float(np.float32(x)) - y) / float(np.float32(x))


@pytest.mark.parametrize("rendering", ["int8", "head_bf16"])
def test_dense_head_filter_at_the_worst_case_of_bf16_rounding(engine, oracle, monkeypatch, rendering):
    """the dense-head filter's threshold is theta - 0.0080 max|q||c|: bf16 round-to-nearest may lower each factor of a
    product by up to 2^-8.  Pairs built to sit right there: rows whose head part is k equal entries with 1/sqrt(k) just
    below a bf16 rounding midpoint (every entry of W rounds DOWN by nearly 2^-8), scores a hair above theta; all found"""
    # (both renderings of the block: bf16 rows with the threshold lowered by the rounding bound, and the default INT8 rows
    # rounded UP -- integer products, no bound at all; the pairs at theta (1 + 3e-5) test the fp32 arithmetic that makes them)
    if rendering != "int8":
        monkeypatch.setenv("APSS_DEBUG", rendering)
    # k whose 1/sqrt(k) loses the most in bf16
    k = max(range(8, 65), key=lambda kk: _bf16_round_down_loss(1.0 / np.sqrt(kk)))
    loss = _bf16_round_down_loss(1.0 / np.sqrt(k))
    assert loss > 0.0030  # close to the 2^-8 = 0.0039 worst case
    dim, theta = 400, 0.8
    rows = []
    head_terms = np.arange(k)
    rng = np.random.default_rng(8)
    # queries: unit rows, all mass on the k head terms (equal entries); candidates: the same direction, norm theta (1 + 3e-5)
    for _ in range(40):
        rows.append((head_terms, np.full(k, 1.0 / np.sqrt(k))))
    for _ in range(40):
        rows.append((head_terms, np.full(k, theta * (1 + 3e-5) / np.sqrt(k))))
    # filler: rows over the other terms, so that terms 0 .. k-1 are the most frequent ones and nothing else pairs up
    for _ in range(200):
        t = np.sort(rng.choice(np.arange(64, dim), size=12, replace=False))
        v = np.abs(rng.standard_normal(12)) + 0.1
        rows.append((t, v / np.sqrt((v * v).sum())))
    rp, idx, val = _csr(rows)
    n = len(rows)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    near = [kv for kv in want.items() if abs(kv[1] - theta * (1 + 3e-5)) < 1e-6]
    assert len(near) == 2 * 40 * 40  # every (unit row, short row) pair, both directions, score = theta (1 + 3e-5)
    with engine.ApssIndex(dim, theta, head_terms=64) as ix:
        got = to_map(*ix.insert_and_query(np.arange(n), rp, idx, val))
        st = ix.stats()
    assert st["head_terms"] == 64 and st["head_survivors"] >= len(near) and st["head_int8"] == (1 if rendering == "int8" else 0)
    missing = [kq for kq in want if kq not in got and abs(want[kq] - theta) > 1e-5]
    assert not missing, (len(missing), missing[:3])
    assert all(kq in want or abs(v - theta) <= 1e-5 for kq, v in got.items())


@pytest.mark.parametrize("geometry", ["default", "head_bf16", "fold_w=256", "fold_w=128", "mix=64", "mix=64,head_bf16"])
def test_folded_head_columns_at_the_worst_case_of_bf16_rounding(engine, oracle, monkeypatch, geometry):
    """the same attack on the FOLDED columns of a wide head (terms beyond the first `exact` add into shared columns; the test is
    an upper bound of the head's partial score only as long as rounding cannot take it below theta): pairs whose whole mass
    sits on k equal entries among the FOLDED terms, each 1/sqrt(k) just below a bf16 rounding midpoint, scores a hair above
    theta -- with the k terms chosen once in distinct columns (no collision helps) and once all in ONE column (a folded
    column is a sum, rounded once).  All found: in the default geometry (one block: 128 + 128 folded columns), another
    split of it, and the two-block forms kept for comparison"""
    if geometry != "default":
        monkeypatch.setenv("APSS_DEBUG", geometry)
    exact, fold = {"default": (128, 128), "head_bf16": (128, 128), "fold_w=256": (256, 256), "fold_w=128": (256, 128), "mix=64": (64, 192),
                   "mix=64,head_bf16": (64, 192)}[geometry]
    k = max(range(8, 65), key=lambda kk: _bf16_round_down_loss(1.0 / np.sqrt(kk)))
    dim, theta, n_head = 6000, 0.8, 256 + 4 * fold
    rng = np.random.default_rng(9)
    # the head is NAMED (apss_set_head_terms: terms 0 .. n_head-1 in this order, so the column of every folded term is known);
    # filler rows give every head term some postings
    rows = []
    for t in range(n_head):
        others = np.sort(rng.choice(np.arange(n_head, dim), size=6, replace=False))
        v = np.abs(rng.standard_normal(7)) + 0.1
        rows.append((np.concatenate([[t], others]), v / np.sqrt((v * v).sum())))
    folded = np.arange(exact, n_head)
    spread = folded[:k]                                       # k folded terms in k different columns
    stacked = folded[(folded - exact) % fold == 5][:4]        # four folded terms of ONE column
    assert spread.size == k and stacked.size == 4
    for terms in (spread, stacked):
        kk = terms.size
        for _ in range(20):
            rows.append((terms, np.full(kk, 1.0 / np.sqrt(kk))))
        for _ in range(20):
            rows.append((terms, np.full(kk, theta * (1 + 3e-5) / np.sqrt(kk))))
    rp, idx, val = _csr(rows)
    n = len(rows)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    near = [kv for kv in want.items() if abs(kv[1] - theta * (1 + 3e-5)) < 1e-6]
    assert len(near) == 2 * 2 * 20 * 20
    with engine.ApssIndex(dim, theta) as ix:
        ix.set_head_terms(np.arange(n_head))
        got = to_map(*ix.insert_and_query(np.arange(n), rp, idx, val))
        st = ix.stats()
        assert ix.head_terms().tolist() == list(range(n_head))
    assert st["head_terms"] == n_head and st["head_columns"] == exact + fold and st["head_survivors"] >= len(near)
    missing = [kq for kq in want if kq not in got and abs(want[kq] - theta) > 1e-5]
    assert not missing, (len(missing), missing[:3])
    assert all(kq in want or abs(v - theta) <= 1e-5 for kq, v in got.items())
