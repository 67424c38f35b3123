"""Randomised parity sweep: ragged rows (empty, single-term, long), duplicate ids, several batches, random tile sizes
and thresholds, all three probe paths, against the CPU oracle's IndexingWorkerActor restatement."""
import numpy as np
import pytest

from apss import synth
from helpers import assert_same_pairs, to_map

pytestmark = pytest.mark.gpu


def _random_case(rng):
    dim = int(rng.choice([8, 64, 300, 2000]))
    n = int(rng.integers(1, 700))
    kind = rng.choice(["short", "mixed", "long"])
    rows, ids = [], []
    for i in range(n):
        if i > 0 and rng.random() < 0.12:  # the same (id, vector) stored again, as when a vector reaches a worker twice
            j = int(rng.integers(0, i))
            rows.append(rows[j])
            ids.append(ids[j])
            continue
        if kind == "short":
            k = int(rng.integers(0, min(dim, 6) + 1))
        elif kind == "mixed":
            k = int(rng.integers(0, min(dim, 40) + 1))
        else:
            k = int(rng.integers(0, min(dim, 700) + 1)) if rng.random() < 0.1 else int(rng.integers(0, min(dim, 12) + 1))
        t = np.sort(rng.choice(dim, size=k, replace=False)).astype(np.int32)
        v = np.abs(rng.standard_normal(k)) + 0.01
        nrm = np.sqrt((v * v).sum())
        rows.append((t, v / nrm if nrm > 0 else v))
        ids.append(10_000 + i)
    rp = np.concatenate([[0], np.cumsum([r[0].size for r in rows])]).astype(np.int64)
    idx = np.concatenate([r[0] for r in rows]) if rows else np.zeros(0, np.int32)
    val = np.concatenate([r[1] for r in rows]) if rows else np.zeros(0)
    theta = float(rng.choice([0.05, 0.3, 0.6, 0.9]))
    cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, size=int(rng.integers(0, 3)))]))
    return dim, theta, np.array(ids, np.int64), rp, idx.astype(np.int32), val, cuts


@pytest.mark.parametrize("seed", range(24))
def test_random_streams(oracle, seed, monkeypatch):
    from apss import _lib
    from apss.engine import ApssIndex
    if seed % 4 == 1:
        monkeypatch.setenv("APSS_DEBUG", "build_lds")  # small batches default to the global-atomic index build
    rng = np.random.default_rng(1000 + seed)
    dim, theta, ids, rp, idx, val, cuts = _random_case(rng)
    flags = [0, _lib.FLAG_EXACT_ACCUM, _lib.FLAG_FORCE_GENERAL][seed % 3]
    tile = int(rng.choice([64, 128, 1024, 0]))
    w = oracle.Worker(dim, theta)
    with ApssIndex(dim, theta, tile_rows=tile, flags=flags) as ix:
        for b0, b1 in zip(cuts[:-1], cuts[1:]):
            sl = slice(rp[b0], rp[b1])
            args = (ids[b0:b1], rp[b0:b1 + 1] - rp[b0], idx[sl], val[sl])
            want = to_map(*w.index_data(*args))
            got = to_map(*ix.insert_and_query(*args))
            assert_same_pairs(got, want, theta)
        # the reference's output map is keyed by id: duplicates collapse to one entry per (q id, c id)
        assert ix.size()[0] == len(ids)


@pytest.mark.parametrize("head", [-1, 64, 600])
def test_single_vector_messages_wait_in_the_tail(oracle, head):
    """the LoadGenerator shape (benchmark/LoadGenerator.scala:58-74): after a warm-up batch, thousands of single-vector
    IndexData messages.  They wait in the tail (scored pair by pair, k_tail_score) and are folded into the tile index
    every 4096 rows; every message's answer equals the oracle worker's, including pairs between two tail rows, and a
    final self-join (tail folded in) equals the oracle's"""
    from apss.engine import ApssIndex
    n0, n1, dim, nnz, theta = 2000, 4400, 900, 14, 0.55
    rp, idx, val = synth.make_vectors(n0 + n1, dim, nnz, 1.0 if head > 0 else 0.0, seed=321, dup_frac=0.3)
    w = oracle.Worker(dim, theta)
    with ApssIndex(dim, theta, tile_rows=512, head_terms=head) as ix:
        e0 = int(rp[n0])
        want = to_map(*w.index_data(np.arange(n0), rp[:n0 + 1], idx[:e0], val[:e0]))
        assert_same_pairs(to_map(*ix.insert_and_query(np.arange(n0), rp[:n0 + 1], idx[:e0], val[:e0])), want, theta)
        found = 0
        r = n0
        sizes = [1] * 4200 + [3, 7, 1, 60, 1, 1, 120, 1, 5]
        for sz in sizes:
            sl = slice(rp[r], rp[r + sz])
            args = (np.arange(r, r + sz), rp[r:r + sz + 1] - rp[r], idx[sl], val[sl])
            want = to_map(*w.index_data(*args))
            got = to_map(*ix.insert_and_query(*args))
            assert_same_pairs(got, want, theta)
            found += len(got)
            r += sz
        assert found > 1000 and ix.size()[0] == r
        st = ix.stats()
        assert st["head_terms"] == (head if head > 0 else 0)  # (600: 256 columns + a folded block; the waiting rows' own W rows
        # and tail views are made when they are folded into the index)
        # frozen-index query while rows wait in the tail
        sl = slice(rp[100], rp[140])
        got = to_map(*ix.query(np.arange(100, 140), rp[100:141] - rp[100], idx[sl], val[sl]))
        want = to_map(*w.index_data(np.arange(100, 140), rp[100:141] - rp[100], idx[sl], val[sl], query_only=True))
        assert_same_pairs(got, want, theta)
        full = to_map(*oracle.selfjoin_pairs(dim, theta, rp[:r + 1], idx[:rp[r]], val[:rp[r]]))
        assert_same_pairs(to_map(*ix.self_join()), full, theta)


@pytest.mark.parametrize("seed", range(24))
def test_random_streams_with_a_forced_head_block(oracle, seed, monkeypatch):
    """the same randomised streams (ragged rows: empty, single-term, long; duplicate ids; several batches; thresholds from
    0.05 to 0.9) with the dense-head block forced on: rows without a head entry, rows with nothing BUT head entries (tail
    ratio 0), dims smaller than the block (the block is refused), batches that wait in the tail, the no_tail hook"""
    from apss.engine import ApssIndex
    if seed % 4 == 3:
        monkeypatch.setenv("APSS_DEBUG", "no_tail" + (",fold_w=128" if seed % 8 == 7 else ""))
    elif seed % 4 == 2:
        monkeypatch.setenv("APSS_DEBUG", "fold_w=128")
    rng = np.random.default_rng(5000 + seed)
    dim, theta, ids, rp, idx, val, cuts = _random_case(rng)
    kh = [64, 128, 256, 400, 1500][seed % 5]  # (beyond 256: a folded second block; dims below the width: every term in the head)
    w = oracle.Worker(dim, theta)
    with ApssIndex(dim, theta, tile_rows=int(rng.choice([64, 1024, 0])), head_terms=kh) as ix:
        for b0, b1 in zip(cuts[:-1], cuts[1:]):
            sl = slice(rp[b0], rp[b1])
            args = (ids[b0:b1], rp[b0:b1 + 1] - rp[b0], idx[sl], val[sl])
            want = to_map(*w.index_data(*args))
            got = to_map(*ix.insert_and_query(*args))
            assert_same_pairs(got, want, theta)
        assert ix.size()[0] == len(ids)
        st = ix.stats()
        assert st["head_terms"] in (0, min(kh, dim)) or st["head_terms"] <= kh
