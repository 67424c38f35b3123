"""Randomised parity sweep: ragged rows (empty, single-term, long), duplicate ids, several batches, random tile sizes
and thresholds, all three probe paths, against the CPU oracle's IndexingWorkerActor restatement."""
import numpy as np
import pytest

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
