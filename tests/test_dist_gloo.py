"""Multi-process test of the sharded join's host logic on CPU: world_size 2, gloo backend, oracle-backed shard
engines (tests/shard_oracle_engine.py).  The HIP shard engine itself is covered by tests/test_gpu_shards.py."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _top_df(k):
    """stand-in for the library's head policy on CPU: the k most frequent terms"""
    def choose(rp, idx, val):
        df = np.bincount(idx)
        return np.argsort(-df, kind="stable")[:k].astype(np.int32)
    return choose


def _worker(rank, world, port, out_dir, term_shards, head_k=0):
    sys.path[:0] = [os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "all-pairs-similarity_amd"), HERE]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from apss import synth
    from apss.dist import ShardedJoin
    from shard_oracle_engine import OracleShardEngine
    n, dim, nnz, theta = 1500, 400, 14, 0.6
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=77, dup_frac=0.1)
    sj = ShardedJoin(dim, theta, rank, world, torch.device("cpu"), term_shards=term_shards,
                     engine_factory=lambda tr, head=None: OracleShardEngine(dim, theta, tr, head),
                     head_terms=head_k if head_k else -1, head_chooser=_top_df(head_k))
    sj.load(rp, idx, val)
    assert sj.head.size == (head_k if term_shards > 1 else 0)
    if head_k and term_shards > 1:  # the block's terms are in no shard's tail test, and every rank holds the same block
        assert sj.engine.head[1:3] == (rank % term_shards, term_shards)
        assert float(abs(sj.engine.xt[:, [t - sj.term_range[0] for t in sj.head if sj.term_range[0] <= t < sj.term_range[1]]]).sum()) == 0.0
    q, c, s = sj.step(return_pairs=True)
    total = sj.step()
    # the partition rule's figures ride along in `last`: sum df / sum df^2 per term range, what every rank visited,
    # the exchange's bytes and this rank's time in it
    ex = sj.last["exchange"]
    assert len(ex["sum_df_per_term_range"]) == len(ex["sum_df2_per_term_range"]) == term_shards
    assert len(ex["device_posting_visits_per_rank"]) == world and ex["visits_imbalance_max_over_mean"] >= 1.0 - 1e-9
    assert ex["sum_df2_imbalance_max_over_mean"] >= 1.0 - 1e-9
    assert (ex["exchange_ms"] > 0 and ex["exchange_effective_GBps"] > 0) == (term_shards > 1)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), q=q, c=c, s=s, lo=sj.term_range[0], hi=sj.term_range[1],
             union=sj.last["exchange"]["union"], r0=sj.row_range[0], r1=sj.row_range[1], total=total)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,term_shards,head_k", [(2, 2, 0), (3, 3, 0), (4, 2, 0), (2, 1, 0), (3, 1, 0), (4, 1, 0), (2, 2, 12), (4, 2, 12), (3, 3, 40)])
def test_sharded_join_matches_oracle(tmp_path, oracle, world, term_shards, head_k):
    """T term shards x D candidate ranges: union over the D groups == the oracle's result; inside a group every rank
    holds the identical group result (all-reduced partial scores).  head_k > 0: the join's dense-head block -- rank 0's
    choice broadcast to every rank, the block's terms out of every tail range, its test cut over the group's ranks by
    candidate tile (Zipf(1) terms: the block holds most of a typical row's weight)"""
    from apss import synth
    from helpers import assert_same_pairs, to_map
    port = 29500 + (os.getpid() % 2000) + 7 * world + term_shards + 31 * (head_k > 0) + head_k
    mp.spawn(_worker, args=(world, port, str(tmp_path), term_shards, head_k), nprocs=world, join=True)
    n, dim, nnz, theta = 1500, 400, 14, 0.6
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=77, dup_frac=0.1)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    assert len(want) > 50
    T, D = term_shards, world // term_shards
    got_all, seen_cross = {}, set()
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        got = to_map(z["q"], z["c"], z["s"])
        r0, r1 = int(z["r0"]), int(z["r1"])
        assert len(got) == z["q"].size  # no pair twice inside a group's answer
        if D == 1:
            assert_same_pairs(got, want, theta, band=1e-5, tol=1e-5)
        else:
            # symmetric scheme across the D row ranges (apss.dist.ShardedJoin.half_spans): a group reports the pairs inside its
            # own range and, IN BOTH DIRECTIONS, the cross pairs it owns -- every pair touches its range, none is reported by two
            # groups, and all are the oracle's
            assert all(r0 <= q < r1 or r0 <= c < r1 for (q, c) in got)
            assert all(k in want or abs(v - theta) <= 1e-5 for k, v in got.items())
            if r % T == 0:
                assert not (set(got) & seen_cross), "a pair was reported by two groups"
                seen_cross |= set(got)
        assert int(z["total"]) == len(want) or abs(int(z["total"]) - len(want)) <= 2  # whole-job count on every rank
        got_all.update(got)
        if r % T == 0:
            assert (int(z["lo"]) == 0)
    assert_same_pairs(got_all, want, theta, band=1e-5, tol=1e-5)


def test_term_ranges_balance_visits():
    from apss.dist import term_ranges
    rng = np.random.default_rng(0)
    df = (1000 / np.arange(1, 2001)).astype(np.int64) + rng.integers(0, 3, 2000)
    for world in (2, 4, 8):
        r = term_ranges(df, world)
        assert r[0][0] == 0 and r[-1][1] == 2000 and all(a < b for a, b in r)
        w = [float((df[a:b].astype(float) ** 2).sum()) for a, b in r]
        assert max(w) <= 2.5 * (sum(w) / world) or world == 8  # heavy head term: a single term can dominate


def test_head_depth_rule_keeps_tail_terms_per_shard():
    """limit_head_depth: a policy head is kept only as deep as leaves ~8 tail terms per row and shard, never below the 256
    terms that hold the long posting lists (C3 with Zipf(1) terms at T = 8 measured 3.3e7 candidates per shard without it)"""
    from apss.dist import limit_head_depth
    dim, n, nnz = 100_000, 50_000, 100
    p = 1.0 / np.arange(1, dim + 1)
    df = np.minimum(n, (n * nnz * p / p.sum())).astype(np.int64)  # Zipf(1) document frequencies, most frequent first
    head = np.arange(8192, dtype=np.int32)
    total = int(df.sum())
    for T, want_max in ((2, 8192), (4, 2048), (8, 256), (16, 256)):
        h = limit_head_depth(head, df, n, T)
        tail = (total - int(df[h].sum())) / n / T
        assert 256 <= h.size <= want_max and np.array_equal(h, head[:h.size])
        assert tail >= 8.0 or h.size == 256
    assert limit_head_depth(head[:100], df, n, 8).size == 100  # a single narrow block is left alone
