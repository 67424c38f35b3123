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


def _worker(rank, world, port, out_dir, term_shards):
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
                     engine_factory=lambda tr: OracleShardEngine(dim, theta, tr))
    sj.load(rp, idx, val)
    q, c, s = sj.step(return_pairs=True)
    total = sj.step()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), q=q, c=c, s=s, lo=sj.term_range[0], hi=sj.term_range[1],
             union=sj.last["exchange"]["union"], r0=sj.row_range[0], r1=sj.row_range[1], total=total)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,term_shards", [(2, 2), (3, 3), (4, 2), (2, 1)])
def test_sharded_join_matches_oracle(tmp_path, oracle, world, term_shards):
    """T term shards x D candidate ranges: union over the D groups == the oracle's result; inside a group every rank
    holds the identical group result (all-reduced partial scores)"""
    from apss import synth
    from helpers import assert_same_pairs, to_map
    port = 29500 + (os.getpid() % 2000) + 7 * world + term_shards
    mp.spawn(_worker, args=(world, port, str(tmp_path), term_shards), nprocs=world, join=True)
    n, dim, nnz, theta = 1500, 400, 14, 0.6
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=77, dup_frac=0.1)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    assert len(want) > 50
    T, D = term_shards, world // term_shards
    got_all = {}
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        got = to_map(z["q"], z["c"], z["s"])
        r0, r1 = int(z["r0"]), int(z["r1"])
        assert all(r0 <= c < r1 for (_, c) in got)  # a group reports candidates of its own row range only
        sub = {k: v for k, v in want.items() if r0 <= k[1] < r1}
        assert_same_pairs(got, sub, theta, band=1e-5, tol=1e-5)
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
