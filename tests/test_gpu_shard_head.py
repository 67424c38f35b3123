"""GPU tests of the dense-head block on TERM-SHARDED joins (BASELINE.json configs[4] as configured: skewed terms, the
index cut by term over the GPUs, the frequent terms as a bf16 MFMA block).  The block's terms H are a part of their own
in the partition {H, T_1 .. T_T} the candidate rule is proved for (include/apss.h): they are in no shard's inverted
index, shard i multiplies the candidate tiles t % T == i of the contraction, and the exact partial scores of phase 2 still
cover every term.  Several shard handles on one GPU, combined in-process exactly as apss.dist combines them across
GPUs (join_shards_local); checked against the CPU oracle (IndexingWorkerActor.scala:74-111 + CommonUtils.scala:98-117)."""
import os

import numpy as np
import pytest

from apss import synth
from helpers import assert_same_pairs, to_map

pytestmark = pytest.mark.gpu


def _top_terms(idx, dim, k):
    df = np.bincount(idx, minlength=dim)
    return np.lexsort((np.arange(dim), -df))[:k].astype(np.int32), df


def _shard_engines(dim, theta, idx, T, head, row_range=None, rp=None, val=None, tile_rows=0, fold=0):
    import torch
    from apss.dist import HipShardEngine, term_ranges
    df = np.bincount(idx, minlength=dim)
    if head is not None:
        df = df.copy()
        df[head] = 0
    dev = torch.device("cuda", 0)
    engines = [HipShardEngine(dim, theta, tr, dev, tile_rows=tile_rows, head=None if head is None else (head, i, T, fold))
               for i, tr in enumerate(term_ranges(df, T))]
    for e in engines:
        e.load(rp, idx, val, row_range)
    return engines


@pytest.mark.parametrize("T,kh", [(2, 64), (4, 40), (3, 256), (8, 128), (4, 700), (2, 1024), (3, 3000)])
@pytest.mark.parametrize("n,dim,nnz,theta", [(3000, 2048, 24, 0.5), (5000, 10000, 50, 0.6)])
def test_term_shards_with_a_head_block_match_oracle(oracle, T, kh, n, dim, nnz, theta):
    """small Zipf(1) batches: T tail shards + the block cut T ways by candidate tile; every block width (40 terms: a
    partially filled 64-wide block), partially filled query blocks and tiles"""
    from apss.dist import join_shards_local
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=41 + kh + T, dup_frac=0.1)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    assert len(want) > 100
    head, df = _top_terms(idx, dim, kh)
    fold = 192 if (kh > 256 and T % 2 == 1) else 0  # (one case with the split named by the caller: 64 terms with a column each + 192 folded columns)
    engines = _shard_engines(dim, theta, idx, T, head, rp=rp, val=val, tile_rows=1024, fold=fold)
    q, c, s, n_cand = join_shards_local(engines, n, theta)
    assert_same_pairs(to_map(q, c, s), want, theta)
    if kh > 256:
        assert all(e.stats["head_columns"] == 256 for e in engines)
    tail_df = df.astype(np.int64).copy()
    tail_df[head] = 0
    # the block's terms are in no shard's index; every other posting is visited by exactly one shard
    assert sum(e.stats["posting_visits"] for e in engines) == int((tail_df ** 2).sum())
    for e in engines:
        assert e.stats["head_terms"] == min(kh, dim) and e.stats["head_flops"] > 0 and e.stats["probe_kernel"].startswith("k_probe_")
    if kh > 256:
        return  # (head_pairs counts the pairs of the first block of 256)
    # pairs sharing a head term, counted once over the shards' tile shares (self pairs excluded): as on a plain handle
    import scipy.sparse as sp
    X = sp.csr_matrix((np.ones(idx.size, np.float32), idx, rp), shape=(n, dim))[:, np.sort(head)]
    assert sum(e.stats["head_pairs"] for e in engines) == (X @ X.T).nnz - int((X.getnnz(axis=1) > 0).sum())


def test_c2_size_zipf1_four_shards_and_the_block(oracle):
    """BASELINE.json configs[1]'s size (N=100k, dim=10k, nnz=50, Zipf(1), theta=0.5), four term shards + a 256-term block:
    the same pairs as one plain handle on the whole batch, and the oracle's on a query sample"""
    from apss.dist import join_shards_local
    from apss.engine import ApssIndex
    cfg, rp, idx, val = synth.make_config("c2")
    n, dim, theta = cfg["n"], cfg["dim"], cfg["theta"]
    with ApssIndex(dim, theta, head_terms=-1) as ix:
        ref = to_map(*ix.insert_and_query(np.arange(n), rp, idx, val))
    head, _ = _top_terms(idx, dim, 256)
    engines = _shard_engines(dim, theta, idx, 4, head, rp=rp, val=val)
    q, c, s, n_cand = join_shards_local(engines, n, theta)
    got = to_map(q, c, s)
    assert len(got) > 1000
    assert_same_pairs(got, ref, theta, band=2e-5, tol=5e-6)
    sample = 1500
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val, 0, sample))
    assert_same_pairs({k: v for k, v in got.items() if k[0] < sample}, want, theta)


def test_c5_power_law_reduced_four_shards_policy_block(oracle):
    """BASELINE.json configs[4] as configured, at reduced N: dim = 1M, nnz = 200, Zipf(1), theta = 0.9, N = 40k, the index
    cut into four term ranges and the block chosen by the library's own policy on a sample (apss.dist.hip_head_chooser:
    what rank 0 of a sharded join does).  Same pairs as one plain handle; the oracle's on a query sample"""
    import torch
    from apss.dist import hip_head_chooser, join_shards_local
    from apss.engine import ApssIndex
    cfg, rp, idx, val = synth.make_config("c5z", n=40_000, device="cuda")
    dim, theta, n = cfg["dim"], cfg["theta"], cfg["n"]
    head, fold = hip_head_chooser(dim, theta, torch.device("cuda", 0))(rp, idx, val)
    assert head.size in (64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768) and len(set(head.tolist())) == head.size
    assert fold == 0  # (the library's default split)
    engines = _shard_engines(dim, theta, idx, 4, head, rp=rp, val=val, fold=fold)
    q, c, s, n_cand = join_shards_local(engines, n, theta)
    got = to_map(q, c, s)
    for e in engines:  # the folded block's width travelled with the terms
        assert e.stats["head_columns"] == (256 if head.size > 256 else e.stats["head_columns"])
    with ApssIndex(dim, theta, head_terms=-1) as ix:
        ref = to_map(*ix.insert_and_query(np.arange(n), rp, idx, val))
    assert len(ref) > 1000
    assert_same_pairs(got, ref, theta, band=2e-5, tol=5e-6)
    sample = 400
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val, 0, sample))
    assert len(want) > 5
    assert_same_pairs({k: v for k, v in got.items() if k[0] < sample}, want, theta)
    # the exchange stays small although every pair of rows shares a head term
    assert max(n_cand) < 200 * len(ref)


def test_term_shards_x_candidate_ranges_with_a_head_block(oracle):
    """the T x D grid: two candidate ranges, each cut into two term shards + the block (queries arrive as an outside batch:
    the square form of the contraction, no mirrored pairs)"""
    from apss.dist import join_shards_local
    n, dim, nnz, theta = 5000, 3000, 30, 0.55
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=97, dup_frac=0.1)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    head, _ = _top_terms(idx, dim, 64)
    got = {}
    for r0, r1 in ((0, 1800), (1800, 5000)):
        engines = _shard_engines(dim, theta, idx, 2, head, row_range=(r0, r1), rp=rp, val=val, tile_rows=512)
        q, c, s, _ = join_shards_local(engines, n, theta)
        assert all(r0 <= x < r1 for x in c)
        got.update(to_map(q, c, s))
    assert len(want) > 200
    assert_same_pairs(got, want, theta)


def test_unnormalised_rows_and_ingest_flags_on_shards_with_a_block(oracle):
    """rows of norm != 1 (the rule divides by the full norms), and the ingest pre-filters that change what a row IS before
    it is scored: APSS_FLAG_NORMALIZE (LoadGenerator.scala:34-37) and APSS_FLAG_VALUE_PRUNE (WriteWorkerActor.scala:188-194)
    -- the block's rows are packed from the caller's batch, so they must see the same transformed rows as the store"""
    import torch
    from apss import _lib
    from apss.engine import ApssIndex
    n, dim, nnz, theta = 3000, 1500, 16, 0.45
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=19, dup_frac=0.12)
    rng = np.random.default_rng(5)
    val2 = val * np.repeat(rng.uniform(0.4, 1.6, size=n), nnz)
    head, df0 = _top_terms(idx, dim, 64)
    dev = torch.device("cuda", 0)
    for flags, thr in ((0, 0.0), (_lib.FLAG_NORMALIZE, 0.0), (_lib.FLAG_VALUE_PRUNE, 0.08), (_lib.FLAG_NORMALIZE | _lib.FLAG_VALUE_PRUNE, 0.1)):
        v = val2.copy()
        if flags & _lib.FLAG_NORMALIZE:
            v = oracle.l2_normalize(rp, v)
        r2, i2, v2 = (rp, idx, v) if not flags & _lib.FLAG_VALUE_PRUNE else oracle.value_prune(rp, idx, v, thr)
        want = to_map(*oracle.selfjoin_pairs(dim, theta, r2, i2, v2))
        assert len(want) > 100
        from apss.dist import term_ranges
        df = df0.copy()
        df[head] = 0
        shards = []
        for i, tr in enumerate(term_ranges(df, 3)):
            ix = ApssIndex(dim, theta, term_range=tr, flags=flags, index_threshold=thr)
            ix.set_head_terms(head, i, 3)
            shards.append(ix)
        keys = []
        for ix in shards:
            qq, cc, _ = ix.insert_and_query(np.arange(n), rp, idx, val2)
            keys.append(qq * n + cc)
        uniq = np.unique(np.concatenate(keys))
        uq = torch.from_numpy(uniq // n).to(dev).to(torch.int32)
        uc = torch.from_numpy(uniq % n).to(dev).to(torch.int32)
        total = torch.zeros(uniq.size, dtype=torch.float32, device=dev)
        for ix in shards:
            part = torch.empty_like(total)
            ix.partial_scores_dev(uq, uc, part)
            total += part
            ix.close()
        keep = (total >= theta).cpu().numpy()
        got = {(int(a), int(b)): float(x) for a, b, x in zip(uniq[keep] // n, uniq[keep] % n, total.cpu().numpy()[keep])}
        assert_same_pairs(got, want, theta, band=2e-5, tol=2e-5)


def test_set_head_terms_contract():
    """apss_set_head_terms: empty handle only; distinct terms inside [0, dim); a shard with a block refuses signed weights
    and the admission filter instead of silently leaving the partition its peers were given; the setting survives
    apss_clear; on a plain handle the caller's terms replace the policy's and give the same pairs"""
    from apss import _lib
    from apss.engine import ApssError, ApssIndex
    n, dim, nnz, theta = 2000, 1000, 12, 0.5
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=3, dup_frac=0.1)
    head, _ = _top_terms(idx, dim, 64)
    with ApssIndex(dim, theta, head_terms=-1) as ix:
        ref = to_map(*ix.insert_and_query(np.arange(n), rp, idx, val))
        with pytest.raises(ApssError) as e:
            ix.set_head_terms(head)
        assert e.value.code == _lib.E_STATE
        ix.clear()
        for bad in ([1, 1], [dim], [-1], list(range(32769))):
            with pytest.raises(ApssError) as e:
                ix.set_head_terms(bad)
            assert e.value.code == _lib.E_INVALID
        with pytest.raises(ApssError) as e:
            ix.set_head_terms(head, 2, 2)
        assert e.value.code == _lib.E_INVALID
        with pytest.raises(ApssError) as e:
            ix.set_head_terms(head, fold_columns=100)
        assert e.value.code == _lib.E_INVALID
        ix.set_head_terms(head)  # cfg.head_terms = -1 forbids the POLICY, not the caller
        for _ in range(2):
            got = to_map(*ix.insert_and_query(np.arange(n), rp, idx, val))
            st = ix.stats()
            assert st["head_terms"] == 64 and sorted(ix.head_terms().tolist()) == sorted(head.tolist())
            assert got.keys() == ref.keys()
            ix.clear()
        ix.set_head_terms([])
        got = to_map(*ix.insert_and_query(np.arange(n), rp, idx, val))
        assert ix.stats()["head_terms"] == 0 and got.keys() == ref.keys()
    with ApssIndex(dim, theta, term_range=(0, 500), flags=_lib.FLAG_ADMISSION) as ix:
        with pytest.raises(ApssError) as e:
            ix.set_head_terms(head, 0, 2)
        assert e.value.code == _lib.E_UNSUPPORTED
    with ApssIndex(dim, theta, term_range=(0, 500)) as ix:
        ix.set_head_terms(head, 0, 2)
        v = val.copy()
        v[3] = -v[3]
        with pytest.raises(ApssError) as e:
            ix.insert(np.arange(n), rp, idx, v)
        assert e.value.code == _lib.E_UNSUPPORTED and ix.size() == (0, 0)
        ix.insert(np.arange(n), rp, idx, val)  # the refused batch left nothing behind
        assert ix.size()[0] == n
        with pytest.raises(ApssError) as e:
            ix.query(np.arange(5), rp[:6], idx[:rp[5]], v[:rp[5]])
        assert e.value.code == _lib.E_UNSUPPORTED


def test_maildir_tfidf_through_term_shards():
    """BASELINE config 1 (TF-IDF vectors of the reference's own mail corpus: rows of up to 2,247 terms, the probe loop of
    IndexingWorkerActor.scala:101-109 over every one of them) through four term-range shards on one GPU: long rows go
    through the FILTER on a shard handle (virtual rows under the shard rule), not the general kernel; then the same with a
    64-term dense-head block.  Same pairs as the committed fixture (oracle output)"""
    from apss.dist import join_shards_local
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "maildir_small_tfidf.npz"))
    dim, theta = int(z["dim"]), float(z["theta"])
    rp, idx, val = z["rowptr"], z["indices"], z["values"]
    n = len(rp) - 1
    want = to_map(z["out_q"], z["out_c"], z["out_sim"])
    assert int(np.diff(rp).max()) > 2000
    for head in (None, _top_terms(idx, dim, 64)[0]):
        engines = _shard_engines(dim, theta, idx, 4, head, rp=rp, val=val)
        q, c, s, n_cand = join_shards_local(engines, n, theta)
        assert_same_pairs(to_map(q, c, s), want, theta)
        kernels = [e.stats["probe_kernel"] for e in engines]
        for e in engines:  # every shard went through the filter (a shard's survivors are its candidates) ...
            assert e.stats["filter_survivors"] > 0 and e.stats["probe_kernel"].startswith(("k_probe_coarse<", "k_probe_even<")), e.stats
        # ... and where a shard's slice of some row still has more than 512 terms, through its virtual-row instantiation
        assert any(k.startswith("k_probe_coarse<512, 5, 128, 512, true, 16, true") for k in kernels), kernels


def test_c1_every_document_plain_and_through_term_shards():
    """BASELINE config 1 at FULL size: all 8,586 documents of data/maildir_small as TF-IDF vectors (HashingTF 2^20, rows of up
    to 6,138 terms), theta = 0.7 -- on one plain handle, then through four term-range shards (long rows through the filter's
    virtual-row instantiation).  The fixture holds term counts + the oracle's pair list (tests/golden/make_maildir_full.py);
    the weights are derived by the function the generator fed the oracle with"""
    import maildir_full
    from apss.dist import join_shards_local
    from apss.engine import ApssIndex
    z, rp, idx, cnt = maildir_full.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "maildir_full_counts.npz"))
    val = maildir_full.weights(rp, idx, cnt)
    dim, theta, n = int(z["dim"]), float(z["theta"]), len(rp) - 1
    assert n == 8586 and int(np.diff(rp).max()) == 6138
    want = to_map(z["out_q"], z["out_c"], z["out_sim"])
    assert len(want) == 39472
    with ApssIndex(dim, theta) as ix:
        got = to_map(*ix.insert_and_query(np.arange(n), rp, idx, val))
        st = ix.stats()
    assert_same_pairs(got, want, theta)
    assert st["filter_survivors"] > 0 and st["probe_kernel"].startswith("k_probe_coarse<"), st
    engines = _shard_engines(dim, theta, idx, 4, None, rp=rp, val=val)
    q, c, s, n_cand = join_shards_local(engines, n, theta)
    assert_same_pairs(to_map(q, c, s), want, theta)
    for e in engines:
        assert e.stats["filter_survivors"] > 0 and e.stats["probe_kernel"].startswith(("k_probe_coarse<", "k_probe_even<")), e.stats


@pytest.mark.parametrize("kh", [64, 900])
def test_streaming_batches_on_term_shards_with_a_head_block(oracle, kh):
    """the IndexData handler batch after batch (IndexingWorkerActor.scala:123-137) on three term shards that share a head
    block: W rows, ratios and tail views are APPENDED per batch (batches that end inside a 64-row tile and inside an 8-row
    pack group); after the stream a frozen-index query batch (outside rows: packed on the fly).  Exchange by hand, as
    apss.dist does it; every batch's answer equals the oracle worker's"""
    import torch
    from apss.dist import term_ranges
    from apss.engine import ApssIndex
    n, dim, nnz, theta = 5000, 2500, 20, 0.5
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=123 + kh, dup_frac=0.15)
    head, df = _top_terms(idx, dim, kh)
    dft = df.copy()
    dft[head] = 0
    dev = torch.device("cuda", 0)
    shards = []
    for i, tr in enumerate(term_ranges(dft, 3)):
        ix = ApssIndex(dim, theta, term_range=tr)
        ix.set_head_terms(head, i, 3)
        shards.append(ix)
    w = oracle.Worker(dim, theta)

    def joined(call, ids, b0, b1, n_store):
        sl = slice(rp[b0], rp[b1])
        args = (ids, rp[b0:b1 + 1] - rp[b0], idx[sl], val[sl])
        keys = []
        for ix in shards:
            qq, cc, _ = getattr(ix, call)(*args)  # external ids: batch rows are b0.., stored rows 0..
            keys.append((qq - b0) * n + cc)
        uniq = np.unique(np.concatenate(keys)) if keys else np.zeros(0, np.int64)
        uq = torch.from_numpy(uniq // n).to(dev).to(torch.int32)
        uc = torch.from_numpy(uniq % n).to(dev).to(torch.int32)
        total = torch.zeros(uniq.size, dtype=torch.float32, device=dev)
        for ix in shards:
            part = torch.empty_like(total)
            if uniq.size:
                ix.partial_scores_dev(uq, uc, part)
                total += part
        keep = (total >= theta).cpu().numpy()
        return {(int(a) + b0, int(b)): float(x) for a, b, x in zip(uniq[keep] // n, uniq[keep] % n, total.cpu().numpy()[keep])}

    cuts = [0, 1003, 1900, 1967, 4100]
    for b0, b1 in zip(cuts[:-1], cuts[1:]):
        sl = slice(rp[b0], rp[b1])
        want = to_map(*w.index_data(np.arange(b0, b1), rp[b0:b1 + 1] - rp[b0], idx[sl], val[sl]))
        got = joined("insert_and_query", np.arange(b0, b1), b0, b1, b1)
        assert_same_pairs(got, want, theta, band=2e-5, tol=2e-5)
    b0, b1 = 4100, 5000  # never stored: an outside batch against the frozen index
    sl = slice(rp[b0], rp[b1])
    want = to_map(*w.index_data(np.arange(b0, b1), rp[b0:b1 + 1] - rp[b0], idx[sl], val[sl], query_only=True))
    got = joined("query", np.arange(b0, b1), b0, b1, 4100)
    assert len(want) > 50
    assert_same_pairs(got, want, theta, band=2e-5, tol=2e-5)
    for ix in shards:
        assert ix.stats()["head_terms"] == kh and ix.size()[0] == 4100
        ix.close()
