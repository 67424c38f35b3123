"""GPU tests of apss_group (include/apss.h, csrc/apss_group.hip): the term-sharded index of one node behind ONE object of
the C ABI -- member shard handles, one host thread each, and the exchange of their answers (all-gather of candidate
lists, all-reduce(SUM) of per-candidate partial scores) below the boundary.  On the one-GPU box the members share device 0
(exchange by device-to-device copies); a group of one member runs the RCCL exchange itself (APSS_GROUP_FORCE_EXCHANGE).
Reference: WriteWorkerActor.scala:164-183 (term buckets) + EntryProxyActor.scala:37-49 (fan-out) +
IndexingWorkerActor.scala:122-137 (IndexData per worker); checked against the CPU oracle."""
import numpy as np
import pytest

from apss import _lib, synth
from helpers import assert_same_pairs, to_map

pytestmark = pytest.mark.gpu


def _group(dim, theta, T, **kw):
    from apss.engine import ApssGroup
    return ApssGroup(dim, theta, [0] * T, **kw)


@pytest.mark.parametrize("T,zipf,theta,head", [(4, 0.0, 0.5, 0), (2, 1.0, 0.6, 64), (3, 1.0, 0.55, 700), (8, 0.0, 0.8, -1), (5, 1.0, 0.5, -1)])
def test_group_on_one_device_matches_oracle(oracle, T, zipf, theta, head):
    """T members on device 0: the member-local joins run on T host threads, the exchange by copies; external ids are not row
    numbers; the answer equals the oracle's and the statistics add up"""
    n, dim, nnz = 4000, 2000, 30
    rp, idx, val = synth.make_vectors(n, dim, nnz, zipf, seed=17 + T, dup_frac=0.1)
    ids = np.arange(n, dtype=np.int64) * 3 + 1000
    oq, oc, os_ = oracle.selfjoin_pairs(dim, theta, rp, idx, val)
    want = to_map(ids[oq], ids[oc], os_)
    assert len(want) > 100
    with _group(dim, theta, T, head_terms=head, tile_rows=1024) as g:
        got = to_map(*g.insert_and_query(ids, rp, idx, val))
        st = g.stats()
        assert_same_pairs(got, want, theta)
        assert st["n_members"] == T and st["exchange"] == _lib.EXCHANGE_COPIES and st["rows"] == n
        assert st["result_pairs"] == len(got) <= st["union_pairs"] <= st["candidates_sum"]
        assert st["all_reduce_bytes"] == 4 * st["union_pairs"] and st["all_gather_bytes"] <= 8 * st["candidates_sum"]
        cuts = st["term_cuts"]
        assert cuts[0] == 0 and cuts[-1] == dim and all(a < b for a, b in zip(cuts[:-1], cuts[1:]))
        assert st["head_terms"] == (0 if head <= 0 else min(head, dim))
        df = np.bincount(idx, minlength=dim).astype(np.int64)
        if st["head_terms"] == 0:  # every posting is visited by exactly one member
            assert st["posting_visits"] == int((df ** 2).sum())
        ms = [g.member_stats(i) for i in range(T)]
        assert sum(m["nnz"] for m in ms) == st["nnz"] and all(m["rows"] == n for m in ms)
        # a second call on the same layout (benchmark step: clear, build, join) gives the same list
        g.clear()
        assert g.stats()["term_cuts"] == cuts
        again = to_map(*g.insert_and_query(ids, rp, idx, val))
        assert again.keys() == got.keys()


def test_group_of_one_member_runs_the_rccl_exchange(oracle):
    """one member with a GPU of its own: ncclCommInitAll over one device, the broadcast-grouped all-gather and the all-reduce
    on the member's stream (librccl loaded at run time); same answer as the plain handle inside"""
    n, dim, nnz, theta = 5000, 3000, 30, 0.5
    rp, idx, val = synth.make_vectors(n, dim, nnz, 0.0, seed=5, dup_frac=0.1)
    ids = np.arange(n, dtype=np.int64) + 7
    oq, oc, os_ = oracle.selfjoin_pairs(dim, theta, rp, idx, val)
    want = to_map(ids[oq], ids[oc], os_)
    from apss.engine import ApssGroup
    with ApssGroup(dim, theta, [0], group_flags=_lib.GROUP_FORCE_EXCHANGE) as g:
        got = to_map(*g.insert_and_query(ids, rp, idx, val))
        st = g.stats()
        assert st["exchange"] == _lib.EXCHANGE_RCCL and st["union_pairs"] == st["candidates_sum"] == len(got)
        assert_same_pairs(got, want, theta)
    with ApssGroup(dim, theta, [0]) as g:  # without the flag: the member's answer is final, no exchange
        got2 = to_map(*g.insert_and_query(ids, rp, idx, val))
        assert g.stats()["exchange"] == _lib.EXCHANGE_NONE
        assert_same_pairs(got2, want, theta)
    with ApssGroup(dim, theta, [0], group_flags=_lib.GROUP_FORCE_EXCHANGE | _lib.GROUP_NO_RCCL) as g:
        got3 = to_map(*g.insert_and_query(ids, rp, idx, val))
        assert g.stats()["exchange"] == _lib.EXCHANGE_COPIES
        assert_same_pairs(got3, want, theta)


def test_group_c2_size_zipf1_with_the_head_block(oracle):
    """BASELINE.json configs[1]'s size (N = 100k, dim = 10k, nnz = 50, Zipf(1), theta = 0.5) through a group of four members that
    share a 256-term dense-head block: the same pairs as one plain handle, and the oracle's on a query sample"""
    from apss.engine import ApssIndex
    cfg, rp, idx, val = synth.make_config("c2")
    n, dim, theta = cfg["n"], cfg["dim"], cfg["theta"]
    ids = np.arange(n, dtype=np.int64)
    with ApssIndex(dim, theta, head_terms=-1) as ix:
        ref = to_map(*ix.insert_and_query(ids, rp, idx, val))
    with _group(dim, theta, 4, head_terms=256) as g:
        got = to_map(*g.insert_and_query(ids, rp, idx, val))
        st = g.stats()
        assert st["head_terms"] == 256 and st["head_ms_max"] > 0
        assert all(g.member_stats(i)["head_terms"] == 256 for i in range(4))
    assert len(got) > 1000
    assert_same_pairs(got, ref, theta, band=2e-5, tol=5e-6)
    sample = 1500
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val, 0, sample))
    assert_same_pairs({k: v for k, v in got.items() if k[0] < sample}, want, theta)


def test_group_c3_uniform_200k(oracle):
    """BASELINE.json configs[3]'s shape (dim 100k, nnz 100, uniform, theta 0.8: the term-sharded index) at N = 200k through a
    group of four members: the plain handle's pairs, the oracle's on a query sample, posting visits = sum df^2, a member's
    whole-store join symmetric"""
    from apss.engine import ApssIndex
    n, dim, nnz, theta = 200_000, 100_000, 100, 0.8
    rp, idx, val = synth.make_vectors(n, dim, nnz, 0.0, seed=synth.CONFIGS["c3"]["seed"])
    ids = np.arange(n, dtype=np.int64)
    with ApssIndex(dim, theta) as ix:
        ref = to_map(*ix.insert_and_query(ids, rp, idx, val))
    with _group(dim, theta, 4) as g:
        got = to_map(*g.insert_and_query(ids, rp, idx, val))
        st = g.stats()
        ms = [g.member_stats(i) for i in range(4)]
    assert len(ref) > 1000
    assert_same_pairs(got, ref, theta, band=2e-5, tol=5e-6)
    assert st["posting_visits"] == int(synth.workload_counts(dim, rp, idx)[1]) and st["head_terms"] == 0
    assert st["device_posting_visits"] < 0.7 * st["posting_visits"] and all(m["symmetric"] == 1 for m in ms)
    assert st["candidates_max"] < 4 * len(ref) + 1000  # the exchange is of the order of the true pairs
    sample = 1000
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val, 0, sample))
    assert_same_pairs({k: v for k, v in got.items() if k[0] < sample}, want, theta)


@pytest.mark.parametrize("head", [-1, 64])
def test_group_streaming_batches_match_the_oracle_worker(oracle, head):
    """the IndexData handler batch after batch (IndexingWorkerActor.scala:123-137) on a group of three members: the layout is
    decided on the first batch, every batch's answer equals the oracle worker's; then a frozen-index query batch and a
    plain insert"""
    n, dim, nnz, theta = 5000, 2500, 20, 0.5
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=77 + head, dup_frac=0.15)
    w = oracle.Worker(dim, theta)
    cuts = [0, 1203, 1900, 1967, 1968, 4100]
    with _group(dim, theta, 3, head_terms=head) as g:
        assert g.query(np.zeros(0, np.int64), np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0)) [0].size == 0
        for b0, b1 in zip(cuts[:-1], cuts[1:]):
            sl = slice(rp[b0], rp[b1])
            args = (np.arange(b0, b1), rp[b0:b1 + 1] - rp[b0], idx[sl], val[sl])
            want = to_map(*w.index_data(*args))
            got = to_map(*g.insert_and_query(*args))
            assert_same_pairs(got, want, theta, band=2e-5, tol=2e-5)
        assert g.stats()["head_terms"] == max(head, 0) and g.stats()["rows"] == 4100
        b0, b1 = 4100, 4600  # an outside batch against the frozen index (stopUpdateIndex, IWA:125-127)
        sl = slice(rp[b0], rp[b1])
        args = (np.arange(b0, b1), rp[b0:b1 + 1] - rp[b0], idx[sl], val[sl])
        want = to_map(*w.index_data(*args, query_only=True))
        assert_same_pairs(to_map(*g.query(*args)), want, theta, band=2e-5, tol=2e-5)
        assert g.stats()["rows"] == 4100
        b0, b1 = 4600, 5000  # insert only, then the next batch sees those rows
        sl = slice(rp[b0], rp[b1])
        args = (np.arange(b0, b1), rp[b0:b1 + 1] - rp[b0], idx[sl], val[sl])
        w.index_data(*args, build_only=True)
        g.insert(*args)
        with pytest.raises(Exception):
            g.fetch()  # an insert invalidates the last results (as on a handle)
        b0, b1 = 0, 300  # the same ids again: stored a second time, self-exclusion by external id (IWA:91)
        sl = slice(rp[b0], rp[b1])
        args = (np.arange(b0, b1), rp[b0:b1 + 1] - rp[b0], idx[sl], val[sl])
        want = to_map(*w.index_data(*args))
        assert_same_pairs(to_map(*g.insert_and_query(*args)), want, theta, band=2e-5, tol=2e-5)


def test_group_device_entry_point_and_named_cuts(oracle):
    """the batch resident in HBM (bench.py's form), the term cuts named by the caller"""
    import torch
    n, dim, nnz, theta = 6000, 4000, 40, 0.6
    rp, idx, val = synth.make_vectors(n, dim, nnz, 0.5, seed=3, dup_frac=0.1)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, rp, idx, val))
    dev = torch.device("cuda", 0)
    d = (torch.arange(n, dtype=torch.int64, device=dev), torch.from_numpy(rp).to(dev), torch.from_numpy(idx).to(dev),
         torch.from_numpy(val.astype(np.float32)).to(dev))
    torch.cuda.synchronize()
    cuts = [0, 700, 1500, 4000]
    with _group(dim, theta, 3, term_cuts=cuts, head_terms=-1) as g:
        n_res = g.insert_and_query_dev([d, d, d])
        got = to_map(*g.fetch())
        assert n_res == len(got) and g.stats()["term_cuts"] == cuts
        assert_same_pairs(got, want, theta)
    with _group(dim, theta, 2) as g:  # cuts decided from the device batch (document frequencies counted on the device)
        g.insert_and_query_dev([d, d])
        assert_same_pairs(to_map(*g.fetch()), want, theta)
        c = g.stats()["term_cuts"]
        df2 = np.bincount(idx, minlength=dim).astype(np.float64) ** 2
        assert abs(df2[:c[1]].sum() / df2.sum() - 0.5) < 0.05  # balanced by sum df^2


def test_group_equals_the_multi_process_twin(oracle):
    """apss.dist (one process per GPU, torch.distributed) and apss_group (one process, the exchange below the C ABI) combine
    the same shard handles: same layout in, same list out"""
    import torch
    from apss.dist import HipShardEngine, join_shards_local
    n, dim, nnz, theta, T = 8000, 5000, 40, 0.55, 4
    rp, idx, val = synth.make_vectors(n, dim, nnz, 1.0, seed=31, dup_frac=0.1)
    with _group(dim, theta, T, head_terms=128) as g:
        got = to_map(*g.insert_and_query(np.arange(n), rp, idx, val))
        st = g.stats()
        head = None
    cuts = st["term_cuts"]
    df = np.bincount(idx, minlength=dim)
    head = np.lexsort((np.arange(dim), -df))[:128].astype(np.int32)
    dev = torch.device("cuda", 0)
    engines = [HipShardEngine(dim, theta, (cuts[i], cuts[i + 1]), dev, head=(head, i, T, 0)) for i in range(T)]
    for e in engines:
        e.load(rp, idx, val)
    q, c, s, n_cand = join_shards_local(engines, n, theta)
    twin = to_map(q, c, s)
    assert twin.keys() == got.keys() and len(got) > 100
    assert max(abs(twin[k] - got[k]) for k in got) <= 2e-6
    assert sum(n_cand) == st["candidates_sum"]


def test_group_error_behaviour():
    """a malformed batch is refused by every member before anything is committed (the require of CommonUtils.scala:99): the
    group answers APSS_E_INVALID with the member's message and keeps working; bad arguments at creation"""
    from apss.engine import ApssError, ApssGroup
    dim, theta = 100, 0.5
    with pytest.raises(ApssError) as e:
        ApssGroup(dim, 0.0, [0, 0])  # term shards need theta > 0
    assert e.value.code == _lib.E_UNSUPPORTED
    with pytest.raises(ApssError) as e:
        ApssGroup(dim, theta, [0, 99])
    assert e.value.code == _lib.E_DEVICE
    with pytest.raises(ApssError):
        ApssGroup(dim, theta, [0, 0], term_cuts=[0, 60, 50])
    with _group(dim, theta, 2) as g:
        with pytest.raises(ApssError):
            g.fetch()
        rp = np.array([0, 2, 4], np.int64)
        good = (np.array([1, 2]), rp, np.array([3, 70, 3, 70], np.int32), np.array([0.6, 0.8, 0.8, 0.6]))
        bad = (np.array([3, 4]), rp, np.array([5, 5, 1, 2], np.int32), np.array([1.0, 1.0, 1.0, 1.0]))  # not strictly increasing
        q, c, s = g.insert_and_query(*good)
        assert sorted(zip(q.tolist(), c.tolist())) == [(1, 2), (2, 1)] and abs(s[0] - 0.96) < 1e-6
        with pytest.raises(ApssError) as e:
            g.insert_and_query(*bad)
        assert e.value.code == _lib.E_INVALID and "member" in str(e.value)
        assert g.stats()["rows"] == 2
        q, c, s = g.insert_and_query(np.array([9]), np.array([0, 2], np.int64), np.array([3, 70], np.int32), np.array([0.6, 0.8]))
        assert sorted(zip(q.tolist(), c.tolist())) == [(9, 1), (9, 2)]
        with pytest.raises(ApssError):  # the layout is fixed once the members exist
            g._chk(g._L.apss_group_set_term_cuts(g._g, np.array([0, 10, 100], np.int32).ctypes.data))


def test_group_ingest_flags(oracle):
    """L2 normalisation and the value prune on ingest (LoadGenerator.scala:34-37, WriteWorkerActor.scala:188-194) run on every
    member on the WHOLE row: the shard ratios and the result equal the oracle's on the transformed rows"""
    n, dim, nnz, theta, thr = 3000, 1500, 24, 0.55, 0.08
    rp, idx, val = synth.make_vectors(n, dim, nnz, 0.0, seed=8, dup_frac=0.15)
    raw = val * np.repeat(np.random.default_rng(1).uniform(0.5, 3.0, size=n), nnz)
    nv = oracle.l2_normalize(rp, raw)
    prp, pidx, pval = oracle.value_prune(rp, idx, nv, thr)
    want = to_map(*oracle.selfjoin_pairs(dim, theta, prp, pidx, pval))
    assert len(want) > 100
    with _group(dim, theta, 3, flags=_lib.FLAG_NORMALIZE | _lib.FLAG_VALUE_PRUNE, index_threshold=thr, head_terms=-1) as g:
        got = to_map(*g.insert_and_query(np.arange(n), rp, idx, raw))
    assert_same_pairs(got, want, theta, band=2e-5, tol=2e-5)


def test_group_admission_filter_drops_the_same_rows_on_every_member(oracle):
    """APSS_FLAG_ADMISSION (EntryProxyActor.scala:81-93: a vector is admitted iff sum_i v_i >= theta) is a decision about the
    WHOLE row, which every member is handed: the members drop the same rows, slots stay aligned across them, and the group's
    answer (external ids through member 0's arrays) equals the oracle worker's on the admitted rows"""
    n, dim, nnz, theta = 4000, 1200, 12, 0.6
    rp, idx, val = synth.make_vectors(n, dim, nnz, 0.0, seed=23, dup_frac=0.4)
    rng = np.random.default_rng(5)
    val = val * np.repeat(rng.uniform(0.12, 1.0, size=n), nnz)  # un-normalised: the row sums straddle theta
    keep = oracle.admission(rp, val, theta)
    assert 0.2 * n < keep.sum() < 0.95 * n
    rows = np.nonzero(keep)[0]
    krp = np.concatenate([[0], np.cumsum((rp[1:] - rp[:-1])[rows])]).astype(np.int64)
    sel = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in rows])
    ids = np.arange(n, dtype=np.int64) + 50
    w = oracle.Worker(dim, theta)
    want = to_map(*w.index_data(ids[rows], krp, idx[sel], val[sel]))
    assert len(want) > 20
    with _group(dim, theta, 3, flags=_lib.FLAG_ADMISSION, head_terms=0) as g:
        got = to_map(*g.insert_and_query(ids, rp, idx, val))
        st = g.stats()
        assert st["rows"] == int(keep.sum()) and st["head_terms"] == 0
        assert all(g.member_stats(i)["rows"] == st["rows"] for i in range(3))
    assert_same_pairs(got, want, theta, band=2e-5, tol=2e-5)
