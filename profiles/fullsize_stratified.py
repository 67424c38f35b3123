"""Full-size parity property for the configs the CPU oracle cannot finish (C5: N=1e7, dim=1e6, nnz=200, 4e12 posting
visits): a stratified synthetic batch (apss.synth.make_vectors_stratified_dev) whose exact pair scores are elementwise
sums, so the device join is checked at full size without the oracle:
  * every reported pair's score equals its exact dot (<= 1e-5) and is >= theta - 1e-5;
  * every planted (duplicate, source) pair and every pair of duplicates of one source with exact dot >= theta + 1e-5 is
    reported, in both directions;
  * nothing else is reported (random stratified rows share ~nnz/w terms: far below theta);
  * posting visits == sum_t df_t^2, candidate pairs counted.
Usage: python profiles/fullsize_stratified.py [--n N --dim D --nnz K --theta T] > profiles/r01_c5_full.json"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "all-pairs-similarity_amd"))
from apss import synth  # noqa: E402
from apss.engine import ApssIndex  # noqa: E402


def check_pairs(out, n, theta, idx, val, src, q, c, s):
    """the size-independent property: every reported score exact, every planted pair present, nothing else"""
    exact = synth.stratified_dot(idx, val, q, c)
    err = (exact - s.double()).abs().max().item() if q.numel() else 0.0
    out["max_abs_score_error"] = err
    assert err <= 1e-5, err
    assert (exact >= theta - 1e-5).all()
    got = torch.unique(q * n + c)
    assert got.numel() == q.numel(), "a pair was reported twice"
    # expected: (dup, src), (src, dup) and sibling pairs with exact dot >= theta + 1e-5
    rows = (src >= 0).nonzero().flatten()
    a, b = rows, src[rows]
    order = torch.argsort(b, stable=True)  # siblings: duplicates of the same source, adjacent after sorting by source
    sa, sb = a[order], b[order]
    sib_a, sib_b = [], []
    for lag in range(1, 24):
        same = sb[lag:] == sb[:-lag]
        sib_a.append(sa[lag:][same])
        sib_b.append(sa[:-lag][same])
    pa = torch.cat([a] + sib_a)
    pb = torch.cat([b] + sib_b)
    d = synth.stratified_dot(idx, val, pa, pb)
    must = d >= theta + 1e-5
    need = torch.cat([pa[must] * n + pb[must], pb[must] * n + pa[must]])
    missing = need[~torch.isin(need, got)]
    out["planted_pairs_required"] = int(need.numel())
    out["missing"] = int(missing.numel())
    assert missing.numel() == 0, missing[:10]
    may = d >= theta - 1e-5
    allowed = torch.cat([pa[may] * n + pb[may], pb[may] * n + pa[may]])
    extra = got[~torch.isin(got, allowed)]
    out["unexpected"] = int(extra.numel())
    assert extra.numel() == 0, extra[:10]
    return out


def run(n, dim, nnz, theta, seed=20244, log=sys.stderr):
    dev = torch.device("cuda:0")
    t0 = time.time()
    rowptr, idx, val, src = synth.make_vectors_stratified_dev(n, dim, nnz, seed, dev)
    torch.cuda.synchronize()
    print(f"[fullsize] generated {n} x {nnz} on the device in {time.time() - t0:.1f} s", file=log, flush=True)
    df = torch.bincount(idx.flatten().to(torch.int64), minlength=dim).double()
    visits = int((df * df).sum().item())
    ids = torch.arange(n, dtype=torch.int64, device=dev)
    out = {"n": n, "dim": dim, "nnz": nnz, "theta": theta, "postings": n * nnz, "posting_visits_analytic": visits}
    with ApssIndex(dim, theta) as ix:
        ix.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        t0 = time.time()
        cnt = ix.insert_and_query_dev(ids, rowptr, idx.flatten(), val.flatten())
        torch.cuda.synchronize()
        wall = time.time() - t0
        st = ix.stats()
        print(f"[fullsize] join done in {wall:.2f} s wall: {cnt} pairs, probe {st['probe_ms']:.0f} ms in "
              f"{st['probe_launches']} launches, build {st['build_ms']:.0f} ms", file=log, flush=True)
        q, c, s = ix.fetch()
        out.update({k: st[k] for k in ("tiles", "posting_visits", "device_posting_visits", "symmetric", "candidate_pairs", "result_pairs", "probe_ms", "build_ms",
                                       "probe_launches", "filter_survivors", "rescore_ms", "hbm_bytes")})
    out["wall_s"] = wall
    out["scored_pairs_per_s"] = st["candidate_pairs"] / wall
    out["algorithmic_GBps_probe"] = 8.0 * st["device_posting_visits"] / (st["probe_ms"] * 1e-3) / 1e9  # (visits the kernels made)
    assert st["posting_visits"] == visits, (st["posting_visits"], visits)
    return check_pairs(out, n, theta, idx, val, src, torch.from_numpy(q).to(dev), torch.from_numpy(c).to(dev),
                       torch.from_numpy(s).to(dev))


def run_term_shards(n, dim, nnz, theta, T, seed=20244, log=sys.stderr):
    """BASELINE.json configs[3] on ONE GPU: the T term-range shard handles of the batch built one after another, joined
    in-process exactly as apss.dist joins them across GPUs (candidate lists -> union -> exact partial scores -> sum),
    checked at full size by the same property."""
    import numpy as np
    from apss.dist import HipShardEngine, join_shards_local, term_ranges
    dev = torch.device("cuda:0")
    rowptr, idx, val, src = synth.make_vectors_stratified_dev(n, dim, nnz, seed, dev)
    df = torch.bincount(idx.flatten().to(torch.int64), minlength=dim)
    visits = int((df.double() ** 2).sum().item())
    ranges = term_ranges(df.cpu().numpy(), T)
    shared = (rowptr, idx.flatten().contiguous(), val.flatten().contiguous())
    engines = []
    for tr in ranges:
        e = HipShardEngine(dim, theta, tr, dev)
        e.load(None, None, None, device_arrays=shared)
        engines.append(e)
    t0 = time.time()
    q, c, s, n_cand = join_shards_local(engines, n, theta)
    torch.cuda.synchronize()
    wall = time.time() - t0
    out = {"n": n, "dim": dim, "nnz": nnz, "theta": theta, "term_shards": T, "term_ranges": ranges, "wall_s": wall,
           "candidates_per_shard": n_cand, "result_pairs": int(len(q)),
           "probe_ms_per_shard": [e.stats["probe_ms"] for e in engines],
           "posting_visits_per_shard": [e.stats["posting_visits"] for e in engines]}
    print(f"[fullsize] {T} term shards joined in {wall:.2f} s: {len(q)} pairs, candidates per shard {n_cand}", file=log, flush=True)
    assert sum(out["posting_visits_per_shard"]) == visits, (out["posting_visits_per_shard"], visits)
    return check_pairs(out, n, theta, idx, val, src, torch.from_numpy(np.asarray(q, np.int64)).to(dev),
                       torch.from_numpy(np.asarray(c, np.int64)).to(dev), torch.from_numpy(np.asarray(s, np.float32)).to(dev))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=synth.CONFIGS["c5"]["n"])
    ap.add_argument("--dim", type=int, default=synth.CONFIGS["c5"]["dim"])
    ap.add_argument("--nnz", type=int, default=synth.CONFIGS["c5"]["nnz"])
    ap.add_argument("--theta", type=float, default=synth.CONFIGS["c5"]["theta"])
    ap.add_argument("--term-shards", type=int, default=0, help="T > 1: the term-sharded join of configs[3] on one GPU")
    a = ap.parse_args()
    print(json.dumps(run_term_shards(a.n, a.dim, a.nnz, a.theta, a.term_shards) if a.term_shards > 1
                     else run(a.n, a.dim, a.nnz, a.theta)))
