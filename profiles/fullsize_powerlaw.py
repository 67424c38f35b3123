"""Full-size parity property for BASELINE.json configs[4] ("N=10M, dim=1M, nnz=200 power-law, theta=0.9"): a Zipf(1) batch
with planted near-duplicates (apss.synth.make_vectors_zipf_dev) joined on one GPU -- dense-head block (256 terms + the
folded block) on the matrix cores, the tail through the sparse filter, exact rescoring -- and checked WITHOUT the oracle
(which needs hours at this size) by direct float64 dot products of row pairs:
  * every reported pair's score equals its exact dot (<= 1e-5) and is >= theta - 1e-5, no pair twice;
  * every planted (duplicate, source) pair and every pair of duplicates of one source whose exact dot is >= theta + 1e-5
    is reported, in both directions.
What this cannot see -- a true pair that is neither reported nor planted -- is what the reduced-size tests check against
the oracle (tests/test_gpu_head.py, tests/test_gpu_shard_head.py).
Usage: python profiles/fullsize_powerlaw.py [--n N --dim D --nnz K --theta T --head-terms H] > profiles/r03_c5z_full.json"""
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
    exact = synth.rows_dot(idx, val, q, c)
    err = (exact - s.double()).abs().max().item() if q.numel() else 0.0
    out["max_abs_score_error"] = err
    assert err <= 1e-5, err
    assert (exact >= theta - 1e-5).all()
    got = torch.unique(q * n + c)
    assert got.numel() == q.numel(), "a pair was reported twice"
    rows = (src >= 0).nonzero().flatten()
    a, b = rows, src[rows]
    order = torch.argsort(b, stable=True)  # siblings: duplicates of the same source, adjacent after sorting by source
    sa, sb = a[order], b[order]
    sib_a, sib_b = [], []
    for lag in range(1, 24):
        same = sb[lag:] == sb[:-lag]
        sib_a.append(sa[lag:][same])
        sib_b.append(sa[:-lag][same])
    pa, pb = torch.cat([a] + sib_a), torch.cat([b] + sib_b)
    d = synth.rows_dot(idx, val, pa, pb)
    must = d >= theta + 1e-5
    need = torch.cat([pa[must] * n + pb[must], pb[must] * n + pa[must]])
    missing = need[~torch.isin(need, got)]
    out["planted_pairs_required"] = int(need.numel())
    out["missing"] = int(missing.numel())
    assert missing.numel() == 0, missing[:10]
    out["reported_not_planted"] = int((~torch.isin(got, torch.cat([pa * n + pb, pb * n + pa]))).sum().item())  # real pairs all the same (checked above)
    return out


def run(n, dim, nnz, theta, seed=20245, head_terms=0, zipf_s=1.0, log=sys.stderr):
    dev = torch.device("cuda:0")
    t0 = time.time()
    rowptr, idx, val, src = synth.make_vectors_zipf_dev(n, dim, nnz, zipf_s, seed, dev, return_src=True)
    torch.cuda.synchronize()
    print(f"[powerlaw] generated {n} x {nnz} Zipf({zipf_s}) rows on the device in {time.time() - t0:.1f} s", file=log, flush=True)
    df = torch.bincount(idx.flatten().to(torch.int64), minlength=dim).double()
    out = {"n": n, "dim": dim, "nnz": nnz, "theta": theta, "zipf_s": zipf_s, "postings": n * nnz,
           "posting_visits_all_sparse": float((df * df).sum().item()), "pairs_n_squared": float(n) * float(n - 1)}
    ids = torch.arange(n, dtype=torch.int64, device=dev)
    with ApssIndex(dim, theta, head_terms=head_terms) as ix:
        ix.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        t0 = time.time()
        cnt = ix.insert_and_query_dev(ids, rowptr, idx.flatten(), val.flatten())
        torch.cuda.synchronize()
        wall = time.time() - t0
        st = ix.stats()
        print(f"[powerlaw] join done in {wall:.2f} s wall: {cnt} pairs, head {st['head_ms']:.0f} ms ({st['head_terms']} terms), "
              f"tail probe {st['probe_ms']:.0f} ms, build {st['build_ms']:.0f} ms", file=log, flush=True)
        q, c, s = ix.fetch()
        out.update({k: st[k] for k in ("tiles", "posting_visits", "device_posting_visits", "symmetric", "candidate_pairs", "result_pairs", "probe_ms", "build_ms",
                                       "probe_launches", "filter_survivors", "rescore_ms", "hbm_bytes", "head_terms", "head_pairs",
                                       "head_survivors", "head_ms", "head_flops", "head_int8", "probe_kernel")})
    out["wall_s"] = wall
    out["scored_pairs_per_s"] = max(st["candidate_pairs"], st["head_pairs"]) / wall
    # (INT8 rendering: integer operations against the int8 MFMA peak, 2 x the bf16 peak)
    out["head_frac_of_mfma_peak"] = st["head_flops"] / (st["head_ms"] * 1e-3) / (5.0e15 if st["head_int8"] else 2.5e15) if st["head_ms"] > 0 else None
    out["tail_algorithmic_frac"] = 8.0 * st["device_posting_visits"] / (st["probe_ms"] * 1e-3) / 8e12 if st["probe_ms"] > 0 else None  # (visits the kernels made)
    return check_pairs(out, n, theta, idx, val, src, torch.from_numpy(q).to(dev), torch.from_numpy(c).to(dev),
                       torch.from_numpy(s).to(dev))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=1_000_000)
    ap.add_argument("--nnz", type=int, default=200)
    ap.add_argument("--theta", type=float, default=0.9)
    ap.add_argument("--head-terms", type=int, default=0)
    a = ap.parse_args()
    print(json.dumps(run(a.n, a.dim, a.nnz, a.theta, head_terms=a.head_terms)))
