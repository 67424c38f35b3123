#!/usr/bin/env python3
"""Streaming harness: the reference's own operating mode -- a stream of IndexData batches.

WriteWorkerActor flushes a DataPacket every `ioTriggerPeriod` ms (WriteWorkerActor.scala:164-183) and the worker builds,
then queries, per batch (IndexingWorkerActor.scala:123-134): batch b sees itself and every batch before it, earlier vectors
are never re-queried.  Here the store grows 0 -> N through `apss_insert_and_query_dev` in batches of B
vectors already resident in HBM; per batch size the line reports vectors/s, scored candidate pairs/s, the posting visits
the device made (the sum over the batches is ~ sum df^2 / 2 + the within-batch half: the same work as the symmetric
whole-store join), and where a batch's time goes (index build, probe kernels, the rest = ingest + host + syncs).
Small batches (B < --full-below) are SAMPLED: the store is pre-filled in bulk to each of --levels fill levels and
--sample-calls batches are timed there; totals are the trapezoid over the levels, and marked "sampled".

`refcpu` beside it: the oracle's IndexingWorkerActor restatement (hash-map posting lists + per-candidate hash-map dot, one
thread = one actor) fed the same batches for a bounded number of seconds -- the harness's baseline, never a fallback.
Not BASELINE.json's metric (bench.py is); workload = configs[2]'s shape (C3) unless told otherwise.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "all-pairs-similarity_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402


def stream_gpu(cfg, d_ids, d_rp, d_idx, d_val, B, n_end, ix=None, start=0, max_calls=None):
    """insert-and-query rows [start, n_end) in batches of B; returns per-call records"""
    import torch
    rec = []
    nnz = cfg["nnz"]
    calls = 0
    for b0 in range(start, n_end, B):
        b1 = min(n_end, b0 + B)
        rp = (d_rp[b0:b1 + 1] - d_rp[b0]).contiguous()
        sl = slice(b0 * nnz, b1 * nnz)
        ids, bi, bv = d_ids[b0:b1], d_idx[sl], d_val[sl]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_res = ix.insert_and_query_dev(ids, rp, bi, bv)
        dt = time.perf_counter() - t0
        st = ix.stats()
        rec.append((b1 - b0, dt, st["probe_ms"] + st["rescore_ms"] + st["head_ms"], st["build_ms"], st["device_posting_visits"], st["candidate_pairs"], n_res))
        calls += 1
        if max_calls and calls >= max_calls:
            break
    return rec


def summarize(rec):
    a = np.array(rec, dtype=np.float64)
    rows, wall = a[:, 0].sum(), a[:, 1].sum()
    return {"calls": int(len(rec)), "vectors": int(rows), "wall_s": wall, "vectors_per_s": rows / wall,
            "ms_per_call_mean": float(a[:, 1].mean() * 1e3), "ms_per_call_p50": float(np.percentile(a[:, 1], 50) * 1e3),
            "ms_per_call_p99": float(np.percentile(a[:, 1], 99) * 1e3),
            "probe_kernels_ms_sum": float(a[:, 2].sum()), "build_ms_sum": float(a[:, 3].sum()),
            "rest_ms_sum": float(wall * 1e3 - a[:, 2].sum() - a[:, 3].sum()),
            "device_posting_visits": float(a[:, 4].sum()), "candidate_pairs": float(a[:, 5].sum()), "result_pairs": int(a[:, 6].sum()),
            "candidate_pairs_per_s": a[:, 5].sum() / wall}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--rows", type=int, default=None)
    ap.add_argument("--batches", default="1,64,1024,16384")
    ap.add_argument("--full-below", type=int, default=64, help="batch sizes below this are sampled at fill levels instead of streamed whole")
    ap.add_argument("--levels", type=int, default=5)
    ap.add_argument("--sample-calls", type=int, default=1500)
    ap.add_argument("--refcpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-refcpu", action="store_true")
    a = ap.parse_args()
    import torch
    from apss import synth
    from apss.engine import ApssIndex
    cfg = dict(synth.CONFIGS[a.workload])
    if a.rows:
        cfg["n"] = a.rows
    n, dim, theta = cfg["n"], cfg["dim"], cfg["theta"]
    rp, idx, val = synth.make_vectors(n, dim, cfg["nnz"], cfg["zipf_s"], cfg["seed"])
    dev = torch.device("cuda", 0)
    d_rp, d_idx, d_val = torch.from_numpy(rp).to(dev), torch.from_numpy(idx).to(dev), torch.from_numpy(val.astype(np.float32)).to(dev)
    d_ids = torch.arange(n, dtype=torch.int64, device=dev)
    df2 = float(synth.workload_counts(dim, rp, idx)[1])
    out = {"metric": "streaming IndexData batches (store grows 0 -> N): vectors/s and scored candidate pairs/s per batch size",
           "config": {"workload": "%s: N=%d dim=%d nnz=%d theta=%g" % (a.workload, n, dim, cfg["nnz"], theta)},
           "posting_visits_whole_store_join": df2, "posting_visits_stream_expected": "~ (sum df^2 + postings) / 2 = %.4g" % ((df2 + idx.size) / 2)}
    # the whole store as ONE batch, for scale (bench.py's step)
    with ApssIndex(dim, theta, capacity_rows=n, capacity_nnz=idx.size) as ix:
        rec = stream_gpu(cfg, d_ids, d_rp, d_idx, d_val, n, n, ix)
        ix.clear()
        rec = stream_gpu(cfg, d_ids, d_rp, d_idx, d_val, n, n, ix)
    out["whole_store_one_batch"] = summarize(rec)
    whole_pairs = rec[0][6]
    rows = {}
    for B in [int(x) for x in a.batches.split(",")]:
        if B >= a.full_below:
            with ApssIndex(dim, theta, capacity_rows=n, capacity_nnz=idx.size) as ix:
                rec = stream_gpu(cfg, d_ids, d_rp, d_idx, d_val, B, n, ix)
            s = summarize(rec)
            s["mode"] = "streamed whole"
            # every unordered pair is found once by the later vector (+ both directions inside a batch): between half and all
            s["result_pairs_vs_whole_store_join"] = s["result_pairs"] / max(1, whole_pairs)
            quart = len(rec) // 4
            s["ms_per_call_by_fill_quartile"] = [float(np.mean([r[1] for r in rec[i * quart:(i + 1) * quart]]) * 1e3) for i in range(4)] if quart else None
        else:
            levels = [int(n * (i + 0.5) / a.levels) for i in range(a.levels)]
            per_level = []
            with ApssIndex(dim, theta, capacity_rows=n, capacity_nnz=idx.size) as ix:
                at = 0
                for lv in levels:
                    if lv > at:
                        stream_gpu(cfg, d_ids, d_rp, d_idx, d_val, 65536, lv, ix, start=at)  # bulk pre-fill (untimed)
                    rec = stream_gpu(cfg, d_ids, d_rp, d_idx, d_val, B, n, ix, start=lv, max_calls=a.sample_calls)
                    at = lv + sum(r[0] for r in rec)
                    ss = summarize(rec)
                    ss["store_rows"] = lv
                    per_level.append(ss)
            wall_per_vec = float(np.mean([1.0 / p["vectors_per_s"] for p in per_level]))
            s = {"mode": "sampled: %d calls at each of %d fill levels" % (a.sample_calls, a.levels), "levels": per_level,
                 "vectors_per_s": 1.0 / wall_per_vec, "wall_s_extrapolated": wall_per_vec * n,
                 "ms_per_call_mean": float(np.mean([p["ms_per_call_mean"] for p in per_level])),
                 "ms_per_call_p50": float(np.mean([p["ms_per_call_p50"] for p in per_level])),
                 "candidate_pairs_per_s": float(np.mean([p["candidate_pairs_per_s"] for p in per_level]))}
        rows[str(B)] = s
    out["batch_size"] = rows
    if not a.no_refcpu:
        from oracle import oracle  # the harness's CPU baseline (test infrastructure), never used by the product path
        ref = {}
        for B in [int(x) for x in a.batches.split(",")]:
            w = oracle.Worker(dim, theta)
            t0, rows_done, pairs = time.perf_counter(), 0, 0
            # the reference's cost grows with the store: feed it from an empty store for a bounded time, report where it got to
            for b0 in range(0, n, B):
                b1 = min(n, b0 + B)
                sl = slice(rp[b0], rp[b1])
                q, c, s_ = w.index_data(np.arange(b0, b1), rp[b0:b1 + 1] - rp[b0], idx[sl], val[sl])
                rows_done, pairs = b1, pairs + q.size
                if time.perf_counter() - t0 > a.refcpu_seconds:
                    break
            dt = time.perf_counter() - t0
            ref[str(B)] = {"vectors": rows_done, "wall_s": dt, "vectors_per_s": rows_done / dt, "result_pairs": int(pairs),
                           "note": "from an empty store, stopped after %.0f s at %d stored vectors (its cost per vector grows with the store)" % (a.refcpu_seconds, rows_done)}
            w.close()
        out["refcpu"] = {"rows": ref, "note": "oracle restatement of one IndexingWorkerActor (IWA:61-111 + CU:98-117), one thread"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
