#!/usr/bin/env python3
"""Latency harness, the counterpart of the reference's benchmark driver (core/src/main/scala/cpslab/benchmark/
LoadGenerator.scala:58-74, 112-158): vectors are inserted ONE per message at a fixed cadence; the response time of a
vector is the moment its SimilarityOutput is available minus the moment it was sent (StartTime.moment ->
SimilarityOutput.outputMoment, Message.scala:20-21,43); avg / max / min are printed at the end like LoadGenerator.postStop.

The index is pre-loaded with `--preload` vectors (the reference's warm-up phase), then `--messages` single-vector
IndexData batches go through the C ABI (host pointers, PCIe included).  Not BASELINE.json's metric: self-consistency only.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preload", type=int, default=200_000)
    ap.add_argument("--messages", type=int, default=200)
    ap.add_argument("--dim", type=int, default=100_000)
    ap.add_argument("--nnz", type=int, default=100)
    ap.add_argument("--theta", type=float, default=0.8)
    ap.add_argument("--interval-ms", type=float, default=0.0, help="writeBatchingDuration (0 = back to back)")
    a = ap.parse_args()
    from apss import synth
    from apss.engine import ApssIndex
    n = a.preload + a.messages
    rp, idx, val = synth.make_vectors(n, a.dim, a.nnz, 0.0, seed=5)
    ids = np.arange(n, dtype=np.int64)
    ix = ApssIndex(a.dim, a.theta)
    e0 = int(rp[a.preload])
    ix.insert(ids[:a.preload], rp[:a.preload + 1], idx[:e0], val[:e0])
    lat, found = [], 0
    for m in range(a.messages):
        r = a.preload + m
        s0, s1 = int(rp[r]), int(rp[r + 1])
        t0 = time.perf_counter()                                      # StartTime(vectorId, now)
        q, c, s = ix.insert_and_query(ids[r:r + 1], [0, s1 - s0], idx[s0:s1], val[s0:s1])
        lat.append((time.perf_counter() - t0) * 1e3)                  # outputMoment - startMoment
        found += q.size
        if a.interval_ms > 0:
            time.sleep(a.interval_ms / 1e3)
    lat = np.array(lat)
    print(json.dumps({"metric": "single-vector IndexData response time (ms), LoadGenerator-style", "avg_ms": float(lat.mean()),
                      "max_ms": float(lat.max()), "min_ms": float(lat.min()), "p50_ms": float(np.median(lat)),
                      "messages": a.messages, "index_size": a.preload, "pairs_found": int(found),
                      "config": {"dim": a.dim, "nnz": a.nnz, "theta": a.theta}}))


if __name__ == "__main__":
    main()
