#!/usr/bin/env python3
"""Latency harness, the counterpart of the reference's benchmark driver (core/src/main/scala/cpslab/benchmark/
LoadGenerator.scala): after a warm-up phase that indexes the data set, `totalMessageCount` vectors are sent ONE per
message, one every `writeBatchingDuration` ms (LoadRunner, LoadGenerator.scala:58-74; defaults 50 messages, 50 ms:
conf/app_client_template.conf:19-26); the test phase replays the data set from its start under fresh ids
(`videos(msgCount % videos.size)`, LoadGenerator.scala:29-40), so every message finds at least its own earlier copy.  The
response time of a vector is the moment its SimilarityOutput is available minus the moment it was sent
(StartTime.moment -> SimilarityOutput.outputMoment, Message.scala:20-21,43; LoadGenerator.scala:135-149); avg / max / min
are reported like LoadGenerator.postStop (LoadGenerator.scala:112-132), plus p50 / p99.

Backends: `gpu` = the C ABI (host pointers in, triples out: PCIe and the host mirror's work included); `refcpu` = the CPU
oracle's IndexingWorkerActor restatement (hash-map posting lists + per-candidate hash-map dot, one thread = one actor) --
the reported baseline of this harness, as `cpu_baseline` is of bench.py; never a fallback of the product.
Shapes: `production` = conf/app.conf:6,15 (vectorDim 1048576, similarityThreshold 0.7); `template` =
conf/app_client_template.conf (vectorDim 1024, threshold 0.7); `c3` = BASELINE.json configs[2]'s dim and theta.
The CC_WEB_VIDEO feature file is not part of the reference repo: vectors are synthetic (apss/synth.py), L2-normalised as
LoadRunner does (LoadGenerator.scala:34-37).  Not BASELINE.json's metric: self-consistency only.
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

SHAPES = {
    "production": dict(dim=1 << 20, theta=0.7, nnz=100),
    "template": dict(dim=1024, theta=0.7, nnz=40),
    "c3": dict(dim=100_000, theta=0.8, nnz=100),
}


def summarize(lat_ms, found):
    lat = np.asarray(lat_ms)
    return {"avg_ms": float(lat.mean()), "max_ms": float(lat.max()), "min_ms": float(lat.min()),
            "p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)), "pairs_found": int(found)}


def run(handler, warmup, preload, messages, interval_ms, rp, idx, val):
    """handler(ids, rowptr, indices, values) -> number of (q, c) pairs: one IndexData message; warmup(...) indexes the data
    set (the reference's warm-up phase, whose answers nobody reads).  Returns (latencies, pairs)."""
    e0 = int(rp[preload])
    warmup(np.arange(preload, dtype=np.int64), rp[:preload + 1], idx[:e0], val[:e0])
    lat, found = [], 0
    next_send = time.perf_counter()
    for m in range(messages):
        r = m % preload                                              # videos(msgCount % videos.size)
        s0, s1 = int(rp[r]), int(rp[r + 1])
        now = time.perf_counter()
        if interval_ms > 0 and now < next_send:
            time.sleep(next_send - now)
        t0 = time.perf_counter()                                     # StartTime(vectorId, now)
        found += handler(np.array([preload + m], np.int64), np.array([0, s1 - s0], np.int64), idx[s0:s1], val[s0:s1])
        lat.append((time.perf_counter() - t0) * 1e3)                 # outputMoment - startMoment
        next_send = t0 + interval_ms / 1e3
    return lat, found


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="both", choices=["gpu", "refcpu", "both"])
    ap.add_argument("--shape", default="production", choices=sorted(SHAPES))
    ap.add_argument("--preload", type=int, default=200_000, help="vectors indexed in the warm-up phase")
    ap.add_argument("--messages", type=int, default=50, help="benchmark.totalMessageCount")
    ap.add_argument("--interval-ms", type=float, default=50.0, help="benchmark.writeBatchingDuration (0 = back to back)")
    ap.add_argument("--zipf", type=float, default=0.0, help="term distribution of the synthetic vectors")
    ap.add_argument("--theta", type=float, default=None, help="override the shape's similarityThreshold (0: what "
                    "conf/app_server_template.conf:17 ships -- every vector sharing a term is an answer)")
    a = ap.parse_args()
    from apss import synth
    shp = SHAPES[a.shape]
    dim, theta, nnz = shp["dim"], shp["theta"] if a.theta is None else a.theta, min(shp["nnz"], shp["dim"])
    rp, idx, val = synth.make_vectors(a.preload, dim, nnz, a.zipf, seed=5)
    out = {"metric": "single-vector IndexData response time (ms), LoadGenerator-style", "messages": a.messages,
           "interval_ms": a.interval_ms, "index_size": a.preload,
           "config": {"shape": a.shape, "dim": dim, "nnz": nnz, "theta": theta, "zipf_s": a.zipf}}
    if a.backend in ("gpu", "both"):
        from apss.engine import ApssIndex
        ix = ApssIndex(dim, theta)

        def gpu_handler(ids, rowptr, indices, values):
            return ix.insert_and_query(ids, rowptr, indices, values)[0].size

        lat, found = run(gpu_handler, ix.insert, a.preload, a.messages, a.interval_ms, rp, idx, val)
        out["gpu"] = summarize(lat, found)
        st = ix.stats()
        out["gpu"].update({"last_probe_kernel_ms": st["probe_ms"], "last_build_ms": st["build_ms"], "head_terms": st["head_terms"]})
        ix.close()
    if a.backend in ("refcpu", "both"):
        from oracle import oracle  # the harness's CPU baseline (test infrastructure), never used by the product path
        w = oracle.Worker(dim, theta)

        def cpu_handler(ids, rowptr, indices, values):
            return w.index_data(ids, rowptr, indices, values)[0].size

        lat, found = run(cpu_handler, lambda *b: w.index_data(*b, build_only=True), a.preload, a.messages, a.interval_ms, rp, idx, val)
        out["refcpu"] = summarize(lat, found)
        out["refcpu"]["note"] = "oracle restatement of one IndexingWorkerActor (IWA:61-111 + CU:98-117), one thread"
        w.close()
    if "gpu" in out and "refcpu" in out:
        assert out["gpu"]["pairs_found"] == out["refcpu"]["pairs_found"], "the two backends disagree on the result size"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
