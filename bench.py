#!/usr/bin/env python3
"""bench.py -- the reference's hot path (inverted-index probe + candidate scoring + threshold prune) on MI355X.

A step is one pass of the hot path over one batch: the IndexData handler on the whole synthetic batch
(IndexingWorkerActor.scala:123-137 = build the index from the batch, then query the batch against it: a
single-batch self-join), inputs already resident in HBM.  Workload = BASELINE.json configs[2] "Synthetic N=1M,
dim=100k, nnz=100, cosine theta=0.8" (the configuration the metric is quoted on); --workload picks another.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (term-range shards, RCCL)

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "all-pairs-similarity_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # ... 6.29 TB/s measured float4 copy
BYTES_PER_VISIT = 8            # one posting = int32 slot + fp32 weight


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", help="c2 | c3 | c3z | c5s | c5 (BASELINE.json configs; c3 is the metric's)")
    ap.add_argument("--n", type=int, default=None, help="override the number of vectors (debug)")
    ap.add_argument("--tile-rows", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--term-shards", type=int, default=None, help="T of the T x D rank grid: term-range shards per candidate range (default 1)")
    ap.add_argument("--solo", default=None, help="T,D,i,j: time shard (term i of T, rows j of D) alone on this GPU (projection)")
    ap.add_argument("--no-term-row", action="store_true", help="skip the term-sharded comparison row of multi-GPU runs")
    ap.add_argument("--term-row-deadline", type=float, default=120.0, help="seconds allowed for the term-sharded comparison row")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the real run) | gloo (rehearsal: all ranks share GPU 0)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU work budget of the baseline sample")
    return ap.parse_args()


def cpu_baseline(cfg, rp, idx, val, budget_s):
    """The oracle's restatement of the reference algorithm ("port": posting lists + the per-candidate hash-map dot
    of CommonUtils.scala:98-117 in double), timed on this box's host cores on a bounded query sample."""
    from oracle import oracle
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(avail, 16)  # one GPU's share of the host (the box exposes every core of the machine)
    n = len(rp) - 1
    pilot_q = min(n, cores * 2)
    p = oracle.selfjoin_sample(0, cfg["dim"], cfg["theta"], rp, idx, val, 0, pilot_q, cores)
    per_q = max(p["seconds"] / max(pilot_q, 1), 1e-7)
    sample = int(min(n, max(pilot_q, budget_s / per_q)))
    r = oracle.selfjoin_sample(0, cfg["dim"], cfg["theta"], rp, idx, val, 0, sample, cores)
    opt = oracle.selfjoin_sample(1, cfg["dim"], cfg["theta"], rp, idx, val, 0, min(n, sample * 20), cores)
    return {
        "value": r["cand_pairs"] / r["seconds"], "unit": "scored candidate pairs/s", "cores": cores, "kind": "port",
        "sample": "all %d vectors indexed, first %d queries timed (%.1f s, %d threads); reference algorithm: posting "
                  "lists + per-candidate HashMap dot in double" % (n, sample, r["seconds"], cores),
        "optimised_cpu_value": opt["cand_pairs"] / opt["seconds"],
        "optimised_cpu_note": "fairness bracket: CSC + dense double accumulator, %d threads" % cores,
    }


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from apss import synth
    from apss.engine import ApssIndex

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % a.gpus)
    if a.backend == "gloo":
        local_rank = 0  # rehearsal on a one-GPU box: every rank drives GPU 0, collectives on CPU tensors
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm_dev = torch.device("cpu") if a.backend == "gloo" else dev
    if world > 1:
        if a.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    cfg = dict(synth.CONFIGS[a.workload])
    if a.n:
        cfg["n"] = a.n
    t0 = time.time()
    rp, idx, val = synth.make_vectors(cfg["n"], cfg["dim"], cfg["nnz"], cfg["zipf_s"], cfg["seed"])
    gen_s = time.time() - t0
    n = cfg["n"]

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if a.solo:
        from apss.dist import HipShardEngine, term_ranges
        T, D, ti, dj = (int(x) for x in a.solo.split(","))
        tr = term_ranges(np.bincount(idx, minlength=cfg["dim"]), T)[ti]
        eng = HipShardEngine(cfg["dim"], cfg["theta"], tr, dev, a.tile_rows)
        eng.load(rp, idx, val, None if D == 1 else (n * dj // D, n * (dj + 1) // D))
        step = eng.join if T == 1 else (lambda: int(eng.candidates()[0].numel()))  # term shards hand their candidates on
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            n_c = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print(json.dumps({"solo_shard": a.solo, "ms_per_step": dt * 1e3, "probe_kernel_ms": eng.stats["probe_ms"],
                          "build_ms": eng.stats["build_ms"], "posting_visits": eng.stats["posting_visits"],
                          "candidates": int(n_c)}))
        return
    if world == 1:
        d_rp = torch.from_numpy(rp).to(dev)
        d_idx = torch.from_numpy(idx).to(dev)
        d_val = torch.from_numpy(val.astype(np.float32)).to(dev)
        d_ids = torch.arange(n, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()  # the library runs on its own stream
        ix = ApssIndex(cfg["dim"], cfg["theta"], device=local_rank, tile_rows=a.tile_rows, capacity_rows=n,
                       capacity_nnz=idx.size)

        def step():
            ix.clear()
            return ix.insert_and_query_dev(d_ids, d_rp, d_idx, d_val)

        probe_ms, build_ms = [], []
        for _ in range(a.warmup):
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            n_pairs = step()
            st = ix.stats()
            probe_ms.append(st["probe_ms"])
            build_ms.append(st["build_ms"])
        sync()
        dt = time.perf_counter() - t0
        visits, cands = st["posting_visits"], st["candidate_pairs"]
        launches = st["probe_launches"]
        parallelism = "1 GPU"
        extra = {}
        hung = False
    else:
        from apss.dist import ShardedJoin
        # distinct scored candidate pairs of the workload (a pair sharing terms in k shards is touched by k shards):
        # counted once, untimed, by a plain single-GPU join on rank 0
        distinct = torch.zeros(1, dtype=torch.float64, device=comm_dev)
        if rank == 0:
            ix0 = ApssIndex(cfg["dim"], cfg["theta"], device=local_rank, tile_rows=a.tile_rows)
            ix0.set_stream(torch.cuda.current_stream(dev).cuda_stream)
            ix0.insert_and_query_dev(torch.arange(n, dtype=torch.int64, device=dev), torch.from_numpy(rp).to(dev),
                                     torch.from_numpy(idx).to(dev), torch.from_numpy(val.astype(np.float32)).to(dev))
            distinct[0] = float(ix0.stats()["candidate_pairs"])
            ix0.close()
            del ix0
            torch.cuda.empty_cache()
        dist.all_reduce(distinct)
        sj = ShardedJoin(cfg["dim"], cfg["theta"], rank, world, dev, tile_rows=a.tile_rows, comm_device=comm_dev,
                         term_shards=a.term_shards)
        sj.load(rp, idx, val)
        probe_ms, build_ms = [], []
        for _ in range(a.warmup):
            sj.step()
        sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            n_pairs = sj.step()
            probe_ms.append(sj.last["probe_ms"])
            build_ms.append(sj.last["build_ms"])
        sync()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        pm = torch.tensor([float(np.mean(probe_ms))], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(pm, op=dist.ReduceOp.MAX)
        probe_ms = [float(pm.item())]  # slowest shard's probe kernel
        visits, cands, launches = sj.last["posting_visits"], int(distinct.item()), world
        # comparison row, measured in the same run: the term-sharded layout of BASELINE.json configs[3] (T = 2 term
        # ranges per candidate range, candidate all-gather + RCCL all-reduce of partial scores); not part of `value`
        term_row = None
        hung = False
        if world % 2 == 0 and (a.term_shards or 1) == 1 and not a.no_term_row:
            del sj.engine
            torch.cuda.empty_cache()

            def term_sharded_step():
                torch.cuda.set_device(dev)
                sj2 = ShardedJoin(cfg["dim"], cfg["theta"], rank, world, dev, tile_rows=a.tile_rows, comm_device=comm_dev,
                                  term_shards=2)
                sj2.load(rp, idx, val)
                sj2.step()
                sync()
                t1 = time.perf_counter()
                n2 = sj2.step()
                sync()
                t2 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=comm_dev)
                dist.all_reduce(t2, op=dist.ReduceOp.MAX)
                return {"grid": "2 term ranges x %d candidate ranges" % (world // 2), "ms_per_step": float(t2.item()) * 1e3,
                        "result_pairs": int(n2), "exchange": sj2.last["exchange"]}

            # the comparison row must never take the benchmark down: it runs under a deadline on every rank; a rank
            # that raises or stalls (its peers then wait in a collective) reports the fact and the main line still prints
            box = {}

            def guarded():
                try:
                    box["row"] = term_sharded_step()
                except Exception as e:
                    box["row"] = {"error": repr(e)[:200]}

            th = threading.Thread(target=guarded, daemon=True)
            th.start()
            th.join(a.term_row_deadline)
            hung = th.is_alive()
            term_row = {"error": "no answer within %g s" % a.term_row_deadline} if hung else box.get("row")
        parallelism = ("%d candidate ranges, no data-path collective (term-sharded layouts: --term-shards)" % sj.D
                       if sj.T == 1 else "%d term-range shards x %d candidate ranges; per term group: candidate "
                       "all-gather + RCCL all-reduce of partial scores" % (sj.T, sj.D))
        extra = {"exchange": sj.last.get("exchange"), "shard_touched_pairs_sum": sj.last["candidate_pairs"],
                 "backend": a.backend, "term_sharded_comparison": term_row}

    if rank != 0:
        if world > 1 and hung:
            os._exit(0)  # a stalled comparison step still holds a collective: do not wait for it
        if world > 1:
            dist.destroy_process_group()
        return

    sec_per_step = dt / a.steps
    kern_s = float(np.mean(probe_ms)) * 1e-3
    alg_bytes = BYTES_PER_VISIT * visits
    out = {
        "metric": "scored candidate pairs/sec + achieved HBM GB/s, N=1M d=100k nnz=100, 1/2/4/8 GPU",
        "value": cands / sec_per_step,
        "unit": "scored candidate pairs/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": sec_per_step * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if world > 1 else "weak",
        "vs_baseline": None,
        "dtype": "u16+f32",  # 16-bit fixed-point filter sums, fp32 exact rescoring of the survivors
        "data": "synthetic",
        "config": {"workload": "%s: N=%d dim=%d nnz=%d %s theta=%g, single-batch self-join (build + probe)" % (
            a.workload, n, cfg["dim"], cfg["nnz"], "Zipf(%g)" % cfg["zipf_s"] if cfg["zipf_s"] else "uniform",
            cfg["theta"]), "parallelism": parallelism, "tile_rows": a.tile_rows or 16384},
        "posting_visits_per_step": visits,
        "candidate_pairs_per_step": cands,
        "result_pairs_per_step": int(n_pairs),
        "posting_visits_per_s": visits / sec_per_step,
        "algorithmic_GBps_whole_step": alg_bytes / sec_per_step / 1e9,
        "build_ms": float(np.mean(build_ms)),
        "probe_kernel_ms": float(np.mean(probe_ms)),
        "datagen_s": gen_s,
        "roofline": {
            "bound": "hbm", "kernel": "k_probe",
            "achieved": alg_bytes / launches / kern_s / 1e9 if kern_s > 0 else None,
            "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (alg_bytes / launches / kern_s / 1e9) / HBM_PEAK_GBS if kern_s > 0 else None,
            "frac_of_measured_copy_peak": (alg_bytes / launches / kern_s / 1e9) / HBM_COPY_GBS if kern_s > 0 else None,
            "traffic": None,
            "note": "achieved = 8 B x posting visits per launch / HIP-event kernel time on the launch stream",
        },
    }
    # HBM-side bytes per probe launch: rocprofv3 PMC passes cannot run inside this process; the committed summary of
    # the same command (profiles/collect_r01.sh -> profiles/r01_probe_traffic.json) is quoted when the workload matches
    tpath = os.path.join(ROOT, "profiles", "r01_probe_traffic.json")
    if world == 1 and a.workload == "c3" and not a.n and not a.tile_rows and os.path.exists(tpath):
        tj = json.load(open(tpath))
        if tj.get("algorithmic_bytes_per_launch") == alg_bytes:
            out["roofline"]["traffic"] = tj["traffic_bytes_per_launch_corrected"]
            out["roofline"]["traffic_note"] = ("bytes per launch, (2 x FETCH_SIZE + WRITE_SIZE) x 1024 from separate rocprofv3 "
                                               "--pmc passes of this command (profiles/r01_probe_traffic.json); gfx950 tallies "
                                               "128-B read requests at 64 B; Infinity-Cache hits are included")
    out.update(extra)
    if not a.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(cfg, rp, idx, val, a.cpu_seconds)
    elif not a.no_cpu_baseline:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)
    if world > 1 and hung:
        os._exit(0)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
