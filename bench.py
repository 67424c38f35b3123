#!/usr/bin/env python3
"""bench.py -- the reference's hot path (inverted-index probe + candidate scoring + threshold prune) on MI355X.

A step is one pass of the hot path over one batch: the IndexData handler on the whole synthetic batch
(IndexingWorkerActor.scala:123-137 = build the index from the batch, then query the batch against it: a
single-batch self-join), inputs already resident in HBM.  Workload = BASELINE.json configs[2] "Synthetic N=1M,
dim=100k, nnz=100, cosine theta=0.8" (the configuration the metric is quoted on); --workload picks another
(c2 | c3 | c3z | c3z1 | c5s | c5 | c5z: apss/synth.py).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...     (one process per GPU: apss/dist.py)
    python bench.py --gpus N ...                  (typed without a launcher: starts the line above as a child, relays its line)
    python bench.py --gpus N --engine group ...   (ONE process: apss_group, the sharded index + RCCL exchange behind the C ABI)

N > 1 times BOTH layouts of SURVEY.md 8(e), K steps each, same barriers: the contract's (term-range shards across the N
GPUs, RCCL all-gather of candidate lists + all-reduce of partial scores: WriteWorkerActor.scala:164-183 turned into
range shards) is `value`; candidate-range shards (no data-path collective) are the `candidate_range_layout` row.

Prints ONE JSON line (rank 0).  A rank that stalls ends the process with exit code 3 (watchdog), never 0.
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
MFMA_BF16_PEAK_TFLOPS = 2500.0  # ... dense bf16 MFMA peak (no sparsity)
MFMA_I8_PEAK_TOPS = 5000.0      # ... v_mfma_i32_32x32x32_i8: the cycles of the bf16 form at twice the K, i.e. 2x bf16 per clock
BYTES_PER_VISIT = 8            # one posting = int32 slot + fp32 weight (the ACCOUNTING unit)
METRIC = "scored candidate pairs/sec + achieved HBM GB/s, N=1M d=100k nnz=100, 1/2/4/8 GPU"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", help="c2 | c3 | c3z | c3z1 | c5s | c5 | c5z (c3 is the metric's)")
    ap.add_argument("--rows", "--n", dest="n", type=int, default=None, help="override the number of vectors (debug); "
                    "spell it --rows under torch.distributed.run, whose own parser claims every prefix of its options")
    ap.add_argument("--tile-rows", type=int, default=0)
    ap.add_argument("--head-terms", type=int, default=0, help="dense-head block: 0 = the library's policy, -1 never, N <= 32768 = that many of the most frequent terms (beyond 256: folded block)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-row", action="store_true", help="skip the fp32-accumulate sibling measurement")
    ap.add_argument("--no-two-directional-row", action="store_true", help="skip the APSS_FLAG_NO_SYMMETRY sibling measurement "
                                                                            "(profiling runs: its launches carry the headline kernel's name)")
    ap.add_argument("--term-shards", type=int, default=None, help="T of the T x D rank grid of the headline layout (default: all ranks)")
    ap.add_argument("--solo", default=None, help="T,D,i,j: time shard (term i of T, rows j of D) alone on this GPU (projection)")
    ap.add_argument("--no-comparison-row", action="store_true", help="skip the candidate-range layout of multi-GPU runs")
    ap.add_argument("--no-symmetric-ranges", action="store_true", help="D > 1: every cell meets every row as a query (round 3's scheme) "
                                                                         "instead of half of the other ranges with mirrored pairs")
    ap.add_argument("--deadline", type=float, default=900.0, help="seconds before the watchdog ends a stalled run (exit 3)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the real run) | gloo (rehearsal: all ranks share GPU 0)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU work budget of the baseline sample")
    ap.add_argument("--engine", default="auto", help="N > 1 without a launcher: auto | dist = start torch.distributed.run as a child "
                    "(one process per GPU, apss/dist.py) and relay its line; group = ONE process, the term-sharded index of the node "
                    "behind the C ABI (apss_group: member threads + RCCL below the boundary)")
    ap.add_argument("--share-device", action="store_true", help="--engine group rehearsal on a one-GPU box: every member on GPU 0 "
                                                                "(exchange by device-to-device copies instead of RCCL)")
    return ap.parse_args()


class Watchdog:
    """A stalled collective or kernel must not look like success: past the deadline the process prints what it was
    doing and leaves with exit code 3 (os._exit: a rank stuck inside a collective cannot unwind)."""

    def __init__(self, seconds, rank):
        self.phase, self.rank = "start", rank
        self.t = threading.Timer(seconds, self.fire)
        self.t.daemon = True
        self.t.start()

    def fire(self):
        sys.stderr.write("[bench] rank %d stalled in phase '%s': exit 3\n" % (self.rank, self.phase))
        if self.rank == 0:
            print(json.dumps({"metric": METRIC, "value": None, "error": "stalled in phase '%s'" % self.phase}), flush=True)
        sys.stderr.flush()
        os._exit(3)

    def done(self):
        self.t.cancel()


def cpu_baseline(cfg, rp, idx, val, budget_s):
    """The oracle's restatement of the reference algorithm ("port": posting lists + the per-candidate hash-map dot
    of CommonUtils.scala:98-117 in double), timed on this box's host cores on a bounded query sample."""
    from oracle import oracle
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(avail, 16)  # one GPU's share of the host (the box exposes every core of the machine): the headline row
    n = len(rp) - 1
    pilot_q = min(n, cores * 2)
    p = oracle.selfjoin_sample(0, cfg["dim"], cfg["theta"], rp, idx, val, 0, pilot_q, cores)
    per_q = max(p["seconds"] / max(pilot_q, 1), 1e-7)
    sample = int(min(n, max(pilot_q, budget_s / per_q)))
    r = oracle.selfjoin_sample(0, cfg["dim"], cfg["theta"], rp, idx, val, 0, sample, cores)
    # the reference's own parallelism (refcpu-T): T workers owning dim % T, every worker sees every vector
    # and re-scores every pair it reaches; the DISTINCT pairs of the sample (counted by the run above) per its wall time
    rt = oracle.selfjoin_sample(2, cfg["dim"], cfg["theta"], rp, idx, val, 0, max(pilot_q, sample // 4), cores)
    rt_distinct = oracle.selfjoin_sample(0, cfg["dim"], cfg["theta"], rp, idx, val, 0, max(pilot_q, sample // 4), cores)["cand_pairs"]
    opt = oracle.selfjoin_sample(1, cfg["dim"], cfg["theta"], rp, idx, val, 0, min(n, sample * 20), cores)
    all_cores = None
    if avail > cores:  # "T = all host cores, core count printed": the same port on every core the process may use
        s_all = sample  # (the same sample: a GPU box may expose every core of the machine while the run is entitled to a share of them)
        ra = oracle.selfjoin_sample(0, cfg["dim"], cfg["theta"], rp, idx, val, 0, s_all, avail)
        all_cores = {"value": ra["cand_pairs"] / ra["seconds"], "cores": avail,
                     "sample": "first %d queries, %.1f s, %d threads" % (s_all, ra["seconds"], avail),
                     "note": "memory-bound: every thread walks the same posting lists and a per-thread candidate stamp array of N "
                             "entries; beyond a socket's worth of threads the port does not scale (it may be SLOWER than the headline "
                             "row's %d threads) -- a stated baseline, not a tuned one" % cores}
    return {
        "value": r["cand_pairs"] / r["seconds"], "unit": "scored candidate pairs/s", "cores": cores, "kind": "port",
        "sample": "all %d vectors indexed, first %d queries timed (%.1f s, %d threads: queries split by range, every "
                  "pair scored once -- more generous than the reference's term-modulo workers, see term_modulo_workers); "
                  "reference algorithm: posting lists + per-candidate HashMap dot in double" % (n, sample, r["seconds"], cores),
        "term_modulo_workers": {
            "value": rt_distinct / rt["seconds"], "scorings_per_distinct_pair": rt["cand_pairs"] / max(1, rt_distinct),
            "note": "%d workers, worker t owns dims with dim %% %d == t and re-scores every pair it reaches "
                    "(EntryProxyActor.scala:41-46, IWA:92); distinct pairs of the first %d queries / wall time"
                    % (cores, cores, max(pilot_q, sample // 4))},
        "all_host_cores": all_cores if all_cores else {"value": r["cand_pairs"] / r["seconds"], "cores": cores,
                                                        "sample": "the process may use %d cores: same as the headline row" % avail},
        "optimised_cpu_value": opt["cand_pairs"] / opt["seconds"],
        "optimised_cpu_note": "fairness bracket: CSC + dense double accumulator, %d threads" % cores,
    }


PROBE_TRAFFIC = os.path.join(ROOT, "profiles", "r04_probe_traffic.json")


def csrc_sha256():
    """content hash of the library's sources (what profiles/collect_r04.sh records beside the counters it collects)"""
    import glob
    import hashlib
    csrc = os.path.join(ROOT, "all-pairs-similarity_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip"))) + sorted(glob.glob(os.path.join(csrc, "*.hpp"))) + [os.path.join(ROOT, "include", "apss.h")]:
        h.update(open(f, "rb").read())
    return h.hexdigest()


def profile_quote(workload, n_override, tile_rows, alg_bytes, kernel):
    """HBM-side bytes and SQ counters of the dominant kernel: rocprofv3 PMC passes cannot run inside this process, so the
    committed summary of the same command (profiles/collect_r04.sh -> profiles/r04_probe_traffic.json) is quoted -- only
    when the workload matches byte for byte AND the kernel that ran here (apss_stats.probe_kernel) is the kernel the
    counters were collected on; anything else is refused (traffic: null), never quoted from another kernel."""
    if n_override or tile_rows or not os.path.exists(PROBE_TRAFFIC):
        return None
    tj = json.load(open(PROBE_TRAFFIC))
    ent = tj.get(workload)
    if not ent or ent.get("algorithmic_bytes_per_launch") != alg_bytes:
        return None
    counted_on = ent.get("detail", {}).get("kernels", {}).get("filter", {}).get("kernel", "")
    if not kernel or ("apss::" + kernel) not in counted_on:
        return None
    if ent.get("csrc_sha256") != csrc_sha256():  # the kernels have changed since the counters were collected
        return None
    ent = dict(ent)
    ent["counted_on"] = {"kernel": counted_on, "csrc_sha256": ent.get("csrc_sha256")}
    return ent


def single_gpu(a, cfg, rp, idx, val, d_arrays, dev, local_rank, sync, wd):
    import torch
    from apss import _lib
    from apss.engine import ApssIndex
    n = cfg["n"]
    d_rp, d_idx, d_val = d_arrays
    d_ids = torch.arange(n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()  # the library runs on its own stream

    def run(flags, steps, warmup):
        ix = ApssIndex(cfg["dim"], cfg["theta"], device=local_rank, tile_rows=a.tile_rows, capacity_rows=n,
                       capacity_nnz=d_idx.numel(), flags=flags, head_terms=a.head_terms)

        def step():
            ix.clear()
            return ix.insert_and_query_dev(d_ids, d_rp, d_idx, d_val)

        for _ in range(warmup):
            step()
        sync()
        per = []
        t0 = time.perf_counter()
        for _ in range(steps):
            n_pairs = step()
            per.append(ix.stats())
        sync()
        dt = time.perf_counter() - t0
        ix.close()
        return dt, n_pairs, per

    wd.phase = "timed steps"
    dt, n_pairs, per = run(0, a.steps, a.warmup)
    st = per[-1]
    mean = lambda k: float(np.mean([s[k] for s in per]))  # noqa: E731
    sec_per_step = dt / a.steps
    visits, cands, launches = st["posting_visits"], st["candidate_pairs"], max(1, st["probe_launches"])
    probe_s, head_s = mean("probe_ms") * 1e-3, mean("head_ms") * 1e-3
    # the roofline is priced on the visits the KERNELS made: a whole-store join is symmetric (the library runs the filter over
    # the tile pairs on or below the diagonal and mirrors the survivors; `visits` / `cands` keep counting what the reference's
    # two-directional probe visits and scores -- see "symmetric_join" in the line)
    dev_visits = st["device_posting_visits"]
    alg_bytes = BYTES_PER_VISIT * dev_visits
    hybrid = st["head_terms"] > 0
    filter_kernel = st.get("probe_kernel", "")
    if filter_kernel.startswith("k_probe_even"):
        filter_kernel += " (F staging waves, chunks dealt evenly over the adding waves; LDS accumulators, 4-B postings) + k_rescore"
    elif filter_kernel.startswith("k_probe_coarse"):
        filter_kernel += " (16-bit LDS accumulators, 4-B postings) + k_rescore"
    # `value` counts what the DEVICE decided: on a symmetric whole-store join the filter meets every unordered tile pair once
    # (device_share = device visits / reference visits, 0.517 at C3) and both directions of a survivor are re-scored exactly;
    # the reference-equivalent count (both directions of every pair, as IndexingWorkerActor scores them) over the same wall
    # time is `value_reference_equivalent`, and the same join with both directions probed is `value_two_directional`
    device_share = dev_visits / visits if visits else 1.0
    out = {
        "value": cands * device_share / sec_per_step,
        "value_reference_equivalent": cands / sec_per_step,
        "device_share_of_candidate_pairs": device_share,
        "ms_per_step": sec_per_step * 1e3,
        "posting_visits_per_step": visits,
        "device_posting_visits_per_step": dev_visits,
        "candidate_pairs_per_step": cands,
        "result_pairs_per_step": int(n_pairs),
        "posting_visits_per_s": dev_visits / sec_per_step,
        "algorithmic_GBps_whole_step": alg_bytes / sec_per_step / 1e9,
        "build_ms": mean("build_ms"),
        "probe_kernel_ms": mean("probe_ms"),
        "rescore_ms": mean("rescore_ms"),
        "filter_survivors": st["filter_survivors"],
    }
    sparse_frac = (alg_bytes / launches / (probe_s / launches) / 1e9) / HBM_PEAK_GBS if probe_s > 0 else None
    quote = profile_quote(a.workload, a.n, a.tile_rows, alg_bytes, st.get("probe_kernel", ""))
    sparse_roof = {
        # what the counters show (profiles/r0N_probe_traffic.json): the filter kernel's busiest unit is the LDS pipeline
        # (returning 16-bit atomics + clears), not HBM; `frac` stays the SURVEY 8(d) accounting figure (8 B per visit)
        "bound": "lds", "kernel": filter_kernel,
        "achieved": alg_bytes / probe_s / 1e9 if probe_s > 0 else None,
        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sparse_frac,
        "frac_note": "ALGORITHMIC: 8 B x posting visits per launch / HIP-event kernel time / 8 TB/s (SURVEY.md 8d accounting "
                     "unit; the filter reads 4-B postings, so this figure may exceed what the memory system moved)",
        "launches": launches,
        "traffic": None, "measured_frac": None, "measured_frac_of_copy_peak": None,
    }
    if quote:
        traffic = quote["traffic_bytes_per_launch_corrected"]
        sparse_roof.update({
            "traffic": traffic,
            "measured_frac": traffic / (probe_s / launches) / 1e9 / HBM_PEAK_GBS,
            "measured_frac_of_copy_peak": traffic / (probe_s / launches) / 1e9 / HBM_COPY_GBS,
            "traffic_note": "bytes per launch at the L2 <-> fabric boundary, (2 x FETCH_SIZE + WRITE_SIZE) x 1024 from separate "
                            "rocprofv3 --pmc passes of this command (profiles/r04_probe_traffic.json); gfx950 tallies 128-B read "
                            "requests at 64 B; Infinity-Cache hits are included, HBM itself sees each posting once per step",
            "lds_issue_frac": quote.get("lds_issue_frac"), "valu_busy_frac": quote.get("valu_busy_frac"),
            "counted_on": quote["counted_on"],
        })
    if hybrid:
        flops = st["head_flops"]
        out.update({"head_terms": st["head_terms"], "head_int8": int(st.get("head_int8", 0)), "head_pairs_per_step": st["head_pairs"], "head_kernel_ms": head_s * 1e3,
                    "head_survivors": st["head_survivors"],
                    "candidate_pairs_note": "max(pairs sharing a tail term, pairs sharing a head term): a lower bound of "
                                            "the distinct pairs scored (a pair sharing both kinds is scored by both filters)"})
        i8 = bool(st.get("head_int8"))
        peak = MFMA_I8_PEAK_TOPS if i8 else MFMA_BF16_PEAK_TFLOPS
        head_roof = {"bound": "mfma", "kernel": "k_head_gemm<%d>%s (%s)" % (
                         min(256, st["head_columns"]), " (one block: 128 terms with a column each + %d terms folded into 128 columns)"
                         % (st["head_terms"] - 128) if st["head_terms"] > 256 else "",
                         "v_mfma_i32_32x32x32_i8: INT8 rows rounded up, exact integer filter" if i8 else "v_mfma_f32_32x32x16_bf16"),
                     "achieved": flops / head_s / 1e12 if head_s > 0 else None, "peak": peak,
                     "unit": "TOP/s" if i8 else "TFLOP/s", "frac": flops / head_s / 1e12 / peak if head_s > 0 else None,
                     "traffic": None,
                     "flops_note": "2 x KH x (512-slot query blocks x 64-row candidate tiles the grid multiplies; the upper "
                                   "triangle of a stored batch is skipped) per launch / HIP-event kernel time"}
        if head_s >= probe_s:
            out["roofline"], out["roofline_sparse_filter"] = head_roof, sparse_roof
        else:
            out["roofline"], out["roofline_dense_head"] = sparse_roof, head_roof
    else:
        out["roofline"] = sparse_roof
    if st["symmetric"] and not a.no_exact_row and not a.no_two_directional_row:
        # the same join with every pair probed in BOTH directions, as the reference does it (APSS_FLAG_NO_SYMMETRY): same result
        # list, measured beside the headline so that the line carries both
        wd.phase = "two-directional sibling"
        dt3, n3, per3 = run(_lib.FLAG_NO_SYMMETRY, max(1, min(2, a.steps)), 1)
        s3 = per3[-1]
        ps3 = float(np.mean([s["probe_ms"] for s in per3])) * 1e-3
        assert int(n3) == int(n_pairs) and s3["candidate_pairs"] == cands and s3["posting_visits"] == visits, "the symmetric join disagrees with the two-directional probe"
        out["symmetric_join"] = {
            "note": "the batch is the whole store, so (q, c) and (c, q) share terms and score: the filter kernels run the (query "
                    "tile, candidate tile) pairs on or below the diagonal, the survivors of the pairs below it are mirrored, and "
                    "BOTH directions are re-scored exactly (same result list, same statistics).  `value` counts the candidate pairs "
                    "the DEVICE decided (candidate_pairs_per_step x device share); value_reference_equivalent counts both directions "
                    "of every pair as the reference scores them",
            "device_share_of_posting_visits": dev_visits / visits if visits else None,
            "two_directional": {"value": cands / (dt3 / len(per3)), "ms_per_step": dt3 / len(per3) * 1e3, "probe_kernel_ms": ps3 * 1e3,
                                "frac": BYTES_PER_VISIT * visits / ps3 / 1e9 / HBM_PEAK_GBS if ps3 > 0 else None,
                                "flag": "APSS_FLAG_NO_SYMMETRY"},
        }
    # the fp32-everywhere sibling: single-pass kernel with exact u32 / fp32 accumulators, 8-B postings (what `frac`
    # means literally); measured in the same run so that the line carries both
    if not a.no_exact_row and not hybrid:
        wd.phase = "exact-accumulate sibling"
        dt2, n2, per2 = run(_lib.FLAG_EXACT_ACCUM, max(1, min(2, a.steps)), 1)
        s2 = per2[-1]
        ps = float(np.mean([s["probe_ms"] for s in per2])) * 1e-3
        l2 = max(1, s2["probe_launches"])
        ex = {"kernel": "k_probe_wave (8-B postings, exact u32 fixed-point accumulators, single pass)",
              "probe_kernel_ms": ps * 1e3, "ms_per_step": dt2 / len(per2) * 1e3, "result_pairs": int(n2),
              "frac": BYTES_PER_VISIT * s2["posting_visits"] / ps / 1e9 / HBM_PEAK_GBS if ps > 0 else None,
              "measured_frac": None, "launches": l2}
        if quote and quote.get("exact_traffic_bytes_per_launch_corrected"):
            ex["traffic"] = quote["exact_traffic_bytes_per_launch_corrected"]
            ex["measured_frac"] = ex["traffic"] / (ps / l2) / 1e9 / HBM_PEAK_GBS
        out["roofline"]["exact_accum"] = ex
        assert int(n2) == int(n_pairs), "the two probe paths disagree on the result set size"
    return out, "1 GPU", {}


def multi_gpu(a, cfg, rp, idx, val, dev, comm_dev, rank, world, local_rank, sync, wd):
    import torch
    import torch.distributed as dist
    from apss.dist import ShardedJoin
    from apss.engine import ApssIndex
    n = cfg["n"]
    # distinct scored candidate pairs of the workload (a pair sharing terms in k shards is touched by k shards):
    # counted once, untimed, by a plain single-GPU join on rank 0
    wd.phase = "distinct pair count (rank 0, untimed)"
    distinct = torch.zeros(1, dtype=torch.float64, device=comm_dev)
    if rank == 0:
        ix0 = ApssIndex(cfg["dim"], cfg["theta"], device=local_rank, tile_rows=a.tile_rows, head_terms=a.head_terms)
        ix0.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        ix0.insert_and_query_dev(torch.arange(n, dtype=torch.int64, device=dev), torch.from_numpy(rp).to(dev),
                                 torch.from_numpy(idx).to(dev), torch.from_numpy(val.astype(np.float32)).to(dev))
        distinct[0] = float(ix0.stats()["candidate_pairs"])
        ix0.close()
        del ix0
        torch.cuda.empty_cache()
    dist.all_reduce(distinct)
    cands = int(distinct.item())

    def timed(T, label):
        wd.phase = label
        sj = ShardedJoin(cfg["dim"], cfg["theta"], rank, world, dev, tile_rows=a.tile_rows, comm_device=comm_dev, term_shards=T,
                         head_terms=a.head_terms, symmetric_ranges=not a.no_symmetric_ranges)
        sj.load(rp, idx, val)
        probe_ms, build_ms, head_ms = [], [], []
        for _ in range(a.warmup):
            sj.step()
        sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            n_pairs = sj.step()
            probe_ms.append(sj.last["probe_ms"])
            build_ms.append(sj.last["build_ms"])
            head_ms.append(sj.last.get("head_ms", 0.0))
        sync()
        hm_mine = float(np.mean(head_ms))
        # (the dense-head kernel of the slowest rank, with THAT rank's flops: the rate is a per-GPU figure)
        t = torch.tensor([time.perf_counter() - t0, float(np.mean(probe_ms)), float(np.mean(build_ms)), hm_mine], dtype=torch.float64,
                         device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # slowest rank
        dt, pm, bm, hm = (float(x) for x in t.tolist())
        hf = torch.tensor([sj.last.get("head_flops", 0.0) if hm_mine >= hm else 0.0], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(hf, op=dist.ReduceOp.MAX)
        head_flops = float(hf.item())
        sec = dt / a.steps
        row = {
            "grid": "%d term-range shards x %d candidate ranges" % (sj.T, sj.D),
            "value": cands / sec, "ms_per_step": sec * 1e3, "result_pairs_per_step": int(n_pairs),
            "probe_kernel_ms_slowest_shard": pm, "build_ms_slowest_shard": bm,
            "posting_visits_per_step": sj.last["posting_visits"], "shard_touched_pairs_sum": sj.last["candidate_pairs"],
            "exchange": sj.last.get("exchange"),
            "collectives": ("per step: all-gather of candidate lists + all-reduce(SUM) of per-candidate partial scores inside "
                            "each term group, all-reduce of counters over all ranks" if sj.T > 1 else
                            "per step: all-reduce of counters only (result sets of different candidate ranges are disjoint)"),
        }
        visits = sj.last.get("device_posting_visits", sj.last["posting_visits"])  # (what the kernels visited: a whole-store shard join is symmetric)
        row["device_posting_visits_per_step"] = visits
        # as at N = 1: `value` counts the candidate pairs the devices decided, the reference-equivalent count rides along
        share = visits / sj.last["posting_visits"] if sj.last["posting_visits"] else 1.0
        row["value_reference_equivalent"] = row["value"]
        row["device_share_of_candidate_pairs"] = share
        row["value"] = cands * share / sec
        sparse_roof = {"bound": "lds", "kernel": sj.last.get("probe_kernel") or "k_probe_coarse",
                       "achieved": BYTES_PER_VISIT * visits / world / (pm * 1e-3) / 1e9 if pm > 0 else None,
                       "peak": HBM_PEAK_GBS, "unit": "GB/s (per GPU: slowest shard's kernel, 1/N of the job's 8-B posting visits)",
                       "frac": BYTES_PER_VISIT * visits / world / (pm * 1e-3) / 1e9 / HBM_PEAK_GBS if pm > 0 else None,
                       "traffic": None}
        row["roofline"] = sparse_roof
        if sj.last.get("head_terms") and hm > 0:
            # skewed terms: the join's dense-head block, cut over the term group's ranks by candidate tile (apss/dist.py)
            nt = sj.last["head_terms"]
            kh = 64 if nt <= 64 else (128 if nt <= 128 else 256)
            i8 = bool(sj.last.get("head_int8"))
            peak = MFMA_I8_PEAK_TOPS if i8 else MFMA_BF16_PEAK_TFLOPS
            head_roof = {"bound": "mfma", "kernel": "k_head_gemm<%d>%s (%s), candidate tiles t %% %d == rank" % (
                             max(kh, 128) if i8 else kh, " (128 terms with a column each + %d folded into 128 columns)" % (nt - 128) if nt > 256 else "",
                             "v_mfma_i32_32x32x32_i8: INT8 rows rounded up" if i8 else "v_mfma_f32_32x32x16_bf16", sj.T),
                         "achieved": head_flops / (hm * 1e-3) / 1e12, "peak": peak,
                         "unit": "%s (per GPU: the slowest rank's kernel and that rank's operations)" % ("TOP/s" if i8 else "TFLOP/s"),
                         "frac": head_flops / (hm * 1e-3) / 1e12 / peak, "traffic": None}
            row.update({"head_terms": sj.last["head_terms"], "head_int8": int(i8), "head_kernel_ms_slowest_rank": hm})
            if hm >= pm:
                row["roofline"], row["roofline_sparse_filter"] = head_roof, sparse_roof
            else:
                row["roofline_dense_head"] = head_roof
        del sj.engine
        del sj
        torch.cuda.empty_cache()
        return row

    T = a.term_shards or world
    head = timed(T, "headline layout: %d term-range shards" % T)
    extra = {"layout": "term-range shards (the contract: BASELINE.json configs[3], SURVEY.md 8e)", "backend": a.backend,
             "exchange": head["exchange"], "shard_touched_pairs_sum": head["shard_touched_pairs_sum"],
             "collectives": head["collectives"]}
    if not a.no_comparison_row and T != 1:
        comp = timed(1, "comparison layout: candidate ranges")
        extra["candidate_range_layout"] = comp
        # (same set up to the pairs within an ulp of theta: a row-range cell reports a cross pair in both directions with ONE fp32
        # score, a term shard sums each direction in its own order -- |score - theta| <= 1e-5 may differ)
        same = lambda x, y: abs(x - y) <= max(4, int(2e-5 * max(x, y)))  # noqa: E731
        assert same(comp["result_pairs_per_step"], head["result_pairs_per_step"]), "the two layouts disagree on the result set size"
        if world >= 4 and world % 2 == 0 and T == world:
            # the grid between the two: 2 term ranges x N/2 candidate ranges -- still a term-sharded index with the RCCL exchange of
            # partial scores inside every pair, but rounds four times as fat as at T = N (projected fastest)
            mid = timed(2, "comparison layout: 2 term ranges x %d candidate ranges" % (world // 2))
            extra["grid_2_term_ranges_layout"] = mid
            assert same(mid["result_pairs_per_step"], head["result_pairs_per_step"]), "the grids disagree on the result set size"
    out = {
        "value": head["value"], "ms_per_step": head["ms_per_step"],
        "posting_visits_per_step": head["posting_visits_per_step"], "candidate_pairs_per_step": cands,
        "result_pairs_per_step": head["result_pairs_per_step"],
        "posting_visits_per_s": head["device_posting_visits_per_step"] / (head["ms_per_step"] * 1e-3),
        "device_posting_visits_per_step": head["device_posting_visits_per_step"],  # (all ranks; a whole-store shard join is symmetric)
        "algorithmic_GBps_whole_step": BYTES_PER_VISIT * head["device_posting_visits_per_step"] / (head["ms_per_step"] * 1e-3) / 1e9,
        "build_ms": head["build_ms_slowest_shard"], "probe_kernel_ms": head["probe_kernel_ms_slowest_shard"],
        "roofline": head["roofline"],
        "value_reference_equivalent": head["value_reference_equivalent"],
        "device_share_of_candidate_pairs": head["device_share_of_candidate_pairs"],
    }
    for k in ("roofline_dense_head", "roofline_sparse_filter", "head_terms", "head_int8", "head_kernel_ms_slowest_rank"):
        if k in head:
            out[k] = head[k]
    return out, head["grid"] + "; " + head["collectives"], extra


def group_gpu(a, cfg, rp, idx, val, wd):
    """N GPUs in ONE process: the term-sharded index of the node behind the C ABI (apss_group, include/apss.h) -- member i on
    GPU i, one host thread per member, candidate all-gather + partial-score all-reduce over RCCL below the boundary.  A step =
    apss_group_clear + apss_group_insert_and_query_dev on batches already resident in every member's HBM."""
    import torch
    from apss import _lib
    from apss.engine import ApssGroup, ApssIndex
    n, N = cfg["n"], a.gpus
    devices = [0] * N if a.share_device else list(range(N))
    wd.phase = "distinct pair count (device 0, untimed)"
    dev0 = torch.device("cuda", 0)
    d0 = (torch.arange(n, dtype=torch.int64, device=dev0), torch.from_numpy(rp).to(dev0), torch.from_numpy(idx).to(dev0),
          torch.from_numpy(val.astype(np.float32)).to(dev0))
    torch.cuda.synchronize()
    ix0 = ApssIndex(cfg["dim"], cfg["theta"], device=0, tile_rows=a.tile_rows, head_terms=a.head_terms)
    ix0.insert_and_query_dev(*d0)
    cands = ix0.stats()["candidate_pairs"]
    ix0.close()
    per_member, made = [], {0: d0}
    for d in devices:
        if d not in made:
            dv = torch.device("cuda", d)
            made[d] = tuple(t.to(dv) for t in d0)
        per_member.append(made[d])
    for d in set(devices):
        torch.cuda.synchronize(d)
    wd.phase = "group: layout from the first batch (untimed), warm-up"
    g = ApssGroup(cfg["dim"], cfg["theta"], devices, tile_rows=a.tile_rows, head_terms=a.head_terms)

    def step():
        g.clear()
        return g.insert_and_query_dev(per_member)

    for _ in range(max(1, a.warmup)):  # (the first call decides the layout: term cuts, dense-head block)
        step()
    wd.phase = "group: timed steps"
    per = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        n_pairs = step()
        per.append(g.stats())
    dt = time.perf_counter() - t0  # (the group call returns when every member has finished: nothing is in flight)
    st = per[-1]
    members = [g.member_stats(i) for i in range(N)]
    g.close()
    sec = dt / a.steps
    mean = lambda k: float(np.mean([s[k] for s in per]))  # noqa: E731
    visits, dev_visits = st["posting_visits"], st["device_posting_visits"]
    share = dev_visits / visits if visits else 1.0
    pm = mean("probe_ms_max")
    exch = {APSS: name for APSS, name in ((_lib.EXCHANGE_NONE, "none"), (_lib.EXCHANGE_COPIES, "device-to-device copies (members share a GPU)"),
                                          (_lib.EXCHANGE_RCCL, "RCCL: ncclBroadcast-grouped all-gather + ncclAllReduce(SUM) on the members' streams"))}
    body = {
        "value": cands * share / sec, "value_reference_equivalent": cands / sec, "device_share_of_candidate_pairs": share,
        "ms_per_step": sec * 1e3, "posting_visits_per_step": visits, "device_posting_visits_per_step": dev_visits,
        "candidate_pairs_per_step": cands, "result_pairs_per_step": int(n_pairs),
        "posting_visits_per_s": dev_visits / sec, "algorithmic_GBps_whole_step": BYTES_PER_VISIT * dev_visits / sec / 1e9,
        "build_ms": mean("build_ms_max"), "probe_kernel_ms": pm,
        "roofline": {"bound": "lds", "kernel": members[0]["probe_kernel"],
                     "achieved": BYTES_PER_VISIT * dev_visits / N / (pm * 1e-3) / 1e9 if pm > 0 else None, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s (per GPU: slowest member's kernel, 1/N of the job's 8-B posting visits)",
                     "frac": BYTES_PER_VISIT * dev_visits / N / (pm * 1e-3) / 1e9 / HBM_PEAK_GBS if pm > 0 else None, "traffic": None},
        "group": {"engine": "apss_group (csrc/apss_group.hip): one process, one host thread per member", "exchange": exch[st["exchange"]],
                  "member_ms_slowest": mean("member_ms_max"), "exchange_ms": mean("exchange_ms"), "partial_scores_ms_slowest": mean("partial_ms_max"),
                  "head_terms": st["head_terms"], "head_kernel_ms_slowest": mean("head_ms_max"),
                  "candidates_per_member_sum": st["candidates_sum"], "candidates_longest_list": st["candidates_max"], "union": st["union_pairs"],
                  "all_gather_bytes_per_member": st["all_gather_bytes"], "all_reduce_bytes": st["all_reduce_bytes"],
                  "term_cuts": st["term_cuts"], "member_touched_pairs_sum": st["member_touched_pairs"],
                  "devices": devices},
    }
    if a.share_device:
        body["group"]["rehearsal"] = "every member on GPU 0: the members' kernels share one device, the exchange runs by copies -- not a scaling number"
    par = "%d term-range members in one process (apss_group); per step: all-gather of candidate lists + all-reduce(SUM) of partial scores, %s" % (
        N, "RCCL" if st["exchange"] == _lib.EXCHANGE_RCCL else "copies")
    return body, par, {"layout": "term-range shards (the contract: BASELINE.json configs[3], SURVEY.md 8e) behind the C ABI", "backend": "rccl-in-library"}


def relay_child(a):
    """`python bench.py --gpus N` typed without a launcher: start torch.distributed.run as a CHILD process (before this
    process touches the GPU), pass the arguments through, relay its one JSON line and its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = [x for x in sys.argv[1:]]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    elif p.returncode == 0:
        sys.stderr.write("[bench] the launched ranks printed no result line\n")
        return 4
    return p.returncode


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from apss import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    group_engine = world == 1 and a.gpus > 1 and a.engine == "group"
    if world == 1 and a.gpus > 1 and not group_engine:
        # typed without a launcher: the ranks are started here, as a child process, before anything touches the GPU
        sys.exit(relay_child(a))
    wd = Watchdog(a.deadline, rank)
    if a.backend == "gloo":
        local_rank = 0  # rehearsal on a one-GPU box: every rank drives GPU 0, collectives on CPU tensors
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm_dev = torch.device("cpu") if a.backend == "gloo" else dev
    if world > 1:
        wd.phase = "process group"
        if a.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    wd.phase = "data generation"
    cfg = dict(synth.CONFIGS[a.workload])
    if a.n:
        cfg["n"] = a.n
    t0 = time.time()
    d_arrays = None
    if cfg.get("gen") == "zipf_dev":
        # drawn on the device (the host loop needs minutes per million Zipf rows); the host copy feeds the CPU baseline
        d_rp, d_idx2, d_val2 = synth.make_vectors_zipf_dev(cfg["n"], cfg["dim"], cfg["nnz"], cfg["zipf_s"], cfg["seed"], dev)
        d_arrays = (d_rp, d_idx2.reshape(-1).contiguous(), d_val2.reshape(-1).contiguous())
        rp, idx = d_rp.cpu().numpy(), d_arrays[1].cpu().numpy()
        val = d_arrays[2].double().cpu().numpy()
    else:
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
        wd.phase = "solo shard"
        T, D, ti, dj = (int(x) for x in a.solo.split(","))
        df = np.bincount(idx, minlength=cfg["dim"])
        head = None
        if T > 1 and a.head_terms >= 0:  # what rank 0 of the sharded join decides and broadcasts (apss/dist.py)
            from apss.dist import hip_head_chooser
            from apss.dist import limit_head_depth
            terms, fold = hip_head_chooser(cfg["dim"], cfg["theta"], dev, a.head_terms)(rp, idx, val)
            if a.head_terms == 0:
                terms = limit_head_depth(terms, df, n, T)
            if terms.size:
                head = (terms, ti, T, fold)
                df[terms] = 0
        tr = term_ranges(df, T)[ti]
        eng = HipShardEngine(cfg["dim"], cfg["theta"], tr, dev, a.tile_rows, head)
        spans = None
        if D > 1 and not a.no_symmetric_ranges:  # the rows this cell owns the cross pairs with (apss.dist.ShardedJoin.half_spans)
            from apss.dist import half_spans
            spans = half_spans(n, D, dj)
        eng.load(rp, idx, val, None if D == 1 else (n * dj // D, n * (dj + 1) // D), query_spans=spans)

        def step():  # one step of the cell: every phase's handle call (term shards hand their candidates on)
            return sum(int(eng.run_phase(ph)) for ph in eng.phases())
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            n_c = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print(json.dumps({"solo_shard": a.solo, "ms_per_step": dt * 1e3, "probe_kernel_ms": eng.stats["probe_ms"],
                          "build_ms": eng.stats["build_ms"], "posting_visits": eng.stats["posting_visits"], "device_posting_visits": eng.stats["device_posting_visits"],
                          "candidates": int(n_c), "probe_kernel": eng.stats["probe_kernel"], "head_terms": eng.stats["head_terms"],
                          "head_ms": eng.stats["head_ms"], "head_survivors": eng.stats["head_survivors"],
                          "head_int8": eng.stats["head_int8"], "queries_per_round": eng.stats.get("queries_per_round"),
                          "filter_survivors": eng.stats["filter_survivors"],
                          "head_frac_of_mfma_peak": (eng.stats["head_flops"] / (eng.stats["head_ms"] * 1e-3) / 1e12 / (MFMA_I8_PEAK_TOPS if eng.stats["head_int8"] else MFMA_BF16_PEAK_TFLOPS)
                                                     if eng.stats["head_ms"] > 0 else None),
                          "algorithmic_frac": BYTES_PER_VISIT * eng.stats["device_posting_visits"] / (eng.stats["probe_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}))
        wd.done()
        return

    if group_engine:
        d_arrays = None
        torch.cuda.empty_cache()
        body, parallelism, extra = group_gpu(a, cfg, rp, idx, val, wd)
    elif world == 1:
        if d_arrays is None:
            d_arrays = (torch.from_numpy(rp).to(dev), torch.from_numpy(idx).to(dev), torch.from_numpy(val.astype(np.float32)).to(dev))
        body, parallelism, extra = single_gpu(a, cfg, rp, idx, val, d_arrays, dev, local_rank, sync, wd)
    else:
        d_arrays = None
        torch.cuda.empty_cache()
        body, parallelism, extra = multi_gpu(a, cfg, rp, idx, val, dev, comm_dev, rank, world, local_rank, sync, wd)

    if rank == 0:
        out = {
            "metric": METRIC,
            "value": body.pop("value"),
            "unit": "scored candidate pairs/s",
            "n_gpus": a.gpus if group_engine else world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": body.pop("ms_per_step"),
            "higher_is_better": True,
            "scaling": "strong" if (world > 1 or group_engine) else "weak",
            "vs_baseline": None,
            # 16-bit fixed-point filter sums (+ the MFMA head filter: INT8 rows rounded up, or bf16), fp32 exact rescoring
            "dtype": "u16+f32" if not body.get("head_terms") else ("u16+i8+f32" if body.get("head_int8") else "u16+bf16+f32"),
            "data": "synthetic",
            "config": {"workload": "%s: N=%d dim=%d nnz=%d %s theta=%g, single-batch self-join (build + probe)" % (
                a.workload, n, cfg["dim"], cfg["nnz"], "Zipf(%g)" % cfg["zipf_s"] if cfg["zipf_s"] else "uniform",
                cfg["theta"]), "parallelism": parallelism,
                "tile_rows": a.tile_rows or "library default (coarse index: 32768 rows per tile, 65536 in the sparse regime)"},
            "datagen_s": gen_s,
        }
        out.update(body)
        out.update(extra)
        sym = out.get("symmetric_join")
        out["value_counts"] = ("candidate pairs the DEVICE decided per second of step wall time: a whole-batch join is symmetric, the "
                               "filter meets every unordered tile pair once (device share = device_posting_visits_per_step / "
                               "posting_visits_per_step) and both directions of a survivor are re-scored exactly; "
                               "value_reference_equivalent = both directions of every pair as the reference scores them, same wall "
                               "time; value_two_directional = the same join with APSS_FLAG_NO_SYMMETRY, timed beside it")
        if sym:
            out["value_two_directional"] = sym["two_directional"]["value"]
            out["ms_per_step_two_directional"] = sym["two_directional"]["ms_per_step"]
        if not a.no_cpu_baseline:
            wd.phase = "cpu baseline"
            out["cpu_baseline"] = cpu_baseline(cfg, rp, idx, val, a.cpu_seconds if (world == 1 and not group_engine) else min(a.cpu_seconds, 6.0))
        print(json.dumps(out), flush=True)
    if world > 1:
        wd.phase = "shutdown"
        dist.barrier()
        dist.destroy_process_group()
    wd.done()


if __name__ == "__main__":
    main()
