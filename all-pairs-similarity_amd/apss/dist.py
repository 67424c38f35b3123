"""Term-range-sharded self-join across the GPUs of one node (one process per GPU, torch.distributed = RCCL over xGMI).

The reference shards its inverted index by term too (dim % maxShardNum, dim % maxIndexEntryActorNum:
WriteWorkerActor.scala:172-175, EntryProxyActor.scala:41-46) but replicates every full vector to every shard and
lets each worker recompute the whole dot product (SparseVectorWrapper.scala:9).  Here shard g owns a contiguous
term RANGE, stores only that slice of every vector, and the exact score is assembled from per-shard partials:

  1. local (no communication): shard g probes its slice and keeps the CANDIDATES, pairs whose partial p_g satisfies
     p_g >= theta * |q_g||c_g| / (|q||c|)  (|x_g| = L2 norm of x restricted to g's terms, |x| = norm of the whole row;
     the library stores the ratios |x_g| / |x| at ingest).  If sum_g p_g >= theta then, since p_g <= |q_g||c_g| and
     sum_g |q_g||c_g| <= |q||c| (Cauchy-Schwarz twice), at least one shard passes the test -- whatever the row norms
     and the signs of the weights: no true pair is lost, and random pairs (one shared term) almost never pass.
  2. all-gather of the candidate lists (a few MB at most), union;
  3. every shard computes its exact partial for every candidate (apss_partial_scores_dev);
  4. ONE all-reduce(SUM) of the per-candidate partial scores over RCCL, then the `>= theta` prune (IWA:93).

A dense all-reduce of per-query accumulators would move 4*N bytes per query (~50x slower than
the single-GPU join); this exchange moves O(#near-pairs).  No precondition on the norms (the reference assumes
normalised vectors, CommonUtils.scala:88; un-normalised input gives the same pairs as the single-GPU join).

Skewed term distributions (TF-IDF is Zipfian; BASELINE.json configs[4]): the most frequent terms H are taken out of
EVERY shard's term range and scored as a dense block on the matrix cores (csrc/apss_head.hpp).  The candidate rule's
proof holds for any partition of the terms, so {H, T_1 .. T_T} is one: a pair with total >= theta passes
p_g >= theta |q_g||c_g| / (|q||c|) for some g in {H, T_1, .., T_T}.  The tail tests run on the shards as before (the
ratios exclude H); the head test  w_q . w_c >= theta  (w_x = x_H |x| / |x_H|, bf16 MFMA contraction) is cut over the T
ranks of a group by CANDIDATE ROW -- rank i multiplies the 64-row candidate tiles t with t % T == i against the whole
query batch -- and its survivors join the same candidate lists: all-gather, exact partial scores (a head term's entry
stays in the store of the shard whose range holds it), all-reduce, threshold.  Every rank must use the same H: rank 0
decides (the library's own policy on a sample of the rows) and broadcasts the terms once, at load time.

The compute engine is injected so the host logic can be exercised on CPU with gloo in tests (tests/ provides an
engine backed by the CPU oracle); the product engine below is HIP-only.
"""
import numpy as np
import torch
import torch.distributed as dist


def term_ranges(df, world):
    """Cut [0, dim) into `world` contiguous ranges with (nearly) equal sum of df^2 (= posting visits of a self-join)."""
    w = df.astype(np.float64) ** 2
    cum = np.concatenate([[0.0], np.cumsum(w)])
    total = cum[-1]
    cuts = [0]
    for g in range(1, world):
        cuts.append(int(np.searchsorted(cum, total * g / world)))
    cuts.append(len(df))
    for g in range(1, world + 1):  # strictly increasing, non-empty
        cuts[g] = max(cuts[g], cuts[g - 1] + 1)
    cuts[-1] = len(df)
    return [(cuts[g], cuts[g + 1]) for g in range(world)]


def hip_head_chooser(dim, theta, device, head_terms=0, sample_rows=131072):
    """The dense-head block of a sharded join, decided by the library's own policy (choose_head in csrc/apss_hip.hip:
    sampled document frequencies, measured selectivity) on the first `sample_rows` rows of the batch: a plain handle
    indexes the sample and is asked which terms it took.  head_terms: 0 = policy, N <= 32768 = that many of the most frequent terms."""
    def choose(rp, idx, val):
        from .engine import ApssIndex
        m = int(min(len(rp) - 1, sample_rows))
        e = int(rp[m])
        with ApssIndex(dim, theta, device=device.index or 0, head_terms=head_terms) as ix:
            ix.insert(np.arange(m, dtype=np.int64), rp[:m + 1], idx[:e], val[:e])
            return ix.head_terms(), 0  # the terms; 0 = the library's default split of a wide head's 256 columns
    return choose


def limit_head_depth(head, df, n_rows, T, min_tail_terms=8.0):
    """A deep head leaves few tail terms per row, and T term ranges cut them T ways: a row with a single term in a range
    pairs up at within-range cosine 1 with every row sharing it, and the candidate rule stops being selective (C3 with
    Zipf(1) terms, 8192-term head, T = 8: 2.6 tail terms per row and shard, 3.3e7 candidates per shard instead of 2e5 at
    T = 2).  Keep the head only as deep as leaves about `min_tail_terms` tail terms per row and shard; never shallower
    than the 256 terms that hold the long posting lists."""
    k = int(head.size)
    total = int(df.sum())
    while k > 256 and (total - int(df[head[:k]].sum())) / max(1, n_rows) / T < min_tail_terms:
        k //= 2
    return head[:max(k, 256)] if head.size > 256 else head


class HipShardEngine:
    """One shard resident on one MI355X (libapss_hip.so through the C ABI; no fallback): the postings of the terms in
    `term_range` for the candidate rows in `row_range`.  Queries are always all rows of the batch.
    head = (terms, part, n_parts): the join's dense-head block and this shard's share of its candidate tiles."""

    def __init__(self, dim, theta, term_range, device, tile_rows=0, head=None):
        from .engine import ApssIndex
        self.device = device
        self.ix = ApssIndex(dim, theta, device=device.index or 0, tile_rows=tile_rows, term_range=term_range)
        if head is not None and len(head[0]):
            self.ix.set_head_terms(head[0], head[1], head[2], head[3] if len(head) > 3 else 0)
        # run on torch's stream: the tensors handed to the library are produced by torch kernels on that stream
        self.ix.set_stream(torch.cuda.current_stream(device).cuda_stream)

    def load(self, rp, idx, val, row_range=None, device_arrays=None):
        """the batch as host CSR (numpy), or already resident: device_arrays = (rowptr int64, idx int32, val fp32) on
        this engine's GPU (several shard engines of one process then share one copy)"""
        dev = self.device
        if device_arrays is not None:
            self.d_rp, self.d_idx, self.d_val = device_arrays
        else:
            self.d_rp = torch.from_numpy(rp).to(dev)
            self.d_idx = torch.from_numpy(idx).to(dev)
            self.d_val = torch.from_numpy(val.astype(np.float32)).to(dev)
        self.n = self.d_rp.numel() - 1
        self.r0, self.r1 = (0, self.n) if row_range is None else row_range
        self.d_ids = torch.arange(self.n, dtype=torch.int64, device=dev)
        self.whole = (self.r0, self.r1) == (0, self.n)
        if not self.whole:  # the candidate rows of this shard as their own CSR batch
            e0, e1 = int(self.d_rp[self.r0].item()), int(self.d_rp[self.r1].item())
            self.s_rp = (self.d_rp[self.r0:self.r1 + 1] - e0).contiguous()
            self.s_idx = self.d_idx[e0:e1].contiguous()
            self.s_val = self.d_val[e0:e1].contiguous()
            self.s_ids = self.d_ids[self.r0:self.r1].contiguous()
        torch.cuda.synchronize()

    def join(self):
        """rebuild the shard's index and run its join; returns the number of pairs it reports (they stay on the device)"""
        self.ix.clear()
        if self.whole:
            n = self.ix.insert_and_query_dev(self.d_ids, self.d_rp, self.d_idx, self.d_val)
        else:
            self.ix.insert_dev(self.s_ids, self.s_rp, self.s_idx, self.s_val)
            n = self.ix.query_dev(self.d_ids, self.d_rp, self.d_idx, self.d_val)
        self.stats = self.ix.stats()
        # query row == batch row and candidate slot == row - r0 only while ingest keeps every row (no admission filter here)
        assert self.stats["rows"] == self.r1 - self.r0, "shard engine: the handle dropped rows"
        return n

    def candidates(self):
        """phase 1: rebuild the shard's index and return its candidate pairs (query row, candidate row), global rows"""
        n = self.join()
        # the lists stay in HBM: (query row, candidate slot) copied device-to-device into torch tensors on the same stream
        q = torch.empty(n, dtype=torch.int32, device=self.device)
        c = torch.empty(n, dtype=torch.int32, device=self.device)
        sc = torch.empty(n, dtype=torch.float32, device=self.device)
        if n:
            self.ix.results_to(q, c, sc)
        self.scores = sc  # final scores when the handle holds the whole term space (T == 1)
        return q.to(torch.int64), c.to(torch.int64) + self.r0  # query row == global row; slot -> global candidate row

    def partial(self, q_row, c_row):
        out = torch.empty(q_row.numel(), dtype=torch.float32, device=self.device)
        if q_row.numel():
            qi = q_row.to(torch.int32).contiguous()
            ci = (c_row - self.r0).to(torch.int32).contiguous()  # candidate slot inside this shard
            self.ix.partial_scores_dev(qi, ci, out)  # same stream as the conversions above (set_stream in __init__)
        return out


def join_shards_local(engines, n, theta):
    """The same four phases with every shard in ONE process (no collectives): used to test the shard kernels and the
    candidate rule on a single GPU.  Returns (q_rows, c_slots, scores) of the pairs >= theta."""
    keys = []
    for e in engines:
        q, c = e.candidates()
        keys.append(q.to(torch.int64) * n + c.to(torch.int64))
    uniq = torch.unique(torch.cat(keys))
    uq = torch.div(uniq, n, rounding_mode="floor")
    uc = uniq - uq * n
    total = None
    for e in engines:
        p = e.partial(uq, uc)
        total = p if total is None else total + p
    keep = total >= theta
    return uq[keep].cpu().numpy(), uc[keep].cpu().numpy(), total[keep].cpu().numpy(), [int(k.numel()) for k in keys]


class ShardedJoin:
    """world = T x D ranks: rank r owns term range (r % T) of the candidate rows in row range (r // T).

    T term shards share a candidate range and combine their partial scores with an all-reduce inside their group
    (the exchange the term-sharded index needs); the D candidate ranges are independent (their result sets are
    disjoint).  A term shard has 1/T of the posting visits of every (query, tile) round but the same number of rounds,
    so its speed-up trails T; candidate ranges need no data-path collective.  A shard that holds ALL rows (D = 1) joins its
    batch symmetrically (half the rounds), a row-range cell meets its queries as an outside batch and
    does not: measured one shard at a time on uniform C3, T = 8 is 3.8x one GPU before the exchange, eight candidate ranges
    3.3x (section 7).  `term_shards` = T picks the layout:
    bench.py's headline is T = world (the layout BASELINE.json names), this class's default T = 1.
    head_terms: dense-head block of the term-sharded layouts (module docstring): 0 = the library's policy decides on rank 0,
    -1 = never, N <= 32768 = that many of the most frequent terms.  (With T = 1 every handle is a plain one and decides
    for itself.)"""

    def __init__(self, dim, theta, rank, world, device, tile_rows=0, engine_factory=None, comm_device=None,
                 term_shards=None, head_terms=0, head_chooser=None):
        self.dim, self.theta, self.rank, self.world, self.device = dim, float(theta), rank, world, device
        # collectives run on `comm_device` tensors: the GPU itself under RCCL, the CPU when rehearsing with gloo
        self.comm = comm_device or device
        self.tile_rows = tile_rows
        T = term_shards or 1
        if world % T:
            raise ValueError("term_shards must divide the world size")
        self.T, self.D = T, world // T
        self.ti, self.dj = rank % T, rank // T
        self.group = None
        if world > 1 and T > 1 and self.D > 1:  # every rank creates every group, in the same order
            for j in range(self.D):
                g = dist.new_group([j * T + i for i in range(T)])
                if j == self.dj:
                    self.group = g
        self.engine_factory = engine_factory or (lambda tr, head=None: HipShardEngine(dim, theta, tr, device, tile_rows, head))
        self.head_mode = head_terms
        self.head_chooser = head_chooser or hip_head_chooser(dim, theta, device, max(0, head_terms))
        self.head = np.zeros(0, np.int32)
        self.head_fold = 0
        self.last = {}

    def _decide_head(self, rp, idx, val):
        """rank 0 asks the policy, every rank gets the same terms (one broadcast at load time, not in the data path)"""
        buf = torch.zeros(32770, dtype=torch.int32, device=self.comm)
        if self.rank == 0:
            chosen = self.head_chooser(rp, idx, val)
            t, fold = chosen if isinstance(chosen, tuple) else (chosen, 0)  # (terms, columns of the folded block | 0)
            t = np.asarray(t, dtype=np.int32)[:32768]
            buf[0], buf[1] = int(t.size), int(fold)
            if t.size:
                buf[2:2 + t.size] = torch.from_numpy(t).to(self.comm)
        if self.world > 1:
            dist.broadcast(buf, src=0)
        h = buf.cpu().numpy()
        self.head_fold = int(h[1])
        return h[2:2 + int(h[0])].astype(np.int32)

    def load(self, rp, idx, val):
        """every rank holds the same batch (same seed / same broadcast); each indexes only its terms x rows"""
        df = np.bincount(idx, minlength=self.dim)
        self.tail_terms_per_row_and_shard = float(idx.size) / max(1, len(rp) - 1) / self.T
        if self.T > 1 and self.head_mode >= 0:
            self.head = self._decide_head(rp, idx, val)
            if self.head_mode == 0:
                self.head = limit_head_depth(self.head, df, len(rp) - 1, self.T)
            df = df.copy()
            df[self.head] = 0  # the block's terms are in no shard's index: the ranges balance the tail's visits
            self.tail_terms_per_row_and_shard = float(df.sum()) / max(1, len(rp) - 1) / self.T
        self.ranges = term_ranges(df, self.T)
        self.term_range = self.ranges[self.ti]
        self.n = len(rp) - 1
        self.row_range = (self.n * self.dj // self.D, self.n * (self.dj + 1) // self.D)
        if self.head.size:
            self.engine = self.engine_factory(self.term_range, head=(self.head, self.ti, self.T, self.head_fold))
        else:
            self.engine = self.engine_factory(self.term_range)
        self.engine.load(rp, idx, val, None if self.D == 1 else self.row_range)

    def _all_gather_var(self, t):
        """all-gather (inside the term group) of 1-D int64 tensors of different lengths"""
        n = torch.tensor([t.numel()], dtype=torch.int64, device=self.comm)
        sizes = [torch.zeros_like(n) for _ in range(self.T)]
        dist.all_gather(sizes, n, group=self.group)
        sizes = [int(s.item()) for s in sizes]
        m = max(sizes + [1])
        buf = torch.zeros(m, dtype=torch.int64, device=self.comm)
        buf[: t.numel()] = t.to(self.comm)
        out = [torch.empty_like(buf) for _ in range(self.T)]
        dist.all_gather(out, buf, group=self.group)
        return torch.cat([o[:s] for o, s in zip(out, sizes)]).to(self.device), sizes

    def step(self, return_pairs=False):
        if self.T == 1 and not return_pairs and hasattr(self.engine, "join"):
            # candidate ranges only: the handle's answer is final; count it without bringing the pairs to the host
            n_mine = self.engine.join()
            st = self.engine.stats
            tot = torch.tensor([float(st.get("posting_visits", 0)), float(st.get("candidate_pairs", 0)), float(n_mine),
                                float(st.get("device_posting_visits", st.get("posting_visits", 0)))], dtype=torch.float64, device=self.comm)
            if self.world > 1:
                dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            self.last = {
                "probe_ms": st.get("probe_ms", 0.0), "build_ms": st.get("build_ms", 0.0), "thin_launches": st.get("thin_launches", 0),
                "probe_kernel": st.get("probe_kernel", ""), "head_ms": st.get("head_ms", 0.0), "head_flops": st.get("head_flops", 0.0),
                "head_terms": st.get("head_terms", 0), "head_int8": st.get("head_int8", 0),
                "posting_visits": int(tot[0].item()), "candidate_pairs": int(tot[1].item()), "device_posting_visits": int(tot[3].item()),
                "exchange": {"term_shards": 1, "candidate_ranges": self.D, "candidates_per_rank": [int(n_mine)],
                             "union": int(n_mine), "all_gather_bytes_per_rank": 0, "all_reduce_bytes": 0,
                             "term_ranges": self.ranges},
            }
            return int(tot[2].item())
        q, c = self.engine.candidates()
        if self.T == 1:
            # the handle holds the whole term space: its answer for this candidate range is already exact and pruned
            uq, uc, sizes = q, c, [int(q.numel())]
            uniq = q
            sc = getattr(self.engine, "scores", None)
            part = torch.as_tensor(sc, dtype=torch.float32, device=self.device) if sc is not None \
                else self.engine.partial(uq, uc)
            keep = torch.ones(q.numel(), dtype=torch.bool, device=self.device) if sc is not None else part >= self.theta
        else:
            key = q.to(torch.int64) * self.n + c.to(torch.int64)
            allk, sizes = self._all_gather_var(key)
            uniq = torch.unique(allk)  # sorted: every rank of the group sees the same order
            uq = torch.div(uniq, self.n, rounding_mode="floor")
            uc = uniq - uq * self.n
            part = self.engine.partial(uq, uc)
            part = part.to(self.comm)
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)  # RCCL all-reduce of partial scores
            part = part.to(self.device)
            keep = part >= self.theta
        st = getattr(self.engine, "stats", {}) or {}
        # whole-job counters: posting visits and touched pairs add up over all ranks; result pairs over the D groups
        mine = float(keep.sum().item()) if self.ti == 0 else 0.0
        tot = torch.tensor([float(st.get("posting_visits", 0)), float(st.get("candidate_pairs", 0)), mine,
                            float(st.get("device_posting_visits", st.get("posting_visits", 0)))], dtype=torch.float64, device=self.comm)
        if self.world > 1:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        self.last = {
            "probe_ms": st.get("probe_ms", 0.0), "build_ms": st.get("build_ms", 0.0), "thin_launches": st.get("thin_launches", 0),
            "probe_kernel": st.get("probe_kernel", ""), "head_ms": st.get("head_ms", 0.0), "head_flops": st.get("head_flops", 0.0),
            "head_terms": int(self.head.size), "head_int8": st.get("head_int8", 0),
            "posting_visits": int(tot[0].item()), "candidate_pairs": int(tot[1].item()), "device_posting_visits": int(tot[3].item()),
            "exchange": {"term_shards": self.T, "candidate_ranges": self.D, "candidates_per_rank": sizes,
                         "union": int(uniq.numel()), "all_gather_bytes_per_rank": 8 * max(sizes + [1]) * self.T,
                         "all_reduce_bytes": 4 * int(uniq.numel()), "term_ranges": self.ranges,
                         "tail_terms_per_row_and_shard": self.tail_terms_per_row_and_shard},
        }
        if return_pairs:  # this rank's group result (the whole result when D == 1)
            return uq[keep].cpu().numpy(), uc[keep].cpu().numpy(), part[keep].cpu().numpy()
        return int(tot[2].item())
