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
import time

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

    def load(self, rp, idx, val, row_range=None, device_arrays=None, query_spans=None):
        """the batch as host CSR (numpy), or already resident: device_arrays = (rowptr int64, idx int32, val fp32) on
        this engine's GPU (several shard engines of one process then share one copy).
        query_spans (with a row_range): the SYMMETRIC scheme across row ranges -- this cell joins its own rows with themselves
        (a whole-store join of the handle: symmetric inside, csrc DESIGN 5c) and meets only the rows of `query_spans`, a list of
        (row0, row1) spans = the ranges it owns the cross pairs with, as an outside batch; the pairs it finds there stand for
        both directions (the score is symmetric) and are mirrored by the caller.  Without it the cell meets EVERY row of the
        batch as a query (each cross pair is then found twice over the grid, once per direction)."""
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
        self.sym = (not self.whole) and query_spans is not None
        if self.sym:  # the outside batch: the rows of the spans, in span order, as their own CSR
            spans = [(int(a), int(b)) for a, b in query_spans if b > a]
            self.o_rows = torch.cat([torch.arange(a, b, dtype=torch.int64, device=dev) for a, b in spans]) if spans else \
                torch.zeros(0, dtype=torch.int64, device=dev)
            lens = (self.d_rp[1:] - self.d_rp[:-1])[self.o_rows]
            self.o_rp = torch.zeros(self.o_rows.numel() + 1, dtype=torch.int64, device=dev)
            if self.o_rows.numel():
                self.o_rp[1:] = torch.cumsum(lens, 0)
            parts_i, parts_v = [], []
            for a, b in spans:
                e0, e1 = int(self.d_rp[a].item()), int(self.d_rp[b].item())
                parts_i.append(self.d_idx[e0:e1])
                parts_v.append(self.d_val[e0:e1])
            self.o_idx = torch.cat(parts_i).contiguous() if parts_i else torch.zeros(0, dtype=torch.int32, device=dev)
            self.o_val = torch.cat(parts_v).contiguous() if parts_v else torch.zeros(0, dtype=torch.float32, device=dev)
            self.o_ids = self.o_rows.contiguous()
        self.phase = None
        torch.cuda.synchronize()

    def phases(self):
        """the handle calls of one step: one for a cell that meets the whole batch, two under the symmetric scheme"""
        return ("own", "outside") if self.sym else ("all",)

    def _merge_stats(self, st, factor):
        """statistics of a step add up over its phases; posting_visits / candidate_pairs keep counting what the reference's
        two-directional probe visits, so a mirrored phase counts twice there and once in device_posting_visits"""
        if not hasattr(self, "stats") or self.stats is None or self._fresh:
            self.stats = dict(st)
            for k in ("posting_visits", "candidate_pairs"):
                self.stats[k] = factor * st[k]
            self._fresh = False
            return
        for k in ("posting_visits", "candidate_pairs"):
            self.stats[k] += factor * st[k]
        # (build_ms is the last insert's: the own phase's, counted there)
        for k in ("device_posting_visits", "probe_ms", "head_ms", "head_flops", "thin_launches", "probe_launches", "head_survivors",
                  "filter_survivors", "rescore_ms"):
            self.stats[k] += st[k]

    def run_phase(self, phase):
        """one handle call of the step; returns the number of pairs it reports (they stay on the device)"""
        self.phase = phase
        if phase == "own":
            self.ix.clear()
            self._fresh = True
            n = self.ix.insert_and_query_dev(self.s_ids, self.s_rp, self.s_idx, self.s_val)
            self._merge_stats(self.ix.stats(), 1)
            assert self.ix.stats()["rows"] == self.r1 - self.r0, "shard engine: the handle dropped rows"
            return n
        if phase == "outside":
            n = self.ix.query_dev(self.o_ids, self.o_rp, self.o_idx, self.o_val) if self.o_rows.numel() else 0
            if self.o_rows.numel():
                self._merge_stats(self.ix.stats(), 2)
            return n
        self._fresh = True
        return self.join()

    def phase_pairs(self, n):
        """(query row, candidate row) of the phase just run, global rows, + the handle's scores"""
        q = torch.empty(n, dtype=torch.int32, device=self.device)
        c = torch.empty(n, dtype=torch.int32, device=self.device)
        sc = torch.empty(n, dtype=torch.float32, device=self.device)
        if n:
            self.ix.results_to(q, c, sc)
        self.scores = sc
        q = q.to(torch.int64)
        if self.phase == "own":
            q = q + self.r0
        elif self.phase == "outside":
            q = self.o_rows[q]
        return q, c.to(torch.int64) + self.r0

    def phase_partial(self, q_row, c_row):
        """exact partial scores of pairs of the phase just run (the handle scores against its LAST query batch)"""
        if self.phase == "own":
            ql = q_row - self.r0
        elif self.phase == "outside":
            ql = torch.searchsorted(self.o_rows, q_row) if self._o_sorted() else self._o_lookup(q_row)
        else:
            ql = q_row
        out = torch.empty(q_row.numel(), dtype=torch.float32, device=self.device)
        if q_row.numel():
            self.ix.partial_scores_dev(ql.to(torch.int32).contiguous(), (c_row - self.r0).to(torch.int32).contiguous(), out)
        return out

    def _o_sorted(self):
        return bool((self.o_rows[1:] > self.o_rows[:-1]).all().item()) if self.o_rows.numel() > 1 else True

    def _o_lookup(self, q_row):
        inv = torch.full((self.n,), -1, dtype=torch.int64, device=self.device)
        inv[self.o_rows] = torch.arange(self.o_rows.numel(), dtype=torch.int64, device=self.device)
        return inv[q_row]

    def join(self):
        """rebuild the shard's index and run its join; returns the number of pairs it reports (they stay on the device)"""
        self.ix.clear()
        if self.whole:
            n = self.ix.insert_and_query_dev(self.d_ids, self.d_rp, self.d_idx, self.d_val)
        else:
            self.ix.insert_dev(self.s_ids, self.s_rp, self.s_idx, self.s_val)
            n = self.ix.query_dev(self.d_ids, self.d_rp, self.d_idx, self.d_val)
        self.stats = self.ix.stats()
        self.phase = "all"
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


def half_spans(n, D, j):
    """The row ranges cell j of D owns the CROSS pairs with (symmetric scheme, D > 1): a pair of rows in two different ranges
    a, b is probed by exactly one of the two cells -- range j takes the (D - 1) // 2 ranges before it (cyclically) and, for even
    D, the range opposite to it when it is the lower-numbered of the two -- and reported in both directions."""
    ranges = [(n * k // D, n * (k + 1) // D) for k in range(D)]
    mine = [(j - d) % D for d in range(1, (D - 1) // 2 + 1)]
    if D % 2 == 0 and D > 1:
        k = (j + D // 2) % D
        if j < k:
            mine.append(k)
    return [ranges[k] for k in sorted(mine)]


class ShardedJoin:
    """world = T x D ranks: rank r owns term range (r % T) of the candidate rows in row range (r // T).

    T term shards share a candidate range and combine their partial scores with an all-reduce inside their group
    (the exchange the term-sharded index needs); the D candidate ranges are independent (their result sets are
    disjoint).  A term shard has 1/T of the posting visits of every (query, tile) round but the same number of rounds,
    so its speed-up trails T; candidate ranges need no data-path collective.  A shard that holds ALL rows (D = 1) joins its
    batch symmetrically (half the rounds); with D > 1 the symmetry is used ACROSS the row ranges too
    (round 4): a cell joins its own rows with themselves and meets the rows of only half of the other ranges as an outside
    batch -- a cross pair is probed by one of its two cells and reported in both directions (half_spans; symmetric_ranges).  `term_shards` = T picks the layout:
    bench.py's headline is T = world (the layout BASELINE.json names), this class's default T = 1.
    head_terms: dense-head block of the term-sharded layouts (module docstring): 0 = the library's policy decides on rank 0,
    -1 = never, N <= 32768 = that many of the most frequent terms.  (With T = 1 every handle is a plain one and decides
    for itself.)"""

    def __init__(self, dim, theta, rank, world, device, tile_rows=0, engine_factory=None, comm_device=None,
                 term_shards=None, head_terms=0, head_chooser=None, symmetric_ranges=True):
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
        # D > 1: a cell joins its own rows symmetrically and meets only HALF of the other ranges' rows as queries, reporting those
        # pairs in both directions (half_spans); False = every cell meets every row (each cross pair found once per direction)
        self.symmetric_ranges = symmetric_ranges
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
        # the partition rule's figures: postings and posting visits (sum df, sum df^2) of every term range
        d64 = df.astype(np.int64)
        self.range_df = [int(d64[lo:hi].sum()) for lo, hi in self.ranges]
        self.range_df2 = [int((d64[lo:hi] ** 2).sum()) for lo, hi in self.ranges]
        self.n = len(rp) - 1
        self.row_range = (self.n * self.dj // self.D, self.n * (self.dj + 1) // self.D)
        if self.head.size:
            self.engine = self.engine_factory(self.term_range, head=(self.head, self.ti, self.T, self.head_fold))
        else:
            self.engine = self.engine_factory(self.term_range)
        if self.D == 1:
            self.engine.load(rp, idx, val, None)
        elif self.symmetric_ranges:
            self.engine.load(rp, idx, val, self.row_range, query_spans=self.half_spans())
        else:
            self.engine.load(rp, idx, val, self.row_range)

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

    def half_spans(self):
        return half_spans(self.n, self.D, self.dj)

    def _exchange(self, q, c, partial):
        """steps 2-4 inside the term group: all-gather of candidate lists, union, exact partial scores, all-reduce, threshold"""
        key = q.to(torch.int64) * self.n + c.to(torch.int64)
        allk, sizes = self._all_gather_var(key)
        uniq = torch.unique(allk)  # sorted: every rank of the group sees the same order
        uq = torch.div(uniq, self.n, rounding_mode="floor")
        uc = uniq - uq * self.n
        part = partial(uq, uc).to(self.comm)
        dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)  # RCCL all-reduce of partial scores
        part = part.to(self.device)
        return uq, uc, part, part >= self.theta, sizes, int(uniq.numel())

    def step(self, return_pairs=False):
        eng = self.engine
        phased = hasattr(eng, "phases")
        phases = eng.phases() if phased else ("all",)
        out_q, out_c, out_s = [], [], []
        mine, sizes_all, union_all = 0.0, [], 0
        exchange_s = 0.0
        for ph in phases:
            mirrored = ph == "outside"  # the pairs of this phase stand for both directions (half_spans)
            if phased:
                n_ph = eng.run_phase(ph)
                if self.T == 1 and not return_pairs:  # candidate ranges only: the handle's answer is final; count it on the device side
                    mine += (2.0 if mirrored else 1.0) * float(n_ph)
                    sizes_all.append(int(n_ph))
                    union_all += int(n_ph)
                    continue
                q, c = eng.phase_pairs(n_ph)
                partial = eng.phase_partial
            else:
                q, c = eng.candidates()
                partial = eng.partial
            if self.T == 1:
                # the handle holds the whole term space: its answer for this candidate range is already exact and pruned
                uq, uc, sizes, n_union = q, c, [int(q.numel())], int(q.numel())
                sc = getattr(eng, "scores", None)
                part = torch.as_tensor(sc, dtype=torch.float32, device=self.device) if sc is not None else partial(uq, uc)
                keep = torch.ones(q.numel(), dtype=torch.bool, device=self.device) if sc is not None else part >= self.theta
            else:
                if torch.device(self.device).type == "cuda":
                    torch.cuda.synchronize(self.device)  # (the handle's kernels are done: what follows is the exchange's own time)
                t_ex = time.perf_counter()
                uq, uc, part, keep, sizes, n_union = self._exchange(q, c, partial)
                if torch.device(self.device).type == "cuda":
                    torch.cuda.synchronize(self.device)
                exchange_s += time.perf_counter() - t_ex
            sizes_all += sizes
            union_all += n_union
            mine += (2.0 if mirrored else 1.0) * float(keep.sum().item()) if self.ti == 0 else 0.0
            if return_pairs:
                out_q.append(uq[keep])
                out_c.append(uc[keep])
                out_s.append(part[keep])
                if mirrored:
                    out_q.append(uc[keep])
                    out_c.append(uq[keep])
                    out_s.append(part[keep])
        if self.T == 1 and not return_pairs and self.ti != 0:
            mine = 0.0
        st = getattr(eng, "stats", {}) or {}
        # whole-job counters: posting visits and touched pairs add up over all ranks; result pairs over the D groups
        tot = torch.tensor([float(st.get("posting_visits", 0)), float(st.get("candidate_pairs", 0)), mine,
                            float(st.get("device_posting_visits", st.get("posting_visits", 0)))], dtype=torch.float64, device=self.comm)
        if self.world > 1:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        # what every rank visited (the realised balance of the partition: term ranges are cut by sum df^2, row ranges by rows)
        mine_v = torch.tensor([float(st.get("device_posting_visits", st.get("posting_visits", 0)))], dtype=torch.float64, device=self.comm)
        per_rank = [torch.zeros_like(mine_v) for _ in range(self.world)]
        if self.world > 1:
            dist.all_gather(per_rank, mine_v)
        else:
            per_rank = [mine_v]
        per_rank = [float(x.item()) for x in per_rank]
        gathered = 8 * max(sizes_all + [1]) * self.T if self.T > 1 else 0
        reduced = 4 * union_all if self.T > 1 else 0
        self.last = {
            "probe_ms": st.get("probe_ms", 0.0), "build_ms": st.get("build_ms", 0.0), "thin_launches": st.get("thin_launches", 0),
            "probe_kernel": st.get("probe_kernel", ""), "head_ms": st.get("head_ms", 0.0), "head_flops": st.get("head_flops", 0.0),
            "head_terms": int(self.head.size) if self.T > 1 else st.get("head_terms", 0), "head_int8": st.get("head_int8", 0),
            "posting_visits": int(tot[0].item()), "candidate_pairs": int(tot[1].item()), "device_posting_visits": int(tot[3].item()),
            "exchange": {"term_shards": self.T, "candidate_ranges": self.D, "candidates_per_rank": sizes_all,
                         "union": union_all,
                         "all_gather_bytes_per_rank": gathered, "all_reduce_bytes": reduced, "term_ranges": self.ranges,
                         "sum_df_per_term_range": self.range_df, "sum_df2_per_term_range": self.range_df2,
                         "sum_df2_imbalance_max_over_mean": (max(self.range_df2) * len(self.range_df2) / max(1, sum(self.range_df2))),
                         "device_posting_visits_per_rank": per_rank,
                         "visits_imbalance_max_over_mean": (max(per_rank) * len(per_rank) / sum(per_rank) if sum(per_rank) > 0 else 1.0),
                         # this rank's wall time of the exchange steps (all-gather of the lists, sorted union, partial scores,
                         # all-reduce, threshold) and the bytes it received + reduced over that time: an EFFECTIVE rate of the
                         # whole exchange, well below a link's wire rate (the collectives are latency-bound at these sizes)
                         "exchange_ms": exchange_s * 1e3,
                         "exchange_effective_GBps": ((gathered * (self.T - 1) / max(1, self.T) + 2 * reduced) / exchange_s / 1e9 if exchange_s > 0 else None),
                         "tail_terms_per_row_and_shard": self.tail_terms_per_row_and_shard,
                         "symmetric_row_ranges": bool(self.D > 1 and len(phases) > 1)},
        }
        if return_pairs:  # this rank's group result (the whole result when D == 1)
            cat = lambda xs, dt: torch.cat(xs) if xs else torch.zeros(0, dtype=dt, device=self.device)  # noqa: E731
            return (cat(out_q, torch.int64).cpu().numpy(), cat(out_c, torch.int64).cpu().numpy(), cat(out_s, torch.float32).cpu().numpy())
        return int(tot[2].item())
