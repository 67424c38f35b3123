// k_probe_even: the FILTER pass of the two-pass join (k_probe_coarse's arithmetic and accumulators) restructured for
// THIN rounds -- a term shard (BASELINE.json configs[3]: a dozen terms per query and shard) or a sparse regime -- where a
// round carries a few hundred postings and its duration is a chain of serially issued instructions and latencies, not a
// throughput.  DESIGN.md 5a has the measurements behind every choice below.
//
// What k_probe_coarse pays per round whatever the round holds: every wave stages its own share of the query's terms
// (term k -> wave k % NW: descriptor loads, a wave-wide scan, strip writes for two live lanes), then adds, then -- after
// the first barrier -- an LDS read of the round's counters (survivors, overflow flag, long list), the report loop, the
// clears and the second barrier.  The T = 8 shard kernel measured 3,340 cycles per round (VALU and LDS half idle).
//
// Here:
//  * FIXED ROLES.  Waves 0 .. F-1 STAGE every round and never add; waves F .. NW-1 ADD every round and never stage;
//    each kind runs its own loop with the same two barriers per round.  F = ceil(longest query / (64 / G)), G = 1, 2 or 4
//    staging lanes per term.  A staging wave scans the round's chunk counts once, two rounds ahead, and deals the chunks
//    out EVENLY over the A = NW - F adding waves: chunk j -> adding wave j % A, strip slot j / A.
//  * NOTHING IS READ AFTER THE FIRST BARRIER.  Everything a round needs to know about itself is a fact of its staging
//    (chunks per wave, long segments, whether the window overflows -> whole-tile clear): an adding wave reads it one round
//    ahead, in the same LDS round trip as its next strip and the current adds.  A crossing is reported by the wave that
//    sees it (ballot-compacted append to the global list), not through an LDS list that every wave re-reads.
//    So an adding wave's round is: strip read + adds (one LDS round trip) -> next loads, tests -> barrier -> clears -> barrier.
//  * NO TRIPS THROUGH THE SCALAR UNIT ON THE HOT PATH.  Idle lanes add to spare LDS words / write spare strip entries
//    (selects on the address, no exec masks); first touches and crossings are counted per lane; one branch per round takes
//    everything unusual.  Loaded values are used a round after their load, by the next stage.
//
// Pipeline (round v of a workgroup = one query against the tile's accumulators):
//   round v - 5 : staging waves load the row's extent                  (load_R)
//   round v - 4 :               load the query's terms                 (load_I)
//   round v - 3 :               load the (tile, term) descriptors      (load_P)
//   round v - 2 :               scan + write chunk descriptors into strips[v % 3], long segments into longs[v % 3]
//   round v - 1 : every adding wave reads its strip and the round's facts, starts its posting loads
//   round v     : adds + crossing tests (register window; chunks past the window straight from the strip), clear
// The two workgroup barriers of round v - 2 separate the strip writes from their readers; the ring of three keeps the
// strip of round v readable during round v while round v + 2 is being staged.
//
// A round whose chunks exceed the strips (A x SLOTS) or whose long segments exceed LONGCAP is flagged at staging and
// swept by the adding waves straight from the index (one term per wave at a time): slow, unbounded, never wrong.
// Queries of more than 64 x NW / 4 terms are not served (they stay with k_probe_coarse; apss_hip.hip chooses).
#pragma once
#include "apss_kernels.hpp"

namespace apss {

template <int BLOCK, int U, int LONGCAP, bool SHARD, bool SIGNED, bool ACC8, bool MERGE>
__device__ __forceinline__ void probe_even_body(const ProbeArgs &a) {
  // (Tried for C3's full rounds: workgroups of TEN waves -- the eight adding waves of the 512-thread kernel plus two staging
  // waves, 87 VGPRs at five steps.  143 vs 111 ms: ten waves spread 3/3/2/2 over the SIMDs and the second workgroup of a CU
  // needs a SIMD to hold six of them, i.e. <= 80 VGPRs; it did not become resident.)
  constexpr int NW = BLOCK / kWave;
  static_assert(NW == 8 || NW == 16, "8 or 16 waves");
  constexpr int NS = NW - 1;  // strips per ring slot: one per adding wave (at least one wave stages)
  constexpr bool SLOT2 = BLOCK <= 512;
  constexpr uint32_t ABITS = ACC8 ? 8u : 16u;
  constexpr bool WIDE = ACC8 && BLOCK > 512;
  constexpr int CH = 16, LPC = CH / 2, GPW = kWave / LPC, WIN = GPW * U;
  constexpr int SLOTS = 2 * WIN < 48 ? 2 * WIN : (WIN > 48 ? WIN : 48);  // strip slots per wave and round (>= WIN)
  static_assert(SLOTS >= WIN, "the register window reads the first WIN slots of a strip");
  // A strip is stored GROUP-major: slot sl = u * GPW + g (window step u, lane group g) lives at entry g * GS + u, so that the
  // entries a lane group needs for two consecutive steps are adjacent and one ds_read_b128 fetches both (the adding waves'
  // strip reads were 23 % of the LDS instructions of a C3 round; GS even: 16-B alignment of every pair)
  // (the 1024-thread kernel's LDS is full to within a few hundred bytes: no padding there, pairs only where GS is even anyway)
  constexpr int GS = BLOCK > 512 ? SLOTS / GPW : ((SLOTS / GPW) + 1) & ~1;  // entries per lane group
  constexpr int SSZ = GPW * GS;                 // entries per strip
  static_assert(GPW == 8 && SLOTS % GPW == 0, "slot <-> (step, group) by shifts");
  constexpr int kLongLen = kLongLenW;
  constexpr int CBMAX = WIDE ? 131072 : (BLOCK <= 512 && !ACC8 ? 32768 : 65536);
  constexpr int APW = 32 / (int)ABITS;
  __shared__ __attribute__((aligned(16))) uint32_t acc[CBMAX / APW + kWave];  // (+ one spare word per lane: idle lanes add there)
  __shared__ __attribute__((aligned(16))) uint2 strips[3 * NS * SSZ + kWave];  // [3][NS][GPW][GS] {byte offset of the chunk's first posting, weight bits} (+ one spare entry per lane)
  __shared__ uint2 longs[3 * LONGCAP];
  __shared__ float long_w[3 * LONGCAP];
  __shared__ uint2 facts[4];  // per ring slot: {chunks of the round, bit 0: flagged for the direct sweep, bits 1..: long segments}
  __shared__ unsigned long long stat[4];
  unsigned char *const smem_raw = reinterpret_cast<unsigned char *>(acc);
  uint32_t *const facts_w = reinterpret_cast<uint32_t *>(facts);

  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid / kWave), ln = tid % kWave;
  const int cb = a.cb;
  const int F = a.flat_waves;   // waves that stage a round
  const int A = NW - F;         // waves that add a round (those not staging in it)
  const int CAP = A * SLOTS;    // chunks per round
  const float rcpA = 1.0f / (float)A;
  // lanes per term in a staging wave (1, 2 or 4): a staging lane writes the chunks k = sub, sub + G, ... of its term, so a
  // term of <= 2 G chunks costs every lane two strip writes and no loop (a wave issues its instructions one after another
  // whatever the number of live lanes: the staging wave's instruction count is on the round's critical path)
  const int LOGG = a.flat_group_log2, G = 1 << LOGG;
  const uint32_t sub = (uint32_t)ln & (uint32_t)(G - 1);
  const int tile = a.tile0 + blockIdx.x / a.n_chunks;
  const int chunk = blockIdx.x % a.n_chunks;
  // MERGED rounds (ProbeArgs::merge_log2 = m > 0): round v stages the terms of the M = 2^m query rows M v .. M v + M - 1 as
  // ONE row -- they are neighbours in the CSR arrays, so the merged row is the extent rowptr[M v] .. rowptr[M v + M) -- and a
  // candidate's accumulator holds the SUM of its M filter sums: an upper bound of each of them (non-negative weights), so a
  // candidate that stays below the threshold fails for every query of the round, and a crossing is reported for the round
  // (k_expand_merged turns it into M survivors; the exact pass / the shard's phase 2 prunes as ever).  a.nq, a.q_chunk and v
  // count ROUNDS; a.nq_rows the rows; a.q_val holds the rows' weights already divided by their shard factors.
  // NOT where a query meets ITSELF: a stored row's product with its own slot crosses the threshold by itself, and the crossing
  // would stand for its M - 1 neighbours too (one false survivor per query and shard).  A workgroup whose candidate tile holds
  // rows of its own chunk (the diagonal of a whole-store join, the last tiles of an appended batch) runs its rows ONE per round
  // (`mlog` = 0 for this workgroup, on the launch's smaller scale) and marks what it reports with kUnmergedBit.
  // (MERGE: instantiations of their own -- the handful of scalars this costs took the unmerged shard kernel to its SGPR limit: 13.0 -> 13.9 ms at T = 8)
  static_assert(!MERGE || SHARD, "merged rounds: term shards only");
  const int mlaunch = MERGE ? a.merge_log2 : 0;
  const int c0 = chunk * a.q_chunk;
  const int row_lo = c0 << mlaunch, row_hi = MERGE ? min(a.nq_rows, (c0 + a.q_chunk) << mlaunch) : 0;  // this workgroup's query rows
  const bool own_tile = mlaunch != 0 && a.q_slot_base >= 0 && (a.q_slot_base + row_lo) / cb <= tile && tile <= (a.q_slot_base + row_hi - 1) / cb;
  const int mlog = own_tile ? 0 : mlaunch;
  const int v0 = row_lo >> mlog;
  const int v1 = own_tile ? row_hi : min(a.nq, c0 + a.q_chunk);
  const int nv = own_tile ? a.nq_rows : a.nq;
  const int qtile = row_lo / cb;  // (symmetric joins: the chunk lies inside one tile)
  if (a.tri && tile > qtile) return;  // the mirrored half (ProbeArgs::tri): before any barrier, the whole workgroup
  const int64_t tile_row0 = (int64_t)tile * cb;
  const int unmerged_mark = own_tile ? kUnmergedBit : 0;
  const uint32_t lo = (uint32_t)(ln % LPC);
  const float cxs = a.cx_scale;

  const int64_t qbase = a.q_rowptr[MERGE ? min(v0 << mlog, a.nq_rows) : v0], qend = a.q_rowptr[MERGE ? min(v1 << mlog, a.nq_rows) : v1];
  const int64_t pbase = a.tile_post_base[tile], pend = a.tile_post_base[tile + 1];
  const __amdgpu_buffer_rsrc_t rs_qi =
      __builtin_amdgcn_make_buffer_rsrc((void *)(a.q_idx + qbase), 0, (int)((qend - qbase) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_qv =
      __builtin_amdgcn_make_buffer_rsrc((void *)(a.q_val + qbase), 0, (int)((qend - qbase) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_tp = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(a.tile_seg + (int64_t)tile * a.seg_stride), 0, (int)(a.seg_stride * 8), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_po =
      __builtin_amdgcn_make_buffer_rsrc((void *)(a.post_c + pbase), 0, (int)((pend - pbase) * 4), 0x00020000);
  constexpr uint32_t kOob = 0xfffffff0u;
  constexpr uint32_t kRare = 0x80000000u;  // WaveWork::info: the round needs more than the register window (see strip_loads)
  constexpr uint32_t kCountMask = 0xffffu;  // WaveWork::info bits 16..19: window steps that hold chunks of this wave
  // SKIP: a wave skips the window steps behind its last chunk (adds, tests, clears).  A wave otherwise issues all U steps
  // whether or not their slots hold postings, and the LDS pipeline pays for each; the windows of shards and of the sparse
  // regime are sized for the upper end of a varying term count.  Measured: C5's shape 244.5 -> 218.5 ms; where nearly every
  // step is needed the branches cost more than they save (C3, plain 512-thread handle: 101.1 -> 102.4 ms): not there.
  constexpr bool SKIP = BLOCK > 512 || SHARD;
  // (the 1024-thread kernel only ever skips its LAST step: with every step behind a branch C5's shape measured 245 vs 219 ms)
  constexpr int kFirstSkippable = BLOCK > 512 ? U - 1 : 0;

  for (int i = tid * 4; i < cb / APW + kWave; i += BLOCK * 4) *reinterpret_cast<uint4 *>(acc + i) = make_uint4(0u, 0u, 0u, 0u);
  if (tid < 8) facts_w[tid] = 0;
  unsigned long long my_visits = 0;
  uint32_t my_cands = 0;

  struct RowExt { int qb; int nnz; };  // as loaded: the low halves of the row's rowptr entries
  struct TermW { uint32_t term; float w, qs; bool valid; };
  struct Seg { uint32_t s, len; float w; };
  struct WaveWork {
    uint32_t info;     // (scalar) chunks of the round dealt to this wave | kRare
    uint32_t flags;    // (per lane, uniform) the round's flags as staged: bit 0 direct sweep, bits 1.. long segments
    apss_u32x2 pc[U];  // two coarse postings per lane and step
    float wq[U];       // query weight x cx_scale of the step's chunk
  };
  // Every stage only ISSUES its loads; whatever is computed from a loaded value is computed by the NEXT stage, a round
  // later (an operation on a value just loaded is a wait for the memory round trip, on the path to the barrier).  The
  // per-row facts -- row extent, shard factor -- are loaded by the staging lanes themselves as vector loads, after the
  // round's first adds: a scalar prefetched by every wave cost SGPRs the kernel does not have (the spill forced a wait
  // for the scalar load at the top of every round), and a vector load at the top of the round would sit in the in-order
  // wait for the round's postings.
  const int32_t *const rowptr_lo = reinterpret_cast<const int32_t *>(a.q_rowptr);
  const int qbase_lo = (int)(uint32_t)qbase;
  auto load_R = [&](int v) {  // raw: the low halves of rowptr[v], rowptr[v + 1]
    RowExt r;
    const int row_a = min(v, nv - 1) << mlog, row_b = MERGE ? min(row_a + (1 << mlog), a.nq_rows) : row_a + 1;
    r.qb = rowptr_lo[2 * row_a];
    r.nnz = rowptr_lo[2 * row_b];
    return r;
  };
  auto load_I = [&](const RowExt &r, const int fi, const int v) {
    TermW t;
    const int kterm = ((fi * kWave) >> LOGG) + (ln >> LOGG);
    const int nnz = v < v1 ? r.nnz - r.qb : 0;
    t.valid = kterm < nnz;
    const uint32_t off = t.valid ? (uint32_t)(r.qb - qbase_lo + kterm) * 4u : kOob;  // (< 2^31: a workgroup's slice of the batch)
    t.term = __builtin_amdgcn_raw_buffer_load_b32(rs_qi, off, 0, 0);
    t.w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_qv, off, 0, 0));
    t.qs = SHARD && !MERGE ? a.q_scale[min(v, nv - 1)] : 1.0f;  // shard rule: |q_g| / |q|, the query's half of the normalisation
    return t;
  };
  auto load_P = [&](const TermW &t) {
    Seg g;
    const apss_u32x2 sg = __builtin_amdgcn_raw_buffer_load_b64(rs_tp, t.valid ? t.term * 8u : kOob, 0, 0);
    g.s = sg.x;
    g.len = sg.y;
    // (rcp: 1 ulp; the coarse threshold's two units of slack cover the 2^-6 units it can add up to over a whole row)
    g.w = SHARD ? (t.qs > 0.f ? t.w * __builtin_amdgcn_rcpf(t.qs) : 0.f) : t.w;
    return g;
  };
  // Staging of one round by one of its F waves: a scan of the terms' chunk counts, the group shares its term's first
  // index through a DPP move, and every lane writes at most two descriptors -- to the strip, or to its own spare entry when
  // it has none (a select on the address, not an exec mask).  Everything unusual -- a long segment, a term of more than
  // 2 G chunks, a round that does not fit -- takes one branch.
  uint2 *const spare_item = strips + 3 * NS * SSZ + ln;
  auto flatten = [&](const Seg &g, const int ring) {
    uint32_t len = g.len;
    my_visits += sub == 0u ? len : 0u;
    const bool is_long = len > (uint32_t)kLongLen;
    len = is_long ? 0u : len;
    const uint32_t nch = (len + CH - 1) / CH;
    // the term's first chunk index: a scan over the staging lanes plus ONE LDS atomic per wave on the round's counter (F > 1:
    // the waves' ranges interleave in whatever order; a returning atomic per TERM on that one address measured 45 vs 32 ms)
    const uint32_t mine = sub == 0u ? nch : 0u;  // a term counts once, in the first lane of its group
    const uint32_t incl = wave_incl_scan(mine);
    uint32_t base = 0;
    if (ln == kWave - 1) base = atomicAdd(&facts_w[2 * ring], incl);
    uint32_t j0 = (uint32_t)__builtin_amdgcn_readlane((int)base, kWave - 1) + incl - mine;
    if (LOGG == 2) j0 = (uint32_t)__builtin_amdgcn_update_dpp((int)j0, (int)j0, 0x00, 0xf, 0xf, false);       // quad_perm [0,0,0,0]
    else if (LOGG == 1) j0 = (uint32_t)__builtin_amdgcn_update_dpp((int)j0, (int)j0, 0xa0, 0xf, 0xf, false);  // quad_perm [0,0,2,2]
    const uint32_t wbits = __float_as_uint(cxs * g.w);
    uint2 *const st = strips + ring * (NS * SSZ);
    auto put = [&](const uint32_t k) {
      // chunk j of the round -> adding wave j % A, slot j / A ((j + 0.5) / A is at least 1 / 2A away from an integer:
      // the float quotient truncates exactly for j < 2^16)
      const uint32_t j = min(j0 + k, (uint32_t)CAP - 1u);  // (a round that does not fit is flagged below; keep the store inside the strips)
      const uint32_t sl = (uint32_t)(((float)j + 0.5f) * rcpA);
      uint2 *const dst = k < nch ? st + (j - sl * (uint32_t)A) * SSZ + (sl & 7u) * GS + (sl >> 3) : spare_item;
      *dst = make_uint2((g.s + k * CH) * 4u, wbits);
    };
    // chunks per term written without the loop: 8 with four lanes per term (two writes each), 6 with two, 3 with one
    const uint32_t two = LOGG == 2 ? 8u : (LOGG == 1 ? 6u : 3u);
    put(sub);
    put(sub + (uint32_t)G);
    if (LOGG < 2) put(sub + 2u * (uint32_t)G);
    const bool unfit = j0 + nch > (uint32_t)CAP;
    if (__any(is_long || unfit || nch > two)) {
      bool bad = unfit;
      if (is_long && sub == 0u) {
        const uint32_t k = atomicAdd(&facts_w[2 * ring + 1], 2u) >> 1;
        if (k < (uint32_t)LONGCAP) {
          longs[ring * LONGCAP + k] = make_uint2(g.s, g.len);
          long_w[ring * LONGCAP + k] = g.w;
        } else {
          bad = true;
        }
      }
      // the round is swept straight from the index; whatever was staged is ignored
      if (bad) atomicOr(&facts_w[2 * ring + 1], 1u);
      for (uint32_t k = two + sub; __any(k < nch); k += (uint32_t)G) put(k);
    }
  };
  // the next round of this wave, first half: the round's facts and this wave's strip (LDS reads, issued ahead of the
  // current round's adds so that one LDS round trip serves both)
  struct StripRead { uint2 fc; uint2 it[U]; };
  auto strip_read = [&](StripRead &sr, const int ring, const int rank) {
    sr.fc = facts[ring];
    const uint2 *const st = strips + (ring * NS + rank) * SSZ + (ln / LPC) * GS;
    if constexpr (GS % 2 == 0) {
#pragma unroll
      for (int u = 0; u + 1 < U; u += 2) {
        const uint4 two = *reinterpret_cast<const uint4 *>(st + u);
        sr.it[u] = make_uint2(two.x, two.y);
        sr.it[u + 1] = make_uint2(two.z, two.w);
      }
      if (U & 1) sr.it[U - 1] = st[U - 1];
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) sr.it[u] = st[u];
    }
  };
  // second half: every LPC lanes take one chunk of the strip and start its posting load.  VALU only -- the one value the
  // next round branches on (is it more than a register window?) goes to the scalar unit here, a round before its use.
  auto strip_loads = [&](WaveWork &f, StripRead &sr, const int rank) {
    const uint32_t tot = (sr.fc.y & 1u) ? 0u : sr.fc.x;  // (a round flagged for the direct sweep has no chunks)
    // chunks j < tot with j % A == rank: ceil((tot - rank) / A)
    const uint32_t mine = (int)tot > rank ? (uint32_t)(((float)((int)tot - rank + A - 1) + 0.5f) * rcpA) : 0u;
    const bool rare = sr.fc.y != 0u || tot > (uint32_t)(A * WIN);
    const uint32_t steps = min((mine + (uint32_t)GPW - 1u) / (uint32_t)GPW, (uint32_t)U);
    f.info = (uint32_t)__builtin_amdgcn_readfirstlane((int)(mine | (rare ? kRare : 0u) | (steps << 16)));
    f.flags = sr.fc.y;
#pragma unroll
    for (int u = 0; u < U; ++u) asm volatile("" : "+v"(sr.it[u].x), "+v"(sr.it[u].y));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // a posting word of zero is no posting (segments are zero-padded to whole chunks; out-of-range reads return zero);
      // a strip slot past the wave's last chunk holds a stale descriptor: its load is sent out of range
      f.wq[u] = __uint_as_float(sr.it[u].y);
      f.pc[u] = __builtin_amdgcn_raw_buffer_load_b64(rs_po, (uint32_t)(u * GPW + ln / LPC) < mine ? sr.it[u].x + lo * 8u : kOob, 0, 0);
    }
  };

  // FIXED ROLES: waves 0 .. F-1 stage every round and never add, waves F .. NW-1 add every round and never stage.  The two
  // kinds run their own loops (same two barriers per round): no per-round role arithmetic on the scalar unit, and the
  // compiler allocates registers for one job at a time.
  // BARRIER PAIRING.  The two sides execute `__syncthreads()` at different places of the source: HIP only promises a barrier
  // for one textual call reached by all threads; gfx950's s_barrier counts arriving WAVES, whichever s_barrier instruction
  // they execute, so what must hold -- and does, by construction -- is that every wave of the workgroup executes the same
  // NUMBER of barriers: 2 before the loops (the prologue below, executed by all), then exactly 2 per query v in [v0, v1) on
  // both sides (staging: one iteration per v; adding: the loop steps by two and leaves after round v when v + 1 >= v1, so an
  // odd count ends after its last round), then the epilogue's.  No path inside a round skips a barrier (`rare` rounds clear
  // the whole tile between the same two).  tests/test_gpu_even.py::test_chunk_shapes_pin_the_barrier_pairing runs chunks of
  // one, two and three queries, a short last chunk and one- and two-row batches.
  // (Tried: a staging wave taking TWO sets of terms per round -- half the staging waves, one more adding wave; C3 7 x 5
  // window steps instead of 6 x 6.  The second set's registers tipped the kernel into scratch: 140 vs 101 ms.)
  const bool stager = wv < F;
  const int rank = wv - F;  // an adding wave's rank
  WaveWork wfa, wfb;
  constexpr int slack = 2;  // (k_probe_coarse: the soundness argument of the coarse threshold)
  const int thr_c = (int)a.cx_theta - slack;
  const uint32_t thr1 = (uint32_t)max(thr_c, 1) - 1u;
  {
    // rounds v0 and v0 + 1 are staged before the loops start
    const TermW I0 = load_I(load_R(v0), wv, v0), I1 = load_I(load_R(v0 + 1), wv, v0 + 1);
    const Seg P0 = load_P(I0), P1 = load_P(I1);
    __syncthreads();
    if (stager) {
      flatten(P0, 0);
      flatten(P1, 1);
    }
    __syncthreads();
    if (!stager) {
      StripRead sr;
      strip_read(sr, 0, rank);
      strip_loads(wfa, sr, rank);
    }
  }

  if (stager) {
    // ---- staging waves: round v loads the row extent of v + 5, the terms of v + 4, the descriptors of v + 3 and
    // stages v + 2 (every value is used a round after its load) ----
    RowExt R4 = load_R(v0 + 4);
    TermW Ic = load_I(load_R(v0 + 3), wv, v0 + 3);
    Seg Pc = load_P(load_I(load_R(v0 + 2), wv, v0 + 2));
    int r0 = 0;
    for (int v = v0; v < v1; ++v) {
      const int r2 = r0 == 0 ? 2 : r0 - 1;  // (v + 2) % 3
      const RowExt R5 = load_R(v + 5);
      const TermW In = load_I(R4, wv, v + 4);
      const Seg Pn = load_P(Ic);
      flatten(Pc, r2);
      __syncthreads();  // every add of the round has landed
      if (tid == 0) facts[r0] = make_uint2(0u, 0u);  // (read one round ago; staged again in the next round)
      __syncthreads();  // cleared: the next query starts from zero
      R4 = R5;
      Ic = In;
      Pc = Pn;
      r0 = r0 == 2 ? 0 : r0 + 1;
    }
  } else {
    // ---- adding waves ----
    const int atid = tid - F * kWave, ABLOCK = A * kWave;  // thread index / count among the adding waves (sweeps, whole-tile clears)
    auto round = [&](WaveWork &w0, WaveWork &w2, const int v, const int r0) {
      const int r1 = r0 == 2 ? 0 : r0 + 1;
      const int q = v;
      // a crossing, reported by the wave that sees it (uniform control flow: one global atomic per wave and call)
      auto report = [&](const bool cross, const uint32_t slot, const uint32_t sum) {
        bool ok = cross;
        if (ok && mlog == 0) ok = a.ext_id[tile_row0 + slot] != a.q_ext[q];  // (merged rounds: k_expand_merged excludes)
        const uint64_t o = wave_append(ok, &a.counters[kCtrResults]);
        if (ok && o < a.res_cap) {
          a.res_q[o] = q | unmerged_mark;
          a.res_c[o] = (int32_t)(tile_row0 + slot);
          a.res_s[o] = (float)sum / cxs;  // coarse score at the crossing, replaced by k_rescore
        }
      };
      // the same from divergent control flow (sweeps): one atomic per lane
      auto report_lane = [&](const uint32_t slot, const uint32_t sum) {
        if (mlog != 0 || a.ext_id[tile_row0 + slot] != a.q_ext[q]) {
          const uint64_t o = atomicAdd(&a.counters[kCtrResults], 1ull);
          if (o < a.res_cap) {
            a.res_q[o] = q | unmerged_mark;
            a.res_c[o] = (int32_t)(tile_row0 + slot);
            a.res_s[o] = (float)sum / cxs;
          }
        }
      };
      auto slot_of = [&](const uint32_t pcw) { return WIDE ? pcw >> 15 : (SLOT2 && !ACC8 ? (pcw & 0xffffu) >> 1 : pcw & 0xffffu); };
      auto prod = [&](const uint32_t pcw, const float wqs) {
        const float x = __builtin_fmaf(wqs, __half2float(__ushort_as_half((unsigned short)(WIDE ? pcw & 0x7fffu : pcw >> 16))), 1.0f);
        return SIGNED ? (uint32_t)max((int)x, 1) : (uint32_t)x;
      };
      auto half_of = [&](const uint32_t old_word, const uint32_t pcw) {
        if (WIDE) return __builtin_amdgcn_ubfe(old_word, (pcw >> 12) & 24u, 8u);
        return SLOT2 ? __builtin_amdgcn_ubfe(old_word, pcw << 3, ABITS) : (old_word >> ((pcw & 1u) << 4)) & 0xffffu;
      };
      constexpr int BATCH = 3;  // (one batch of all six steps of C3's window: no change, 101.1 ms)
      // An idle lane (zero word) adds into ITS OWN spare word behind the accumulators (a select on the address) instead of
      // being masked off: an exec mask around each atomic is a trip VALU -> scalar unit -> VALU (compare, s_and_saveexec,
      // s_or) that cost ~64 cycles of the wave's serial issue per posting slot (profiles/microbench/issue_rate.hip).  Its
      // "old value" is replaced by thr1 + 1 afterwards: never a first touch (not 0), never a crossing (thr1 - old wraps).
      const uint32_t spare = (uint32_t)(CBMAX / APW) * 4u + (uint32_t)ln * 4u;
      auto add_or_spare = [&](const uint32_t pcw, const uint32_t p) -> uint32_t {
        const uint32_t addr = WIDE ? (pcw >> 15) & 0x1fffcu : (SLOT2 ? pcw & 0xfffcu : ((pcw & 0xffffu) >> 1) * 4u);
        const uint32_t sh = WIDE ? (pcw >> 12) & 24u : (SLOT2 ? (pcw << 3) & 31u : (pcw & 1u) << 4);
        return atomicAdd(reinterpret_cast<uint32_t *>(smem_raw + (pcw ? addr : spare)), p << sh);
      };
      // two postings of a sweep the same way: no masks around the atomics, one (rare, divergent) branch for both crossings
      auto visit2 = [&](const uint32_t x, const uint32_t y, const float wqs) {
        const uint32_t px = prod(x, wqs), py = prod(y, wqs);
        uint32_t ox = add_or_spare(x, px), oy = add_or_spare(y, py);
        ox = x ? half_of(ox, x) : thr1 + 1u;
        oy = y ? half_of(oy, y) : thr1 + 1u;
        my_cands += (ox == 0u ? 1u : 0u) + (oy == 0u ? 1u : 0u);
        if ((thr1 - ox < px) | (thr1 - oy < py)) {
          if (thr1 - ox < px) report_lane(slot_of(x), ox + px);
          if (thr1 - oy < py) report_lane(slot_of(y), oy + py);
        }
      };
      const int n_steps = SKIP ? (int)((w0.info >> 16) & 0xfu) : U;  // (scalar: a branch on it costs no trip from the VALU)
      auto issue_batch = [&](const int u0, uint32_t (&p0)[BATCH], uint32_t (&p1)[BATCH], uint32_t (&o0)[BATCH], uint32_t (&o1)[BATCH]) {
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
          const int u = u0 + j;
          if (SKIP && u >= kFirstSkippable && u >= n_steps) break;  // (one forward exit: the steps behind this wave's last chunk)
          if (u < U) {
            p0[j] = prod(w0.pc[u].x, w0.wq[u]);
            p1[j] = prod(w0.pc[u].y, w0.wq[u]);
            o0[j] = add_or_spare(w0.pc[u].x, p0[j]);
            o1[j] = add_or_spare(w0.pc[u].y, p1[j]);
          }
        }
      };
      auto check_batch = [&](const int u0, uint32_t (&p0)[BATCH], uint32_t (&p1)[BATCH], uint32_t (&o0)[BATCH], uint32_t (&o1)[BATCH]) {
        // first touches and crossings are counted per lane (VALU only); one trip to the scalar unit per batch decides
        // whether any lane crossed
        uint32_t n_cross = 0;
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
          const int u = u0 + j;
          if (SKIP && u >= kFirstSkippable && u >= n_steps) break;
          if (u < U) {
            o0[j] = w0.pc[u].x ? half_of(o0[j], w0.pc[u].x) : thr1 + 1u;
            o1[j] = w0.pc[u].y ? half_of(o1[j], w0.pc[u].y) : thr1 + 1u;
            my_cands += (o0[j] == 0u ? 1u : 0u) + (o1[j] == 0u ? 1u : 0u);
            n_cross += (thr1 - o0[j] < p0[j] ? 1u : 0u) + (thr1 - o1[j] < p1[j] ? 1u : 0u);
          }
        }
        if (__any(n_cross != 0u)) {
#pragma unroll
          for (int j = 0; j < BATCH; ++j) {
            const int u = u0 + j;
            if (SKIP && u >= kFirstSkippable && u >= n_steps) break;
            if (u < U) {
              report(thr1 - o0[j] < p0[j], slot_of(w0.pc[u].x), o0[j] + p0[j]);
              report(thr1 - o1[j] < p1[j], slot_of(w0.pc[u].y), o1[j] + p1[j]);
            }
          }
        }
      };
      StripRead sr;
      strip_read(sr, r1, rank);
      {
        uint32_t p0[BATCH], p1[BATCH], o0[BATCH], o1[BATCH];
        issue_batch(0, p0, p1, o0, o1);
        // (nothing that waits for the strip or the facts may be scheduled ahead of the adds' issue)
        asm volatile("" : "+v"(sr.fc.x), "+v"(sr.fc.y) : : "memory");
        __builtin_amdgcn_sched_barrier(0);
        strip_loads(w2, sr, rank);
        check_batch(0, p0, p1, o0, o1);
      }
#pragma unroll
      for (int u0 = BATCH; u0 < U; u0 += BATCH) {
        uint32_t p0[BATCH], p1[BATCH], o0[BATCH], o1[BATCH];
        issue_batch(u0, p0, p1, o0, o1);
        check_batch(u0, p0, p1, o0, o1);
      }
      const bool rare = (w0.info & kRare) != 0u;  // more than the register windows: one branch per round
      if (rare) {
        const uint32_t flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)w0.flags);
        const int mc = (int)(w0.info & kCountMask);
        if (mc > WIN) {  // chunks past the register window: straight from this wave's strip
          const uint2 *const st = strips + (r0 * NS + rank) * SSZ;
          for (int c0 = WIN; c0 < mc; c0 += GPW) {
            const int c = c0 + ln / LPC;
            const int cc = min(c, SLOTS - 1);
            const uint2 it = st[(cc & 7) * GS + (cc >> 3)];
            const apss_u32x2 two = __builtin_amdgcn_raw_buffer_load_b64(rs_po, c < mc ? it.x + lo * 8u : kOob, 0, 0);
            visit2(two.x, two.y, __uint_as_float(it.y));
          }
        }
        if (flags & 1u) {  // flagged at staging: every term straight from the index, one term per adding wave at a time
          RowExt cur = load_R(v);
          cur.nnz -= cur.qb;
          cur.qb -= qbase_lo;
          const float qsv = SHARD && !MERGE ? uniform_load(a.q_scale + v) : 1.0f;
          const float iq = qsv > 0.f ? 1.0f / qsv : 0.f;
          for (int k = rank; k < cur.nnz; k += A) {
            const uint32_t off = (uint32_t)(cur.qb + k) * 4u;
            const uint32_t term = __builtin_amdgcn_raw_buffer_load_b32(rs_qi, off, 0, 0);
            const float wq_ = cxs * __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_qv, off, 0, 0)) * iq;
            const apss_u32x2 sg = __builtin_amdgcn_raw_buffer_load_b64(rs_tp, term * 8u, 0, 0);
            for (uint32_t p = 2u * (uint32_t)ln; p < sg.y; p += 2u * kWave) {
              const apss_u32x2 two = __builtin_amdgcn_raw_buffer_load_b64(rs_po, (sg.x + p) * 4u, 0, 0);
              visit2(two.x, p + 1u < sg.y ? two.y : 0u, wq_);
            }
          }
        } else {
          const uint32_t n_long = min(flags >> 1, (uint32_t)LONGCAP);
          for (uint32_t j = 0; j < n_long; ++j) {  // long segments: swept by all adding waves, two postings per lane and pass
            const uint2 sgm = longs[r0 * LONGCAP + j];
            const float wq_ = cxs * long_w[r0 * LONGCAP + j];
            for (uint32_t k = 2u * (uint32_t)atid; k < sgm.y; k += 2u * (uint32_t)ABLOCK) {
              const apss_u32x2 a0 = __builtin_amdgcn_raw_buffer_load_b64(rs_po, (sgm.x + k) * 4u, 0, 0);
              visit2(a0.x, k + 1u < sgm.y ? a0.y : 0u, wq_);
            }
          }
        }
      }
      __syncthreads();  // every add of the round has landed

      if (rare) {  // (long or direct sweeps, or chunks past the window: their postings are not in registers) whole-tile clear
        for (int i = atid * 4; i < cb / APW; i += ABLOCK * 4) *reinterpret_cast<uint4 *>(acc + i) = make_uint4(0u, 0u, 0u, 0u);
      } else {
        unsigned short *acc16w = reinterpret_cast<unsigned short *>(acc);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (SKIP && u >= kFirstSkippable && u >= n_steps) break;
          // unconditional: an idle lane (zero word) clears slot 0, which is zero at the end of a query either way
          if (WIDE) {
            smem_raw[w0.pc[u].x >> 15] = 0;
            smem_raw[w0.pc[u].y >> 15] = 0;
          } else if (SLOT2 && ACC8) {
            smem_raw[w0.pc[u].x & 0xffffu] = 0;
            smem_raw[w0.pc[u].y & 0xffffu] = 0;
          } else if (SLOT2) {
            *reinterpret_cast<unsigned short *>(smem_raw + (w0.pc[u].x & 0xffffu)) = 0;
            *reinterpret_cast<unsigned short *>(smem_raw + (w0.pc[u].y & 0xffffu)) = 0;
          } else {
            acc16w[w0.pc[u].x & 0xffffu] = 0;
            acc16w[w0.pc[u].y & 0xffffu] = 0;
          }
        }
      }
      __syncthreads();  // cleared: the next query starts from zero
    };
    int r0 = 0;
    for (int v = v0; v < v1; v += 2) {
      round(wfa, wfb, v, r0);
      r0 = r0 == 2 ? 0 : r0 + 1;
      if (v + 1 >= v1) break;
      round(wfb, wfa, v + 1, r0);
      r0 = r0 == 2 ? 0 : r0 + 1;
    }
  }
  __syncthreads();
  if (tid < 3) stat[tid] = 0;
  __syncthreads();
  atomicAdd(&stat[0], my_visits);
  atomicAdd(&stat[1], (unsigned long long)my_cands);
  __syncthreads();
  if (tid == 0) {
    const unsigned long long twice = a.tri && tile < qtile ? 2ull : 1ull;  // (both counts are symmetric in the two tiles)
    atomicAdd(&a.counters[kCtrVisits], stat[0] * twice);
    atomicAdd(&a.counters[kCtrCands], stat[1] * twice);
    atomicAdd(&a.counters[kCtrDevVisits], stat[0]);
  }
}

template <int BLOCK, int U, int LONGCAP, bool SHARD, bool SIGNED, bool ACC8>
__global__ __launch_bounds__(BLOCK, BLOCK <= 512 ? 2 * BLOCK / 256 : BLOCK / 256) void k_probe_even(const ProbeArgs a) {
  probe_even_body<BLOCK, U, LONGCAP, SHARD, SIGNED, ACC8, false>(a);
}

// the same rounds with M = 2^merge_log2 neighbouring query rows each (a term shard's thin rounds; see MERGED rounds above)
template <int U, bool ACC8>
__global__ __launch_bounds__(512, 4) void k_probe_even_merged(const ProbeArgs a) {
  probe_even_body<512, U, 128, true, false, ACC8, true>(a);
}

}  // namespace apss
