// apss_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4) of the all-pairs-similarity hot path.
//
// What each kernel restates (reference: /root/reference/core/src/main/scala/cpslab/...):
//   k_ingest_*      client/ingest pre-filters: LoadGenerator.scala:34-37 (L2 normalise), EntryProxyActor.scala:81-93
//                   (admission sum_i v_i >= theta), WriteWorkerActor.scala:188-194 (value > indexThreshold),
//                   plus the term-range restriction of a multi-GPU shard
//   k_tile_hist/scan/scatter   IndexingWorkerActor.buildInvertedIndex, IWA:61-71 (CSR rows -> per-tile posting lists)
//   k_probe         IndexingWorkerActor.querySimilarItems, IWA:74-111 + CommonUtils.calculateSimilarity, CU:98-117
//                   + the `sim >= similarityThreshold` prune of IWA:93
//   k_partial_scores  CU:98-117 restricted to a shard's dims, for (q, c) pairs named by the host
//
// Data layout in HBM: the store is CSR (int64 rowptr, int32 idx, fp32 val, int64 ext id); the index
// is tile-major CSC: candidate slots are cut into tiles of `cb` consecutive rows, tile T owns the posting
// records {u32 local slot, f32 weight} of its rows grouped by term, every (tile, term) segment starting on a 128-B
// boundary (16 postings), and a table tile_seg[T][dim] of {first posting, length} per term.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace apss {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kGroup = 16;         // lanes that share one short-segment work item (16 x 8 B = one 128-B line)
constexpr int kItemCap = 2048;     // short-segment work items per round held in LDS
constexpr int kLongCap = 256;      // long segments per round held in LDS
constexpr int kLongLen = 64;       // segments longer than this are swept by the whole workgroup
constexpr int kSurvCap = 1024;     // threshold-crossing candidates per round held in LDS
constexpr int kProbeBlock = 1024;  // threads per probe workgroup (16 waves, one workgroup per CU)

struct alignas(8) Posting {
  uint32_t slot;  // candidate slot relative to its tile
  float w;
};

// Wave-uniform load of a read-only kernel input through the SCALAR cache (s_load): the constant address space tells the
// compiler the location is invariant for the kernel's lifetime.  As an ordinary global load a per-row fact (a row extent,
// a shard factor, an external id) is a VECTOR memory instruction on the in-order vmcnt counter: issued at the top of a
// round it sits behind the round's posting loads in the wait before the first add -- a memory round trip per round.
template <typename T>
__device__ __forceinline__ T uniform_load(const T *p) {
  return *reinterpret_cast<const __attribute__((address_space(4))) T *>(reinterpret_cast<uintptr_t>(p));
}

// (kCtrDevVisits: posting visits the kernels actually made -- a symmetric whole-store join counts the mirrored half of
// kCtrVisits / kCtrCands without visiting it; kCtrSnap: the survivor count before k_mirror_survivors appends)
// (k_probe_even, merged launches: res_q carries a ROUND, or a query row | kUnmergedBit from a workgroup that ran its rows one per round)
constexpr int kUnmergedBit = 0x40000000;
// kCtrPre: the rounds the filter reported before k_expand_merged turned them into pairs; kCtrOver: the list capacity that
// k_expand_merged / k_shard_prune would have needed when their INPUT list had overflowed (a maximum, in a counter of its own:
// added to kCtrResults it would shift the slots the kernels hand out, and the list's first `cap` entries would stay unwritten)
enum Counter { kCtrResults = 0, kCtrVisits = 1, kCtrCands = 2, kCtrFlags = 3, kCtrDevVisits = 4, kCtrSnap = 5, kCtrPre = 6, kCtrOver = 7, kCtrCount = 8 };

// ---------------------------------------------------------------------------------------------------------
// wavefront ballot / prefix-sum compaction: every active lane with `pred` gets a distinct slot of a global list
// with ONE atomic per wave (the compiler would not merge a data-dependent atomicAdd by itself).
__device__ __forceinline__ uint64_t wave_append(bool pred, unsigned long long *counter) {
  const unsigned long long mask = __ballot(pred);
  if (mask == 0) return ~0ull;
  const int lane = __lane_id();
  const int leader = __ffsll((long long)mask) - 1;
  unsigned long long base = 0;
  if (lane == leader) base = atomicAdd(counter, (unsigned long long)__popcll(mask));
  base = __shfl(base, leader);
  const unsigned long long below = mask & ((1ull << lane) - 1ull);
  return pred ? base + (unsigned long long)__popcll(below) : ~0ull;
}

// ---------------------------------------------------------------------------------------------------------
// ingest

struct IngestArgs {
  int64_t n;
  int64_t nnz;            // elements in idx / val: row extents outside [0, nnz] are rejected, never dereferenced
  const int64_t *rowptr;  // [n+1], relative to the batch
  const int32_t *idx;
  const float *val;
  int32_t dim, term_lo, term_hi;
  uint32_t flags;  // APSS_FLAG_*
  float theta, index_threshold;
  const int32_t *head_pos;  // [dim] or null: terms of a shard's dense-head block (>= 0) count for no term range's sub-norm
  // pass-1 outputs
  int64_t *row_keep;  // [n] 0/1
  int64_t *row_cnt;   // [n] kept entries (0 for dropped rows)
  float *row_inv;     // [n] 1/norm (1 when not normalising)
  float *row_sub;     // [n] |x_g| / |x|: norm of the entries kept for this term range over the row's full norm
  unsigned int *flags_out;  // [0]: bit0 malformed indices, bit1 non-finite value, bit2 negative value kept; [1]: max kept row length; [2]: bits of the max squared row norm; [3]: non-empty kept rows
};

// one G-lane group per row (G = 16 for short rows, 64 for rows of about a hundred entries)
template <int G>
__global__ void k_ingest_count(IngestArgs a) {
  // per-batch summaries: combined inside the workgroup (LDS atomics) over ALL the rows it loops over, then ONE global
  // atomic each per workgroup -- same-address global atomics are serialised at the memory side (a million of them were
  // 4x the cost of everything else here, 250k still most of it)
  __shared__ unsigned sh[4];
  if (threadIdx.x < 4) sh[threadIdx.x] = 0;
  __syncthreads();
  const int gl = threadIdx.x % G;
  const int64_t rows_per_sweep = (int64_t)gridDim.x * (blockDim.x / G);
  for (int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G; row - threadIdx.x / G < a.n; row += rows_per_sweep) {
  int64_t b = 0, e = 0;
  if (row < a.n) {
    b = a.rowptr[row];
    e = a.rowptr[row + 1];
  }
  unsigned bad = 0;
  if (b < 0 || e < b || e > a.nnz || (row == 0 && b != 0)) {  // malformed extents (a batch's rowptr starts at 0): flag the batch and read nothing
    bad |= 1;
    b = e = 0;
  }
  const bool normalise = (a.flags & 4u) != 0;  // without it one pass over the row does everything
  float sumsq = 0.f, sum = 0.f, sub = 0.f, full = 0.f;
  int cnt = 0;
  auto account = [&](const float v, const int32_t t) {
    sum += v;  // EPA:89 with max-weight 1.0
    const bool keep_v = !(a.flags & 1u) || v > a.index_threshold;  // WWA:192: the pruned vector is what is scored
    if (keep_v) full += v * v;
    const bool keep = keep_v && t >= a.term_lo && t < a.term_hi;
    if (keep) {
      cnt++;
      // (a shard with a dense-head block: the block's terms are a part of their own in the {H, T_1 .. T_T} partition of the
      // shard rule; an entry of the block that lies in this range is stored for the exact partial score, but |x_g| is the
      // norm of the range WITHOUT it)
      if (!a.head_pos || a.head_pos[t] < 0) sub += v * v;
      if (v < 0.f) bad |= 4;
    }
  };
  for (int64_t k = b + gl; k < e; k += G) {
    const float v = a.val[k];
    const int32_t t = a.idx[k];
    if (!(t >= 0 && t < a.dim) || (k > b && a.idx[k - 1] >= t)) bad |= 1;  // SV:75 strictly increasing, < size
    if (!isfinite(v)) bad |= 2;
    sumsq += v * v;
    if (!normalise) account(v, t);
  }
  float inv = 1.0f;
  if (normalise) {
    for (int o = G / 2; o; o >>= 1) sumsq += __shfl_xor(sumsq, o, G);
    inv = sumsq > 0.f ? 1.0f / sqrtf(sumsq) : 0.f;  // LG:35-37: values / sqrt(foldLeft(sum + v*v))
    for (int64_t k = b + gl; k < e; k += G) account(a.val[k] * inv, a.idx[k]);
  }
  for (int o = G / 2; o; o >>= 1) {
    sum += __shfl_xor(sum, o, G);
    full += __shfl_xor(full, o, G);
    sub += __shfl_xor(sub, o, G);
    cnt += __shfl_xor(cnt, o, G);
    bad |= __shfl_xor(bad, o, G);
  }
  if (gl == 0 && row < a.n) {
    const bool admit = !(a.flags & 2u) || sum >= a.theta;  // EPA:89
    a.row_keep[row] = admit ? 1 : 0;
    a.row_cnt[row] = admit ? cnt : 0;
    a.row_inv[row] = inv;
    // the shard rule's scale: |x_g| / |x| (x_g = the row restricted to this handle's term range), rounded DOWN a hair: as a
    // factor of a threshold and as a divisor of a weight that errs on the side of reporting more
    a.row_sub[row] = full > 0.f ? fminf(1.0f, sqrtf(sub / full)) * 0.999999f : 0.f;
    if (bad) atomicOr(&sh[0], bad);
    if (admit) {
      atomicMax(&sh[1], (unsigned)cnt);         // longest kept row: picks the probe kernel
      atomicMax(&sh[2], __float_as_uint(full));  // largest squared L2 norm of a row (fixed-point range: a shard's
                                                 // normalised partial p_g / (r_q r_c) is bounded by |q||c|, like a whole score)
      if (cnt > 0) atomicAdd(&sh[3], 1u);       // rows with at least one kept entry
    }
  }
  }  // rows of this workgroup
  __syncthreads();
  if (threadIdx.x == 0) {
    if (sh[0]) atomicOr(a.flags_out, sh[0]);
    if (sh[1]) atomicMax(a.flags_out + 1, sh[1]);
    if (sh[2]) atomicMax(a.flags_out + 2, sh[2]);
    if (sh[3]) atomicAdd(a.flags_out + 3, sh[3]);
  }
}

struct IngestWriteArgs {
  IngestArgs in;
  const int64_t *row_dst;  // [n+1] exclusive scan of row_keep; null = no row is dropped
  const int64_t *nnz_dst;  // [n+1] exclusive scan of row_cnt
  int64_t dst_row0, dst_nnz0;  // where the batch lands in the destination arrays
  int64_t *o_rowptr;  // destination rowptr (absolute offsets); o_rowptr[dst_row0 + r + 1] is written
  int32_t *o_idx;
  float *o_val;
  int64_t *o_ext;
  float *o_sub;  // sub-norm per destination row (may be null)
  uint32_t *o_erow;  // destination row of every kept entry (store only; the LDS index build streams entries)
  const int64_t *ext;
};

__global__ void k_ingest_write(IngestWriteArgs w) {
  const IngestArgs &a = w.in;
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
  const int gl = threadIdx.x % kGroup;
  if (row >= a.n) return;
  if (!a.row_keep[row]) return;
  const int64_t b = a.rowptr[row], e = a.rowptr[row + 1];
  const float inv = a.row_inv[row];
  const int64_t dr = w.dst_row0 + (w.row_dst ? w.row_dst[row] : row);  // no row dropped: identity placement
  int64_t out = w.dst_nnz0 + w.nnz_dst[row];
  for (int64_t k0 = b; k0 < e; k0 += kGroup) {
    const int64_t k = k0 + gl;
    bool keep = false;
    float v = 0.f;
    int32_t t = 0;
    if (k < e) {
      v = a.val[k] * inv;
      t = a.idx[k];
      keep = (!(a.flags & 1u) || v > a.index_threshold) && t >= a.term_lo && t < a.term_hi;
    }
    // ordered compaction inside the 16-lane group
    const unsigned long long m = __ballot(keep);
    const int lane = __lane_id();
    const int gbase = lane & ~(kGroup - 1);
    const unsigned gm = (unsigned)((m >> gbase) & 0xffffu);
    if (keep) {
      const int64_t o = out + __popc(gm & ((1u << gl) - 1u));
      w.o_idx[o] = t;
      w.o_val[o] = v;
      if (w.o_erow) w.o_erow[o] = (uint32_t)dr;
    }
    out += __popc(gm);
  }
  if (gl == 0) {
    w.o_rowptr[dr + 1] = w.dst_nnz0 + w.nnz_dst[row] + a.row_cnt[row];
    w.o_ext[dr] = w.ext[row];
    if (w.o_sub) w.o_sub[dr] = a.row_sub[row];
  }
}

// single-workgroup exclusive scan of int64 (n+1 outputs, out[n] = total); ingest-only, n <= a few million
__global__ __launch_bounds__(1024) void k_scan_i64(const int64_t *in, int64_t *out, int64_t n) {
  __shared__ int64_t part[1024];
  const int tid = threadIdx.x;
  const int64_t per = (n + 1023) / 1024;
  const int64_t b = (int64_t)tid * per, e = b + per < n ? b + per : n;
  int64_t s = 0;
  for (int64_t i = b; i < e; ++i) s += in[i];
  part[tid] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    int64_t v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int64_t run = part[tid] - s;
  for (int64_t i = b; i < e; ++i) {
    const int64_t v = in[i];
    out[i] = run;
    run += v;
  }
  if (tid == 1023) out[n] = part[1023];
}

// the same scan for large n, three launches: k_scan_sums (one sum per block of kScanBlock elements), k_scan_i64 over the
// block sums, k_scan_apply (block-local exclusive scan + the block's offset).  The single-workgroup kernel alone walked a
// million elements with one strided stream per thread: 3.6 ms per call, two calls per ingest of a term shard (7 of the
// 33 ms of a T = 8 shard's step).
constexpr int kScanBlock = 4096;  // elements per 1024-thread block, 4 consecutive per thread
template <typename T>
__device__ __forceinline__ T block_excl_scan_1024(T v, T *wave_tot /*[17] LDS*/, T *total) {
  const int tid = threadIdx.x, ln = tid % kWave, wv = tid / kWave;
  T x = v;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const T y = __shfl_up(x, o);
    if (ln >= o) x += y;
  }
  if (ln == kWave - 1) wave_tot[wv] = x;
  __syncthreads();
  if (tid == 0) {
    T run = 0;
    for (int w = 0; w < 16; ++w) {
      const T t = wave_tot[w];
      wave_tot[w] = run;
      run += t;
    }
    wave_tot[16] = run;
  }
  __syncthreads();
  *total = wave_tot[16];
  return wave_tot[wv] + x - v;
}
__global__ __launch_bounds__(1024) void k_scan_sums(const int64_t *in, int64_t *sums, int64_t n) {
  __shared__ int64_t wave_tot[17];
  const int64_t base = (int64_t)blockIdx.x * kScanBlock + (int64_t)threadIdx.x * 4;
  int64_t s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) s += base + k < n ? in[base + k] : 0;
  int64_t total;
  (void)block_excl_scan_1024(s, wave_tot, &total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void k_scan_apply(const int64_t *in, const int64_t *block_off, int64_t *out, int64_t n, int64_t n_blocks) {
  __shared__ int64_t wave_tot[17];
  const int64_t base = (int64_t)blockIdx.x * kScanBlock + (int64_t)threadIdx.x * 4;
  int64_t v[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = base + k < n ? in[base + k] : 0;
    s += v[k];
  }
  int64_t total;
  int64_t run = block_off[blockIdx.x] + block_excl_scan_1024(s, wave_tot, &total);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = block_off[n_blocks];
}

// ---------------------------------------------------------------------------------------------------------
// index build (IWA:61-71): for rows [row0, row1) of the store, fill the posting lists of their tiles.
// tile_seg[T][t] = {first posting, length} of term t in tile T, relative to the tile's posting base.  Every
// segment starts on a 128-B boundary (a multiple of kSegAlign postings): an L2 line then never straddles two
// segments, which cut the probe's memory-side traffic by a quarter (profiles/r01_summary.md).
// k_tile_hist counts lengths into .y; k_tile_scan turns them into aligned starts (.x) and clears .y;
// k_tile_scatter's cursor increments rebuild .y while placing the postings.
constexpr int kSegAlign = 16;   // exact index: 16 postings of 8 B = one 128-B line
constexpr int kSegAlignC = 32;  // coarse index: 32 postings of 4 B = one 128-B line

// coarse posting: u16 slot field | fp16 weight << 16 (4 B).  Read only by the coarse filter pass; every pair it
// lets through is re-scored from the fp32 store.  In tiles of <= 32768 rows the slot field holds slot * 2 = the LDS byte
// offset of the candidate's 16-bit accumulator (one AND gives the atomic's word address, one shift its half).
__device__ __forceinline__ uint32_t pack_coarse(uint32_t slot, float w) {
  // never the zero word: zero marks padding and idle lanes in the probe (a weight below fp16's range becomes its
  // smallest subnormal, which errs upward: safe for a filter)
  return (slot & 0xffffu) | (max((uint32_t)__half_as_ushort(__float2half_rn(w)), 1u) << 16);
}

// wide coarse posting (tiles of up to 131072 rows, 8-bit accumulators): slot << 15 | fp16 weight without its sign bit
// (weights are non-negative on this path); never the zero word either
__device__ __forceinline__ uint32_t pack_coarse_wide(uint32_t slot, float w) {
  return (slot << 15) | max((uint32_t)__half_as_ushort(__float2half_rn(w)) & 0x7fffu, 1u);
}

struct BuildArgs {
  const int64_t *rowptr;
  const int32_t *idx;             // term of every entry (a handle with a dense-head block builds from its tail view: apss_head.hpp)
  const float *val;
  int64_t row0, row1;
  int32_t cb;
  int32_t dim;
  uint2 *tile_seg;
  int64_t seg_stride;             // dim
  const int64_t *tile_post_base;  // [n_tiles + 1] first posting of each tile in `post`
  Posting *post;                  // exact format (8 B) ...
  uint32_t *post_c;               // ... or coarse format (4 B), when `coarse`
  int32_t coarse;
  int32_t coarse_shift;           // 1: the slot field holds slot * 2, the byte offset of the 16-bit accumulator (tiles <= 32768 rows)
  int32_t seg_align;              // postings per aligned unit (kSegAlign / kSegAlignC)
  int32_t coarse_wide;            // 1: pack_coarse_wide (17-bit slots)
  const uint32_t *erow;           // store row of every entry (LDS build)
  const float *row_scale;         // shard rule (coarse rendering): row r's weights are stored divided by row_scale[r] = |x_g| / |x|,
                                  //   so that the filter's threshold does not depend on the candidate (null: as they are)
  int32_t range_terms;            // terms per range of the LDS build's (tile, range) workgroups: 0 = kBuildRange; the bucketed build
                                  //   of large dims names its own (8192: measured)
  const int64_t *ent_base;        // LDS build over BUCKETED entries (dims of more than kBuildMaxRanges ranges): workgroup w = (tile, range)
                                  //   reads entries [ent_base[w], ent_base[w + 1]) of idx / erow / val, which then point at the
                                  //   bucketed copies (k_bucket_scatter); null: the tile's entries as they lie in the store
};

// one wave per row: coalesced reads of the row's entries
__global__ void k_tile_hist(BuildArgs a) {
  const int64_t row = a.row0 + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const int lane = threadIdx.x % kWave;
  if (row >= a.row1) return;
  uint2 *sg = a.tile_seg + (row / a.cb) * a.seg_stride;
  const int64_t b = a.rowptr[row], e = a.rowptr[row + 1];
  for (int64_t k = b + lane; k < e; k += kWave) {
    const int32_t t = a.idx[k];
    if ((uint32_t)t < (uint32_t)a.dim) atomicAdd(&sg[t].y, 1u);
  }
}

// Exclusive scan of a tile's aligned segment lengths -> segment starts; tile_total[tile] = postings incl. padding.  Two
// launches over (tile, block of kScanBlock terms) workgroups: k_tile_scan_part leaves every block's sum (and the build's
// statistics), k_tile_scan_place adds up the sums of the blocks before its own and writes the starts.  (One workgroup per
// tile walking its dim entries block after block took 85 us at dim = 100k whatever the number of tiles -- 25 dependent
// trips -- which a streamed batch paid on every append to the last tile; a block per workgroup is one trip.)
__global__ __launch_bounds__(1024) void k_tile_scan_part(const uint2 *tile_seg, int64_t seg_stride, int32_t dim, int64_t tile0,
                                                         int32_t n_blk, uint32_t align, uint32_t *part, uint32_t *max_len,
                                                         unsigned long long *chunk_w, uint32_t count_long) {
  __shared__ uint32_t wave_tot[17];
  const int64_t tile = tile0 + blockIdx.x / n_blk;
  const int32_t blk = (int32_t)(blockIdx.x % n_blk);
  const uint2 *sg = tile_seg + tile * seg_stride;
  const int tid = threadIdx.x;
  const int32_t i0 = blk * kScanBlock + tid * 4;
  uint32_t s = 0, longest = 0, n_long = 0;
  unsigned long long cw = 0;  // sum over the tile's short segments of length x 16-posting chunks
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t len = i0 + k < dim ? sg[i0 + k].y : 0u;
    s += (len + align - 1) / align * align;
    longest = max(longest, len);
    n_long += len > 256u ? 1u : 0u;  // (kLongLenW: what the probe kernels sweep as a long segment)
    cw += len <= 256u ? (unsigned long long)len * ((len + 15u) / 16u) : 0ull;
  }
  uint32_t total;
  (void)block_excl_scan_1024<uint32_t>(s, wave_tot, &total);
  if (tid == 0) part[blockIdx.x] = total;
  // the longest (tile, term) segment of the build: tells the probe whether its long-segment machinery is needed at all.
  // Combined per workgroup first (LDS), then ONE global atomic each per workgroup: same-address global atomics are serialised
  // at the memory side (a wave-level atomic per statistic made this kernel 215 us over 31 tiles x 25 blocks)
  __shared__ uint32_t sh_long[2];
  __shared__ unsigned long long sh_cw;
  if (tid == 0) {
    sh_long[0] = sh_long[1] = 0u;
    sh_cw = 0ull;
  }
  __syncthreads();
  for (int o = kWave / 2; o; o >>= 1) {
    longest = max(longest, (uint32_t)__shfl_xor((int)longest, o));
    n_long += (uint32_t)__shfl_xor((int)n_long, o);
  }
  // A row of the tile holds term t with probability len_t / rows, and a query holding t meets ceil(len_t / 16) chunks of this
  // tile: sum_t len_t * ceil(len_t / 16) / rows = the chunks an average round (a query distributed like the tile's rows) deals
  // out, whatever the term distribution -- what the probe sizes its register window from
  for (int o = kWave / 2; o; o >>= 1) cw += __shfl_xor(cw, o);
  if ((tid % kWave) == 0) {
    if (longest) atomicMax(&sh_long[0], longest);
    if (n_long) atomicAdd(&sh_long[1], n_long);
    if (cw) atomicAdd(&sh_cw, cw);
  }
  __syncthreads();
  if (tid == 0) {
    if (max_len && sh_long[0]) {
      atomicMax(max_len, sh_long[0]);
      if (sh_long[1] && count_long) atomicAdd(max_len + 1, sh_long[1]);  // [1]: long segments over all tiles of the build (an appended-to tile was counted before)
    }
    if (chunk_w && sh_cw) atomicAdd(chunk_w, sh_cw);
  }
}

__global__ __launch_bounds__(1024) void k_tile_scan_place(uint2 *tile_seg, int64_t seg_stride, int32_t dim, int64_t tile0, int32_t n_blk,
                                                          uint32_t align, uint32_t keep_len, const uint32_t *part, int64_t *tile_total) {
  __shared__ uint32_t wave_tot[17];
  __shared__ uint32_t before;
  const int64_t tile = tile0 + blockIdx.x / n_blk;
  const int32_t blk = (int32_t)(blockIdx.x % n_blk);
  uint2 *sg = tile_seg + tile * seg_stride;
  const int tid = threadIdx.x;
  // the sums of this tile's blocks before this one
  uint32_t mine = 0u;
  for (int32_t j = tid; j < blk; j += 1024) mine += part[(blockIdx.x / n_blk) * n_blk + j];
  uint32_t tot_before;
  (void)block_excl_scan_1024<uint32_t>(mine, wave_tot, &tot_before);
  if (tid == 0) before = tot_before;
  __syncthreads();  // (wave_tot is reused below)
  const uint32_t carry = before;
  const int32_t i0 = blk * kScanBlock + tid * 4;
  uint32_t len[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    len[k] = i0 + k < dim ? sg[i0 + k].y : 0u;
    s += (len[k] + align - 1) / align * align;
  }
  uint32_t total;
  uint32_t run = carry + block_excl_scan_1024<uint32_t>(s, wave_tot, &total);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // (the atomic scatter rebuilds .y as its cursor, the LDS scatter never touches it)
    if (i0 + k < dim) sg[i0 + k] = make_uint2(run, keep_len ? len[k] : 0u);
    run += (len[k] + align - 1) / align * align;
  }
  if (blk == n_blk - 1 && tid == 0) tile_total[tile] = (int64_t)carry + (int64_t)total;
}

__global__ void k_tile_scatter(BuildArgs a) {
  const int64_t row = a.row0 + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const int lane = threadIdx.x % kWave;
  if (row >= a.row1) return;
  const int64_t tile = row / a.cb;
  uint2 *sg = a.tile_seg + tile * a.seg_stride;
  const int64_t pbase = a.tile_post_base[tile];
  const uint32_t local = (uint32_t)(row - tile * a.cb);
  const int64_t b = a.rowptr[row], e = a.rowptr[row + 1];
  for (int64_t k = b + lane; k < e; k += kWave) {
    const int32_t t = a.idx[k];
    if ((uint32_t)t >= (uint32_t)a.dim) continue;
    // one 64-bit returning atomic on {start, cursor}: bumps the cursor (high word) and brings the start along
    const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long *>(&sg[t]), 1ull << 32);
    const uint32_t pos = (uint32_t)old + (uint32_t)(old >> 32);
    if (a.coarse) {
      const float sc = a.row_scale ? a.row_scale[row] : 1.0f;
      const float wv_ = sc > 0.f ? a.val[k] / sc : a.val[k];
      a.post_c[pbase + pos] = a.coarse_wide ? pack_coarse_wide(local, wv_) : pack_coarse(local << a.coarse_shift, wv_);
    } else {
      Posting p;
      p.slot = local;
      p.w = a.val[k];
      a.post[pbase + pos] = p;
    }
  }
}

// APPEND build (a streamed batch lands in the last, partly filled tile: IndexingWorkerActor.scala:61-71 appends to its posting
// lists): the new rows' terms are counted ON TOP of the tile's segment lengths (k_tile_hist over the new rows only), the scan
// gives every segment its new start -- a segment only ever grows, the ones behind it move right --, this kernel moves the
// tile's old postings from a scratch copy to their new places and leaves every cursor behind them, and k_tile_scatter places
// the new rows' postings after them.  Work: the new rows + one pass over the tile's postings, instead of recounting and
// re-scattering every row of the tile with two global atomics each.  16 lanes per term.
struct ShiftArgs {
  const uint2 *seg_old;   // [dim] {start, length} before the append (scratch copy)
  uint2 *seg_new;         // [dim] {new start, 0} from k_tile_scan; .y becomes the old length (the scatter's cursor)
  const uint32_t *old_c;  // the tile's old postings (scratch copy, offsets relative to the tile), coarse ...
  const Posting *old_x;   // ... or exact
  uint32_t *new_c;        // the tile's region in the posting array
  Posting *new_x;
  int32_t dim;
};
__global__ void k_tile_shift(ShiftArgs a) {
  const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
  const int gl = threadIdx.x % kGroup;
  if (t >= a.dim) return;
  const uint2 o = a.seg_old[t];
  const uint32_t n0 = a.seg_new[t].x;
  if (a.old_c) {
    for (uint32_t i = gl; i < o.y; i += kGroup) a.new_c[n0 + i] = a.old_c[o.x + i];
  } else {
    for (uint32_t i = gl; i < o.y; i += kGroup) a.new_x[n0 + i] = a.old_x[o.x + i];
  }
  if (gl == 0) a.seg_new[t].y = o.y;
}

// ---------------------------------------------------------------------------------------------------------
// LDS index build, for dims of at most kBuildMaxRanges ranges of kBuildRange terms: one workgroup per (tile, term range)
// streams the tile's entries and keeps the range's counters / cursors in LDS, so the 2 x 1e8 global atomics of
// k_tile_hist / k_tile_scatter (2.7e10/s on this part, whatever their scope) become LDS atomics.  Each range re-reads
// the tile's entries, which is why large dims (C5: 1M terms = 31 ranges) stay with the global-atomic kernels.
constexpr int kBuildRange = 16384;    // 64 KB of LDS counters: two workgroups per CU (C3: 217 workgroups instead of 124; build 5.0 -> 3.8 ms)
constexpr int kBuildMaxRanges = 16;

__global__ __launch_bounds__(1024) void k_tile_hist_lds(BuildArgs a, int64_t tile0, int32_t n_ranges) {
  __shared__ uint32_t cnt[kBuildRange];
  const int tid = threadIdx.x;
  const int64_t tile = tile0 + blockIdx.x / n_ranges;
  const int32_t rt = a.range_terms ? a.range_terms : kBuildRange;
  const int32_t lo = (int32_t)(blockIdx.x % n_ranges) * rt;
  const uint32_t span = (uint32_t)(min(a.dim, lo + rt) - lo);
  for (int i = tid; i < rt; i += 1024) cnt[i] = 0u;
  __syncthreads();
  const int64_t rA = tile * a.cb, rB = min(a.row1, rA + (int64_t)a.cb);
  const int64_t eA = a.ent_base ? a.ent_base[blockIdx.x] : a.rowptr[rA], eB = a.ent_base ? a.ent_base[blockIdx.x + 1] : a.rowptr[rB];
  for (int64_t k = eA + tid; k < eB; k += 4096) {  // four loads in flight per thread
    int32_t t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = k + 1024 * j < eB ? a.idx[k + 1024 * j] : -1;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if ((uint32_t)(t[j] - lo) < span) atomicAdd(&cnt[t[j] - lo], 1u);
  }
  __syncthreads();
  uint2 *sg = a.tile_seg + tile * a.seg_stride + lo;
  for (uint32_t i = tid; i < span; i += 1024) sg[i] = make_uint2(0u, cnt[i]);
}

__global__ __launch_bounds__(1024) void k_tile_scatter_lds(BuildArgs a, int64_t tile0, int32_t n_ranges) {
  __shared__ uint32_t cur[kBuildRange];
  const int tid = threadIdx.x;
  const int64_t tile = tile0 + blockIdx.x / n_ranges;
  const int32_t rt = a.range_terms ? a.range_terms : kBuildRange;
  const int32_t lo = (int32_t)(blockIdx.x % n_ranges) * rt;
  const uint32_t span = (uint32_t)(min(a.dim, lo + rt) - lo);
  const uint2 *sg = a.tile_seg + tile * a.seg_stride + lo;
  for (uint32_t i = tid; i < span; i += 1024) cur[i] = sg[i].x;  // cursor = segment start: the atomic returns the position
  __syncthreads();
  const int64_t pbase = a.tile_post_base[tile];
  const int64_t rA = tile * a.cb, rB = min(a.row1, rA + (int64_t)a.cb);
  const int64_t eA = a.ent_base ? a.ent_base[blockIdx.x] : a.rowptr[rA], eB = a.ent_base ? a.ent_base[blockIdx.x + 1] : a.rowptr[rB];
  // every entry's row and value are loaded with its term, matching or not: three times the read traffic of this pass
  // (it comes from the caches) instead of a second dependent memory round trip for the quarter that matches
  for (int64_t k = eA + tid; k < eB; k += 4096) {
    int32_t t[4];
    uint32_t er[4];
    float vv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t e = k + 1024 * j;
      const bool in = e < eB;
      t[j] = in ? a.idx[e] : -1;
      er[j] = in ? a.erow[e] : 0u;
      vv[j] = in ? a.val[e] : 0.f;
    }
    uint32_t pos[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if ((uint32_t)(t[j] - lo) < span) pos[j] = atomicAdd(&cur[t[j] - lo], 1u);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if ((uint32_t)(t[j] - lo) < span) {
        const uint32_t local = er[j] - (uint32_t)rA;
        if (a.coarse) {
          const float sc = a.row_scale ? a.row_scale[er[j]] : 1.0f;
          const float wv_ = sc > 0.f ? vv[j] / sc : vv[j];
          a.post_c[pbase + pos[j]] = a.coarse_wide ? pack_coarse_wide(local, wv_) : pack_coarse(local << a.coarse_shift, wv_);
        } else {
          Posting p;
          p.slot = local;
          p.w = vv[j];
          a.post[pbase + pos[j]] = p;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// BUCKETED entries for the LDS build of LARGE dims (more than kBuildMaxRanges ranges of kBuildRange terms: vectorDim = 2^20,
// the reference's HashingTF default).  The LDS build re-reads a tile's entries once per term range -- 64 times at dim = 2^20
// -- which is why such dims kept the global-atomic kernels (two atomics per posting: 39 of the 161 ms of a C5-shaped step,
// 0.59 of configs[4]'s 13.9 s).  Here a tile's entries are first PARTITIONED by term range, a counting sort in two passes
// over (tile, slice) workgroups: count per range in LDS, one global atomic per (workgroup, range) to reserve a run of the
// range's bucket, scatter through LDS cursors.  The LDS build's workgroup (tile, range) then reads ITS bucket only
// (BuildArgs::ent_base).  Order inside a bucket -- hence inside a posting segment -- is whatever the atomics give, as it
// is with the atomic build.
constexpr int kBucketSlices = 64;       // workgroups per tile
constexpr int kBucketMaxRanges = 1024;  // dims up to 2^24
struct BucketArgs {
  const int64_t *rowptr;
  const int32_t *idx;
  const float *val;
  const uint32_t *erow;
  int64_t row1;          // rows of the build end here
  int32_t cb, n_ranges, range_terms;
  int64_t tile0;
  unsigned long long *bucket_cnt;       // [tiles x n_ranges] entries per (tile, range)        (pass 1 out)
  const int64_t *bucket_base;           // [tiles x n_ranges + 1] their exclusive scan          (pass 2 in)
  unsigned long long *bucket_cur;       // [tiles x n_ranges] entries placed so far             (pass 2)
  int32_t *o_idx;
  uint32_t *o_erow;
  float *o_val;
};

template <bool SCATTER>
__global__ __launch_bounds__(1024) void k_bucket_pass(BucketArgs a) {
  __shared__ uint32_t cnt[kBucketMaxRanges];
  __shared__ int64_t base[kBucketMaxRanges];
  const int tid = threadIdx.x;
  const int64_t tl = blockIdx.x / kBucketSlices;  // tile of the build (0-based)
  const int sl = blockIdx.x % kBucketSlices;
  const int64_t rA = (a.tile0 + tl) * a.cb, rB = min(a.row1, rA + (int64_t)a.cb);
  const int64_t eA = a.rowptr[rA], eB = a.rowptr[rB];
  const int64_t per = (eB - eA + kBucketSlices - 1) / kBucketSlices;
  const int64_t k0 = eA + sl * per, k1 = min(eB, k0 + per);
  for (int i = tid; i < a.n_ranges; i += 1024) cnt[i] = 0u;
  __syncthreads();
  for (int64_t k = k0 + tid; k < k1; k += 1024) atomicAdd(&cnt[(uint32_t)a.idx[k] / (uint32_t)a.range_terms], 1u);
  __syncthreads();
  unsigned long long *const dst = (SCATTER ? a.bucket_cur : a.bucket_cnt) + tl * a.n_ranges;
  for (int i = tid; i < a.n_ranges; i += 1024) {
    const uint32_t c = cnt[i];
    if (c) {
      const unsigned long long at = atomicAdd(&dst[i], (unsigned long long)c);  // (pass 2: the start of this workgroup's run)
      if (SCATTER) base[i] = a.bucket_base[tl * a.n_ranges + i] + (int64_t)at;
    }
    if (SCATTER) cnt[i] = 0u;  // becomes the run's cursor
  }
  if (!SCATTER) return;
  __syncthreads();
  for (int64_t k = k0 + tid; k < k1; k += 1024) {
    const int32_t t = a.idx[k];
    const uint32_t r = (uint32_t)t / (uint32_t)a.range_terms;
    const int64_t p = base[r] + atomicAdd(&cnt[r], 1u);
    a.o_idx[p] = t;
    a.o_erow[p] = a.erow[k];
    a.o_val[p] = a.val[k];
  }
}

// Bank-aware posting order inside a segment (coarse rendering, 16-bit accumulators, segments of <= 64 postings).  In the
// probe, lane j of a chunk's 8 lanes adds postings 2j and 2j + 1 of the chunk, and one atomic instruction covers the same
// position of 8 chunks (8 different segments).  Postings are ordered so that position 2j, 2j + 1 of a chunk hold candidates of
// LDS bank class j (bank = (slot / 2) % 64, class = bank / 8): the 8 lanes of an instruction that share j then spread over 8
// banks of their own instead of all 64 lanes over all 64 banks (expected deepest bank queue 2.3 instead of 3.5).
// One wave per (tile, term) segment, one lane per posting; key = (rank in class / 2, class, rank in class % 2), position =
// number of smaller keys.
__global__ void k_seg_bank_order(const uint2 *tile_seg, int64_t seg_stride, const int64_t *tile_post_base, uint32_t *post_c,
                                 int64_t tile0, int64_t n_tiles, int32_t dim) {
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const int lane = threadIdx.x % kWave;
  if (wave >= (n_tiles - tile0) * (int64_t)dim) return;
  const int64_t tile = tile0 + wave / dim;
  const int32_t term = (int32_t)(wave % dim);
  const uint2 sg = tile_seg[tile * seg_stride + term];
  if (sg.y < 3u || sg.y > (uint32_t)kWave) return;
  uint32_t *p = post_c + tile_post_base[tile] + sg.x;
  const bool live = (uint32_t)lane < sg.y;
  const uint32_t w = live ? p[lane] : 0u;
  const uint32_t cls = ((w & 0xffffu) >> 4) & 7u;  // slot field = slot * 2: bank = (field >> 2) & 63, class = bank >> 3
  uint32_t rank = 0;
#pragma unroll
  for (uint32_t c = 0; c < 8; ++c) {
    const unsigned long long m = __ballot(live && cls == c);
    if (cls == c) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  }
  const uint32_t key = live ? ((rank >> 1) << 4 | cls << 1 | (rank & 1u)) : 0xffffffffu;
  uint32_t pos = 0;
  for (int j = 0; j < kWave; ++j) pos += (uint32_t)__shfl((int)key, j) < key ? 1u : 0u;
  if (live) p[pos] = w;  // every posting was read into a register before the first is written back (one wave, in order)
}

// per-tile minimum of the positive shard sub-norms (threshold scale of a tile in shard mode)
__global__ void k_tile_min_sub(const float *sub, int64_t n_rows, int32_t cb, float *tile_min, int64_t tile0) {
  const int64_t tile = tile0 + blockIdx.x;
  const int64_t r0 = tile * cb, r1 = r0 + cb < n_rows ? r0 + cb : n_rows;
  float m = 3.0e38f;
  for (int64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
    const float v = sub[r];
    if (v > 0.f) m = fminf(m, v);
  }
  __shared__ float red[1024];
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = blockDim.x / 2; o; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fminf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) tile_min[tile] = red[0] > 1.0e38f ? 0.f : red[0];
}

// inclusive prefix sum over the 64 lanes with DPP (VALU only, no LDS round trips)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x111, 0xf, 0xf, true);  // row_shr:1, lanes without a source add 0
  x += (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x112, 0xf, 0xf, true);  // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x114, 0xf, 0xf, true);  // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x118, 0xf, 0xf, true);  // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1, 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2, 3
  return x;
}

// ---------------------------------------------------------------------------------------------------------
// probe (IWA:74-111 + CU:98-117 + IWA:93)

struct ProbeArgs {
  // index
  const uint2 *tile_seg;          // [n_tiles][seg_stride] {first posting, length} per term
  int64_t seg_stride;
  const Posting *post;
  const int64_t *tile_post_base;  // [n_tiles + 1] tile T's postings are post[base[T] .. base[T + 1])
  const int64_t *ext_id;        // candidate external ids
  const float *c_scale;         // shard sub-norm per candidate slot (null: 1)
  const float *tile_scale;      // min positive sub-norm per tile (null: 1)
  int64_t n_rows;
  int32_t cb;
  int32_t n_tiles;
  int32_t tile0;                  // first tile of this launch (a long join is cut into launches of bounded duration)
  // query batch
  const int64_t *q_rowptr;  // offsets into q_idx / q_val
  const int32_t *q_idx;
  const float *q_val;
  const int64_t *q_ext;
  const float *q_scale;     // shard sub-norm per query row (null: 1)
  int64_t q_nnz_end;        // q_idx / q_val hold at least this many elements (> 0 when a probe is launched)
  int32_t nq;
  int32_t q_chunk;          // queries per workgroup
  int32_t n_chunks;
  int32_t flat_waves;       // k_probe_even: waves that stage a round = ceil(longest query row / (64 / lanes per term))
  int32_t flat_group_log2;  // k_probe_even: log2 of the staging lanes per term (0, 1 or 2)
  int32_t merge_log2;       // k_probe_even: 2^merge_log2 neighbouring query rows share a round (apss_even.hpp); nq, q_chunk then count rounds
  int32_t nq_rows;          //   ... and this is the number of query ROWS (= nq when nothing is merged)
  int32_t tri;              // SYMMETRIC whole-store join (filter kernels): query row v is stored row v and q_chunk divides cb; a
                            // workgroup whose candidate tile lies ABOVE its queries' tile leaves at once, one BELOW counts its
                            // statistics twice, and k_mirror_survivors adds (c, q) for every survivor (q, c) of such a tile pair
  int64_t q_slot_base;      // slot of query row 0 when the batch is stored in the index, else -1
  float theta;
  float fx_scale;     // fixed-point accumulators: 1.0 is this many units (2^30 or 2^28), k_probe_wave
  uint32_t theta_fx;  // ceil(theta * fx_scale), computed in double on the host
  int32_t theta_fxi;  // the same, signed (theta <= 0 allowed): k_probe<.., FX>
  const uint32_t *post_c;  // coarse postings (k_probe_coarse)
  // virtual rows (k_probe_coarse, queries of more than 512 terms): part v covers q_idx[vrow_ptr[v] .. vrow_ptr[v+1])
  // of query vrow_q[v]; vq_first[q] = first part of query q ([nq + 1]); all null: one part per query
  const int32_t *vq_first;
  const int64_t *vrow_ptr;
  const int32_t *vrow_q;  // [nv + 1], sentinel nq at the end
  float cx_scale;          // coarse accumulator units per 1.0 (2^15 or 2^14)
  float cx_theta;          // theta * cx_scale * (1 - 2^-11 - 1e-6): the coarse threshold before the per-query slack
  // output
  int32_t *res_q;
  int32_t *res_c;
  float *res_s;
  uint64_t res_cap;
  unsigned long long *counters;  // [kCtrCount]
  unsigned long long *dbg;       // diagnostic build only: per-segment cycle sums [8]
};

// dynamic-LDS carve (all offsets multiples of 16 B)
struct ProbeLds {
  float *acc;        // [cb] fp32 accumulators of the tile's candidates
  uint2 *items;      // [kItemCap] {first posting, count | term slot << 8}
  uint2 *longs;      // [kLongCap] {first posting, length}
  float *long_w;     // [kLongCap]
  float *wq;         // [BLOCK] query weights of the current term pass
  uint32_t *surv;    // [kSurvCap] local slots that crossed the threshold
  uint32_t *bitmap;  // [cb / 32] touched bits (MODE 2 only)
  uint32_t *ctr;     // [2][4]: items, longs, survivors, pad -- double-buffered by round parity
};

__host__ __device__ inline size_t probe_lds_bytes(int cb, int block, int mode) {
  size_t b = (size_t)cb * 4 + (size_t)kItemCap * 8 + (size_t)kLongCap * 8 + (size_t)kLongCap * 4 +
             (size_t)block * 4 + (size_t)kSurvCap * 4 + 64;
  if (mode == 2) b += (size_t)((cb + 31) / 32 + 3) / 4 * 16;
  return (b + 15) / 16 * 16;
}

// MODE 0: non-negative weights and a positive threshold: a candidate's running sum is monotone, so the one add
//         that takes it across the threshold is detected from the returning LDS atomic; no accumulator scan.
// MODE 1: general weights, positive threshold: scan all accumulators after the round.
// MODE 2: threshold <= 0: additionally track touched candidates in an LDS bitmap (an untouched candidate is
//         never scored by the reference even though 0 >= theta).
// FX: accumulate in signed 32-bit fixed point (scale a.fx_scale) instead of fp32 -- LDS integer atomics are ~25x
// faster than ds_add_f32 on gfx950; the host picks FX whenever the row norms bound every partial score.
template <int MODE, int BLOCK, bool FX>
__global__ __launch_bounds__(BLOCK) void k_probe(const ProbeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  ProbeLds L;
  {
    unsigned char *p = smem_raw;
    L.acc = (float *)p;
    p += (size_t)a.cb * 4;
    L.items = (uint2 *)p;
    p += (size_t)kItemCap * 8;
    L.longs = (uint2 *)p;
    p += (size_t)kLongCap * 8;
    L.long_w = (float *)p;
    p += (size_t)kLongCap * 4;
    L.wq = (float *)p;
    p += (size_t)BLOCK * 4;
    L.surv = (uint32_t *)p;
    p += (size_t)kSurvCap * 4;
    L.ctr = (uint32_t *)p;
    p += 64;
    L.bitmap = (uint32_t *)p;
  }
  const int tid = threadIdx.x;
  const int cb = a.cb;
  const int tile = a.tile0 + blockIdx.x / a.n_chunks;
  const int chunk = blockIdx.x % a.n_chunks;
  const int q0 = chunk * a.q_chunk;
  const int q1 = min(a.nq, q0 + a.q_chunk);
  const uint2 *tp = a.tile_seg + (int64_t)tile * a.seg_stride;
  const int64_t tile_row0 = (int64_t)tile * cb;
  const Posting *post = a.post + a.tile_post_base[tile];
  const float tile_scale = a.tile_scale ? a.tile_scale[tile] : 1.0f;

  if (tid < 8) L.ctr[tid] = 0;
  unsigned long long my_visits = 0;
  unsigned long long my_cands = 0;  // first touches seen by this lane
  unsigned long long my_self = 0;
  __syncthreads();

  // software pipeline of the query-side loads (each level is issued one round before it is needed, so no
  // dependent global-load chain sits in front of a round): R = row extent, I = term + weight, P = segment
  int64_t qb1 = 0, qe1 = 0, qb2 = 0, qe2 = 0;  // extents of rounds r+1, r+2
  int32_t term1 = 0;
  float w1 = 0.f;  // round r+1
  uint32_t s0 = 0, len0 = 0;
  float w0 = 0.f;  // round r
  int64_t qb0 = 0, qe0 = 0;
  if (q0 < q1) {
    qb0 = a.q_rowptr[q0];
    qe0 = a.q_rowptr[q0 + 1];
    if (qb0 + tid < qe0) {
      const int32_t t = a.q_idx[qb0 + tid];
      w0 = a.q_val[qb0 + tid];
      const uint2 sg = tp[t];
      s0 = sg.x;
      len0 = sg.y;
    }
    if (q0 + 1 < q1) {
      qb1 = a.q_rowptr[q0 + 1];
      qe1 = a.q_rowptr[q0 + 2];
      if (qb1 + tid < qe1) {
        term1 = a.q_idx[qb1 + tid];
        w1 = a.q_val[qb1 + tid];
      }
    }
    if (q0 + 2 < q1) {
      qb2 = a.q_rowptr[q0 + 2];
      qe2 = a.q_rowptr[q0 + 3];
    }
  }

  for (int q = q0; q < q1; ++q) {
    uint32_t *ctr = L.ctr + ((q - q0) & 1) * 4;
    const int64_t qb = qb0, qe = qe0;
    const int nnz = (int)(qe - qb);
    const float qs = a.q_scale ? a.q_scale[q] : 1.0f;
    const float thr = a.theta * qs * tile_scale;
    const float fxs = FX ? a.fx_scale : 1.0f, fxinv = FX ? 1.0f / a.fx_scale : 1.0f;
    // integer threshold: exact for the plain join (computed in double on the host), rounded down in shard mode
    const int thr_i = (a.q_scale || a.tile_scale) ? (int)floorf(thr * fxs * (thr > 0 ? 0.999999f : 1.000001f)) : a.theta_fxi;
    int *acci = reinterpret_cast<int *>(L.acc);

    // ---- issue the next rounds' loads: P(r+1), I(r+2), R(r+3) ----
    uint32_t s1 = 0, len1 = 0;
    if (q + 1 < q1 && qb1 + tid < qe1) {
      const uint2 sg = tp[term1];
      s1 = sg.x;
      len1 = sg.y;
    }
    int32_t term2 = 0;
    float w2 = 0.f;
    if (q + 2 < q1 && qb2 + tid < qe2) {
      term2 = a.q_idx[qb2 + tid];
      w2 = a.q_val[qb2 + tid];
    }
    int64_t qb3 = 0, qe3 = 0;
    if (q + 3 < q1) {
      qb3 = a.q_rowptr[q + 3];
      qe3 = a.q_rowptr[q + 4];
    }

    // ---- zero the tile's accumulators ----
    {
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int i = tid * 4; i < cb; i += BLOCK * 4) *reinterpret_cast<float4 *>(L.acc + i) = z;
      if (MODE == 2)
        for (int i = tid; i < (cb + 31) / 32; i += BLOCK) L.bitmap[i] = 0;
    }

    // ---- term passes (one pass unless the query has more than BLOCK terms) ----
    for (int t0 = 0; t0 < nnz; t0 += BLOCK) {
      uint32_t s = 0, len = 0;
      float w = 0.f;
      if (t0 == 0) {
        s = s0;
        len = len0;
        w = w0;
      } else {
        __syncthreads();  // previous pass drained, its counter reset visible
        if (t0 + tid < nnz) {
          const int32_t t = a.q_idx[qb + t0 + tid];
          w = a.q_val[qb + t0 + tid];
          const uint2 sg = tp[t];
          s = sg.x;
          len = sg.y;
        }
      }
      my_visits += len;
      L.wq[tid] = w;
      bool pending = len > 0;
      while (true) {
        if (pending) {
          if (len <= (uint32_t)kLongLen) {
            const uint32_t nch = (len + kGroup - 1) / kGroup;
            const uint32_t base = atomicAdd(&ctr[0], nch);
            if (base + nch <= (uint32_t)kItemCap) {
              for (uint32_t k = 0; k < nch; ++k) {
                const uint32_t c = min((uint32_t)kGroup, len - k * kGroup);
                L.items[base + k] = make_uint2(s + k * kGroup, c | ((uint32_t)tid << 8));
              }
              pending = false;
            } else {
              for (uint32_t k = base; k < min(base + nch, (uint32_t)kItemCap); ++k) L.items[k] = make_uint2(0u, 0u);
            }
          } else {
            const uint32_t base = atomicAdd(&ctr[1], 1u);
            if (base < (uint32_t)kLongCap) {
              L.longs[base] = make_uint2(s, len);
              L.long_w[base] = w;
              pending = false;
            }
          }
        }
        const int any_pending = __syncthreads_or(pending ? 1 : 0);

        // ---- accumulate: acc[c] += w_q * w_c for every posting of every listed segment ----
        const uint32_t n_items = min(ctr[0], (uint32_t)kItemCap);
        const uint32_t n_long = min(ctr[1], (uint32_t)kLongCap);
        auto visit = [&](const Posting pc, const float wq_) {
          if (FX) {
            const int p = __float2int_rn(wq_ * pc.w * fxs);
            if (MODE == 0) {
              const int old = atomicAdd(&acci[pc.slot], p);  // ds_add_rtn_u32
              my_cands += (old == 0) ? 1u : 0u;
              if (old < thr_i && old + p >= thr_i) {
                const uint32_t k = atomicAdd(&ctr[2], 1u);
                if (k < (uint32_t)kSurvCap) L.surv[k] = pc.slot;
              }
            } else {
              atomicAdd(&acci[pc.slot], p);
              if (MODE == 2) atomicOr(&L.bitmap[pc.slot >> 5], 1u << (pc.slot & 31));
            }
          } else {
            const float p = wq_ * pc.w;
            if (MODE == 0) {
              const float old = atomicAdd(&L.acc[pc.slot], p);  // ds_add_rtn_f32
              my_cands += (old == 0.0f) ? 1u : 0u;
              if (old < thr && old + p >= thr) {
                const uint32_t k = atomicAdd(&ctr[2], 1u);
                if (k < (uint32_t)kSurvCap) L.surv[k] = pc.slot;
              }
            } else {
              atomicAdd(&L.acc[pc.slot], p);
              if (MODE == 2) atomicOr(&L.bitmap[pc.slot >> 5], 1u << (pc.slot & 31));
            }
          }
        };
        // Four work items / four long segments at a time: their postings are requested together and added afterwards.  One at a
        // time every trip was a dependent chain (LDS descriptor -> global load -> LDS atomics), ~1 us each with nothing else in
        // flight: 25 + 40 such trips per round on the template shape at theta = 0, where the SQ counters showed 73 % of the
        // wave-cycles waiting (round 4)
        constexpr int UN = 4;
        const int grp = tid / kGroup, gl = tid % kGroup;
        for (uint32_t i0 = grp; i0 < n_items; i0 += UN * (BLOCK / kGroup)) {
          Posting pv[UN];
          float wv[UN];
          bool on[UN];
#pragma unroll
          for (int u = 0; u < UN; ++u) {
            const uint32_t i = i0 + u * (BLOCK / kGroup);
            on[u] = false;
            if (i < n_items) {
              const uint2 it = L.items[i];
              on[u] = (uint32_t)gl < (it.y & 0xffu);
              if (on[u]) {
                pv[u] = post[it.x + gl];
                wv[u] = L.wq[it.y >> 8];
              }
            }
          }
#pragma unroll
          for (int u = 0; u < UN; ++u)
            if (on[u]) visit(pv[u], wv[u]);
        }
        for (uint32_t j0 = 0; j0 < n_long; j0 += UN) {
          Posting pv[UN];
          float wv[UN];
          uint2 sgv[UN];
          bool on[UN];
#pragma unroll
          for (int u = 0; u < UN; ++u) {  // the first BLOCK postings of UN segments
            on[u] = false;
            sgv[u] = make_uint2(0u, 0u);
            if (j0 + u < n_long) {
              sgv[u] = L.longs[j0 + u];
              wv[u] = L.long_w[j0 + u];
              on[u] = (uint32_t)tid < sgv[u].y;
              if (on[u]) pv[u] = post[sgv[u].x + tid];
            }
          }
#pragma unroll
          for (int u = 0; u < UN; ++u)
            if (on[u]) visit(pv[u], wv[u]);
#pragma unroll
          for (int u = 0; u < UN; ++u)  // what a segment holds beyond BLOCK postings
            for (uint32_t k = tid + BLOCK; k < sgv[u].y; k += BLOCK) visit(post[sgv[u].x + k], wv[u]);
        }
        __syncthreads();
        if (tid == 0) {
          ctr[0] = 0;
          ctr[1] = 0;
        }
        if (!any_pending) break;
        __syncthreads();
      }
    }
    if (nnz == 0) __syncthreads();  // keep the barrier count per round uniform for the zeroing hazard

    // ---- threshold prune + compaction (IWA:93-95) ----
    const int64_t qext = a.q_ext[q];
    if (MODE == 0) {
      if (a.q_slot_base >= 0) {  // the query's own slot is a touched candidate that is not a (q, c != q) pair
        const int64_t sl = a.q_slot_base + q - tile_row0;
        // checked by the thread that zeroes this accumulator next round (program order, no extra barrier)
        if (sl >= 0 && sl < cb && tid == (int)((sl >> 2) % BLOCK) && acci[sl] != 0) my_self++;
      }
      const uint32_t n_surv = ctr[2];
      if (n_surv > 0) {
        if (n_surv <= (uint32_t)kSurvCap) {
          for (uint32_t i = tid; i < (n_surv + kWave - 1) / kWave * kWave; i += BLOCK) {
            bool ok = false;
            uint32_t c = 0;
            float sc = 0.f;
            if (i < n_surv) {
              c = L.surv[i];
              sc = FX ? (float)acci[c] * fxinv : L.acc[c];
              const int64_t gs = tile_row0 + c;
              ok = a.ext_id[gs] != qext && (!a.c_scale || sc >= a.theta * qs * a.c_scale[gs]);
            }
            const uint64_t o = wave_append(ok, &a.counters[kCtrResults]);
            if (ok && o < a.res_cap) {
              a.res_q[o] = q;
              a.res_c[o] = (int32_t)(tile_row0 + c);
              a.res_s[o] = sc;
            }
          }
        } else {
          // more crossings than the LDS list holds: fall back to the accumulator scan for this round
          for (int i = tid; i < (cb + BLOCK - 1) / BLOCK * BLOCK; i += BLOCK) {
            bool ok = false;
            float sc = 0.f;
            if (i < cb) {
              sc = FX ? (float)acci[i] * fxinv : L.acc[i];
              const int64_t gs = tile_row0 + i;
              ok = (FX ? acci[i] >= thr_i : sc >= thr) && gs < a.n_rows && a.ext_id[gs] != qext &&
                   (!a.c_scale || sc >= a.theta * qs * a.c_scale[gs]);
            }
            const uint64_t o = wave_append(ok, &a.counters[kCtrResults]);
            if (ok && o < a.res_cap) {
              a.res_q[o] = q;
              a.res_c[o] = (int32_t)(tile_row0 + i);
              a.res_s[o] = sc;
            }
          }
        }
        __syncthreads();  // survivors read before the next round zeroes acc / reuses the list
        if (tid == 0) ctr[2] = 0;
      }
    } else {
      // the accumulator scan: candidate i of the tile is a result iff it was touched, is not the query itself and reaches theta
      auto judge = [&](const int i, float &sc, bool &cand) -> bool {
        cand = false;
        sc = 0.f;
        if (i >= cb) return false;
        sc = FX ? (float)acci[i] * fxinv : L.acc[i];
        const int64_t gs = tile_row0 + i;
        const bool touched = (MODE == 2) ? ((L.bitmap[i >> 5] >> (i & 31)) & 1u) : (acci[i] != 0);
        if (!touched || gs >= a.n_rows) return false;
        if (a.ext_id[gs] == qext) return false;
        cand = true;
        return (FX ? acci[i] >= thr_i : sc >= thr) && (!a.c_scale || sc >= a.theta * qs * a.c_scale[gs] * 0.999999f);
      };
      if (MODE == 2) {
        // theta <= 0 (the reference's server template ships similarityThreshold = 0): nearly every touched candidate is a
        // result, so the round's OUTPUT is the work.  One reservation of the global list per workgroup and round -- a ballot
        // count per (scan step, wave), a scan of those counts by one wave, ONE global atomic -- then every wave writes its
        // results as a contiguous run.  (One atomic per wave and step on the one list counter, as the sparse-output modes do
        // it, is serialised at the memory side: 4.5e7 of them were nearly all of the 676 ms of a 60k x 60k join at dim 1024.)
        constexpr int NWV = BLOCK / kWave;
        const int n_steps = (cb + BLOCK - 1) / BLOCK;  // <= 32: cb <= 32768
        uint32_t *const cnt = L.surv;                  // [n_steps][NWV] (the crossing list is not used in this mode)
        unsigned long long *const masks = reinterpret_cast<unsigned long long *>(L.items);  // [n_steps][NWV] (nor is the work list, by now)
        const int wvi = tid / kWave;
        for (int k = 0; k < n_steps; ++k) {
          float sc;
          bool cand;
          const bool ok = judge(k * BLOCK + tid, sc, cand);
          my_cands += cand ? 1u : 0u;
          const unsigned long long m = __ballot(ok);
          if ((tid % kWave) == 0) {
            cnt[k * NWV + wvi] = (uint32_t)__popcll(m);
            masks[k * NWV + wvi] = m;  // (the writing pass does not judge again: no second read of the external ids)
          }
        }
        __syncthreads();
        if (tid < kWave) {  // n_steps * NWV <= 512 counts: eight per lane
          const int n_cnt = n_steps * NWV;
          uint32_t v[8], run = 0;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            v[j] = tid * 8 + j < n_cnt ? cnt[tid * 8 + j] : 0u;
            run += v[j];
          }
          const uint32_t incl = wave_incl_scan(run);
          const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);
          unsigned long long base = 0;
          if (tid == 0 && total) base = atomicAdd(&a.counters[kCtrResults], (unsigned long long)total);
          base = __shfl(base, 0);
          uint32_t off = incl - run;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (tid * 8 + j < n_cnt) cnt[tid * 8 + j] = off;
            off += v[j];
          }
          if (tid == 0) {
            L.surv[kSurvCap - 2] = (uint32_t)base;
            L.surv[kSurvCap - 1] = (uint32_t)(base >> 32);
          }
        }
        __syncthreads();
        const unsigned long long base = (unsigned long long)L.surv[kSurvCap - 2] | ((unsigned long long)L.surv[kSurvCap - 1] << 32);
        for (int k = 0; k < n_steps; ++k) {
          const int i = k * BLOCK + tid;
          const unsigned long long m = masks[k * NWV + wvi];
          const bool ok = (m >> (tid % kWave)) & 1ull;
          const uint64_t o = base + cnt[k * NWV + wvi] + (uint64_t)__popcll(m & ((1ull << (tid % kWave)) - 1ull));
          if (ok && o < a.res_cap) {
            a.res_q[o] = q;
            a.res_c[o] = (int32_t)(tile_row0 + i);
            a.res_s[o] = FX ? (float)acci[i] * fxinv : L.acc[i];
          }
        }
      } else {
      for (int i = tid; i < (cb + BLOCK - 1) / BLOCK * BLOCK; i += BLOCK) {
        float sc;
        bool cand;
        const bool ok = judge(i, sc, cand);
        my_cands += cand ? 1u : 0u;
        const uint64_t o = wave_append(ok, &a.counters[kCtrResults]);
        if (ok && o < a.res_cap) {
          a.res_q[o] = q;
          a.res_c[o] = (int32_t)(tile_row0 + i);
          a.res_s[o] = sc;
        }
      }
      }
      __syncthreads();
    }

    // ---- rotate the software pipeline ----
    qb0 = qb1;
    qe0 = qe1;
    s0 = s1;
    len0 = len1;
    w0 = w1;
    qb1 = qb2;
    qe1 = qe2;
    term1 = term2;
    w1 = w2;
    qb2 = qb3;
    qe2 = qe3;
  }

  // ---- statistics: one atomic per workgroup (scratch lives in the dynamic carve: a static __shared__ object
  // would shift the 16-B alignment of the accumulator array) ----
  unsigned long long *stat = reinterpret_cast<unsigned long long *>(L.ctr + 8);
  __syncthreads();
  if (tid < 3) stat[tid] = 0;
  __syncthreads();
  atomicAdd(&stat[0], my_visits);
  atomicAdd(&stat[1], my_cands);
  if (my_self) atomicAdd(&stat[2], my_self);
  __syncthreads();
  if (tid == 0) {
    atomicAdd(&a.counters[kCtrVisits], stat[0]);
    atomicAdd(&a.counters[kCtrCands], stat[1] - stat[2]);
  }
}

// ---------------------------------------------------------------------------------------------------------
// k_probe_wave: the speed path of the probe for the common regime (non-negative weights, theta > 0, at most
// BLOCK terms per query, scores bounded so that u32 fixed point holds them).  Differences from k_probe:
//   * u32 fixed-point accumulators: LDS float atomics are serialised on gfx950 (ds_add_f32: ~193 cycles per
//     wave-instruction = 0.33 lanes/clk/CU) while integer LDS atomics run at ~8 lanes/clk/CU with random
//     addresses (profiles/microbench/lds_atomics.hip).  Every product is rounded once to a multiple of
//     1/fx_scale and sums are exact: |error| <= terms * 2^-31 at scale 2^30 (better than an fp32 running sum);
//   * no workgroup-wide work list: term k of the query belongs to wave k % NW, lane k / NW.  A wave cuts its own
//     segments into 8-posting chunks (64 B), numbers them with a DPP prefix sum and keeps the chunk descriptors
//     in a wave-private LDS strip, so no workgroup barrier sits between finding segments and loading them;
//   * deep prefetch: the postings of round r+2 are requested at the top of round r (segment look-ups for r+3,
//     term ids for r+4, row extents for r+5), every load unconditional so the waits can be counted;
//   * accumulators are re-zeroed by writing 0 to exactly the slots the round touched (their postings are
//     still in registers): LDS traffic proportional to posting visits, not to the tile size.  Rounds that
//     overflow the register window or sweep a long segment fall back to clearing the tile.
constexpr int kChunk = 8;        // postings per chunk: 8 lanes x 8 B = 64 B
constexpr int kLongLenW = 256;   // longer segments are swept by the whole workgroup

__host__ __device__ inline size_t probe_wave_lds_bytes(int cb, int block, int u, int longcap, int survcap) {
  return ((size_t)(cb + kWave) * 4 + (size_t)(block / kWave) * (kWave / kChunk) * u * 8 + 3 * (size_t)longcap * 12 +
          (size_t)survcap * 4 + 128 + 15) / 16 * 16;
}

__device__ __forceinline__ int64_t uniform64(int64_t x) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)x >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}


// LONGCAP: long segments per round kept in the workgroup list (more stay with their wave); SURVCAP: threshold
// crossings per round kept in LDS (more: the round falls back to scanning the accumulators)
typedef unsigned int apss_u32x2 __attribute__((ext_vector_type(2)));

template <int BLOCK, int U, int LONGCAP, int SURVCAP, bool SHARD, bool DIAG = false>
__global__ __launch_bounds__(BLOCK, BLOCK <= 512 && U <= 5 ? 2 * BLOCK / 256 : BLOCK / 256) void k_probe_wave(const ProbeArgs a) {
  constexpr int NW = BLOCK / kWave;
  constexpr int GPW = kWave / kChunk;  // chunk groups per wave step (8)
  constexpr int WIN = GPW * U;         // chunks in one wave's register window
  static_assert(WIN <= kWave, "the window strip is cleared by one store per lane");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint32_t *acc = (uint32_t *)smem_raw;                  // [cb] accumulators (+ 64 words of slack)
  uint2 *items = (uint2 *)(acc + a.cb + kWave);          // [NW][WIN] {first posting * 8 + (count - 1), weight bits}
  uint2 *longs = (uint2 *)(items + NW * WIN);            // [3][LONGCAP]
  float *long_w = (float *)(longs + 3 * LONGCAP);        // [3][LONGCAP]
  uint32_t *surv = (uint32_t *)(long_w + 3 * LONGCAP);   // [SURVCAP]
  // ctr[0..2] long segments of round (r % 3); ctr[4 + 2p] "clear whole tile", ctr[5 + 2p] survivors (p = r & 1)
  uint32_t *ctr = surv + SURVCAP;
  uint32_t n_long_next = 0;  // long segments of the coming round, read behind the previous round's barrier
  unsigned long long *stat = reinterpret_cast<unsigned long long *>(ctr + 8);

  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid / kWave), ln = tid % kWave;
  const int cb = a.cb;
  const int tile = a.tile0 + blockIdx.x / a.n_chunks;
  const int chunk = blockIdx.x % a.n_chunks;
  const int q0 = chunk * a.q_chunk;
  const int q1 = min(a.nq, q0 + a.q_chunk);
  const int64_t tile_row0 = (int64_t)tile * cb;
  const float tile_scale = SHARD ? a.tile_scale[tile] : 1.0f;
  const int kterm = ln * NW + wv;               // the term of the query this lane looks after
  const uint32_t lo = (uint32_t)(ln % kChunk);  // posting of the chunk this lane handles
  uint2 *wl = items + wv * WIN;                 // this wave's chunk strip
  const float fxs = a.fx_scale, fxinv = 1.0f / a.fx_scale;

  // buffer descriptors built from wave-uniform values only (32-bit offsets, hardware range check: a read past
  // the end returns 0, which is what makes every load below unconditional and every address clamp-free)
  const int64_t qbase = a.q_rowptr[q0], qend = a.q_rowptr[q1];
  const int64_t pbase = a.tile_post_base[tile], pend = a.tile_post_base[tile + 1];
  const __amdgpu_buffer_rsrc_t rs_qi =
      __builtin_amdgcn_make_buffer_rsrc((void *)(a.q_idx + qbase), 0, (int)((qend - qbase) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_qv =
      __builtin_amdgcn_make_buffer_rsrc((void *)(a.q_val + qbase), 0, (int)((qend - qbase) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_tp = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(a.tile_seg + (int64_t)tile * a.seg_stride), 0, (int)(a.seg_stride * 8), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_po =
      __builtin_amdgcn_make_buffer_rsrc((void *)(a.post + pbase), 0, (int)((pend - pbase) * 8), 0x00020000);
  constexpr uint32_t kOob = 0xfffffff0u;  // an offset no descriptor covers

  for (int i = tid * 4; i < cb + kWave; i += BLOCK * 4) *reinterpret_cast<uint4 *>(acc + i) = make_uint4(0u, 0u, 0u, 0u);
  if (tid < 16) ctr[tid] = 0;
  unsigned long long my_visits = 0;
  uint32_t my_cands = 0;  // first touches seen by this lane (< 2^32 per workgroup chunk)

  // ---- pipeline stages ----
  struct RowExt { int qb; int nnz; };  // qb relative to qbase
  struct TermW { uint32_t term; float w; bool valid; };
  struct Seg { uint32_t s, len; float w; };
  struct WaveWork {
    uint32_t s, len, excl;  // per lane: its term's segment and the number of chunks of the lanes before it
    float w;                // per lane: query weight of its term
    int totch;              // wave-uniform: chunks of this wave this round
    int tw;                 // wave-uniform: terms of this wave this round
    apss_u32x2 pc[U];       // prefetched postings {slot, weight bits}: step u, chunk u * GPW + lane / 8, posting lane % 8
    float wq[U];            // query weight of that chunk's term times fx_scale; 0 in lanes past the chunk's end
  };
  auto load_R = [&](int q) {
    RowExt r;
    const int qq = min(q, a.nq - 1);
    const int64_t b = a.q_rowptr[qq], e = a.q_rowptr[qq + 1];
    r.qb = (int)(b - qbase);
    r.nnz = q < q1 ? (int)(e - b) : 0;
    return r;
  };
  auto load_I = [&](const RowExt &r) {
    TermW t;
    t.valid = kterm < r.nnz;
    const uint32_t off = t.valid ? (uint32_t)(r.qb + kterm) * 4u : kOob;
    t.term = __builtin_amdgcn_raw_buffer_load_b32(rs_qi, off, 0, 0);
    t.w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_qv, off, 0, 0));
    return t;
  };
  auto load_P = [&](const TermW &t) {
    Seg g;
    const apss_u32x2 sg = __builtin_amdgcn_raw_buffer_load_b64(rs_tp, t.term * 8u, 0, 0);  // {first posting, length}
    g.s = sg.x;
    g.len = t.valid ? sg.y : 0u;
    g.w = t.w;
    return g;
  };
  // Cut the wave's segments into 8-posting chunks, number them with a wave prefix sum, keep the descriptors of the
  // first WIN chunks in the wave's LDS strip, and request the postings of those chunks (8 lanes per chunk).
  auto flatten = [&](WaveWork &f, const Seg &g, const RowExt &r, int li) {
    uint32_t len = g.len;
    my_visits += len;
    if (len > (uint32_t)kLongLenW) {  // swept by the whole workgroup in its own round
      const uint32_t k = atomicAdd(&ctr[li], 1u);
      if (k < (uint32_t)LONGCAP) {
        longs[li * LONGCAP + k] = make_uint2(g.s, len);
        long_w[li * LONGCAP + k] = g.w;
        len = 0;
      }  // else: the list is full, the segment stays with this wave (slow path)
    }
    const uint32_t nch = (len + kChunk - 1) / kChunk;
    const uint32_t incl = wave_incl_scan(nch);
    const uint32_t excl = incl - nch;
    f.s = g.s;
    f.len = len;
    f.excl = excl;
    f.w = g.w;
    f.tw = wv < r.nnz ? (r.nnz - wv + NW - 1) / NW : 0;
    f.totch = __builtin_amdgcn_readlane((int)incl, kWave - 1);
    // clear the strip (an all-zero descriptor has weight 0: it adds 0 to the slot of posting 0), then every lane
    // stores the chunks of its own segment; LDS operations of one wave execute in order, so the reads see them
    if (ln < WIN) wl[ln] = make_uint2(0u, 0u);
    const uint32_t wbits = __float_as_uint(fxs * g.w);
    auto put = [&](const uint32_t k) {
      if (k < nch && excl + k < (uint32_t)WIN)
        wl[excl + k] = make_uint2((g.s + k * kChunk) * 8u + (min((uint32_t)kChunk, len - k * kChunk) - 1u), wbits);
    };
    put(0);
    put(1);
    put(2);
    if (__any(nch > 3u))  // rare: a segment of more than 24 postings in a 16384-row tile
      for (uint32_t k = 3; __any(k < nch && excl + k < (uint32_t)WIN); ++k) put(k);
    uint2 it[U];
#pragma unroll
    for (int u = 0; u < U; ++u) it[u] = wl[u * GPW + ln / kChunk];  // all reads in flight before the first use
#pragma unroll
    for (int u = 0; u < U; ++u) asm volatile("" : "+v"(it[u].x), "+v"(it[u].y));  // whole ds_read_b64s, no branch around half
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t c1 = it[u].x & 7u;                               // postings in the chunk - 1
      f.wq[u] = lo <= c1 ? __uint_as_float(it[u].y) : 0.0f;          // 0 marks the lanes past the chunk's end
      f.pc[u] = __builtin_amdgcn_raw_buffer_load_b64(rs_po, (it[u].x ^ c1) + lo * 8u, 0, 0);
    }
  };

  RowExt R1 = load_R(q0 + 1), R2 = load_R(q0 + 2), R3 = load_R(q0 + 3), R4 = load_R(q0 + 4);
  TermW I3, I4;
  Seg P2, P3;
  WaveWork wfa, wfb, wfc;
  {
    const RowExt R0 = load_R(q0);
    const TermW I0 = load_I(R0), I1 = load_I(R1), I2 = load_I(R2);
    I3 = load_I(R3);
    const Seg P0 = load_P(I0), P1 = load_P(I1);
    P2 = load_P(I2);
    __syncthreads();  // counters zeroed before the first long-segment pushes
    flatten(wfa, P0, R0, 0);
    flatten(wfb, P1, R1, 1);
  }
  __syncthreads();  // accumulators cleared, long lists of rounds 0 and 1 complete
  n_long_next = ctr[0];

  unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define APSS_STAMP(k)                                   \
  if (DIAG) {                                           \
    const unsigned long long t_ = clock64();            \
    tsum[k] += t_ - tprev;                              \
    tprev = t_;                                         \
  }
  // One round: query q against this tile.  w0 holds round q's prefetched work, w2 receives round q+2's; l3 is
  // (q - q0) % 3, a compile-time constant because the loop below is unrolled three times with the three
  // WaveWork register sets rotating by name (no register copies).
  auto round = [&](WaveWork &w0, WaveWork &w2, const int q, const int l3) {
    unsigned long long tprev = DIAG ? clock64() : 0;
    const int par = (q - q0) & 1;
    const float qs = SHARD ? a.q_scale[q] : 1.0f;
    const float thr = a.theta * qs * tile_scale;
    // shard mode: the candidate test is a necessary condition that is verified exactly later, so round it down
    const uint32_t thr_fx = SHARD ? max(1u, (uint32_t)(thr * fxs * 0.999999f)) : a.theta_fx;
    const uint32_t thr1 = thr_fx - 1u;  // theta > 0 => thr_fx >= 1

    // ---- stage loads for the rounds ahead ----
    const RowExt R5 = load_R(q + 5);
    I4 = load_I(R4);
    P3 = load_P(I3);
    flatten(w2, P2, R2, l3 == 0 ? 2 : l3 - 1);
    APSS_STAMP(0)

    // ---- accumulate round q ----
    // old < thr <= old + p  <=>  (thr - 1 - old) < p in unsigned arithmetic: this add crossed the threshold
    auto crossed = [&](const uint32_t slot, const uint32_t p, const uint32_t old) {
      if (thr1 - old < p) {
        const uint32_t k = atomicAdd(&ctr[5 + 2 * par], 1u);
        if (k < (uint32_t)SURVCAP) surv[k] = slot;
      }
    };
    auto visit = [&](const apss_u32x2 pc, const float wqs) {
      const uint32_t p = (uint32_t)__builtin_fmaf(wqs, __uint_as_float(pc.y), 0.5f);
      const uint32_t old = atomicAdd(&acc[pc.x], p);
      my_cands += old == 0u ? 1u : 0u;
      crossed(pc.x, p, old);
    };
    {
      // the register window: LDS atomics of every real posting issued back to back, then the threshold checks.
      // Lanes past a chunk's end (weight 0) stay out of the atomics: fewer LDS bank conflicts, exact counts.
      uint32_t p[U], old[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        p[u] = (uint32_t)__builtin_fmaf(w0.wq[u], __uint_as_float(w0.pc[u].y), 0.5f);
        old[u] = 1u;
        if (w0.wq[u] != 0.0f) old[u] = atomicAdd(&acc[w0.pc[u].x], p[u]);  // ds_add_rtn_u32
      }
      bool any_cross = false;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        my_cands += old[u] == 0u ? 1u : 0u;  // first touch of this candidate in this round
        any_cross |= thr1 - old[u] < p[u];    // idle lanes: old = 1, p = 0: never true
      }
      if (any_cross) {  // rare: some add of this lane took a candidate across the threshold
#pragma unroll
        for (int u = 0; u < U; ++u) crossed(w0.pc[u].x, p[u], old[u]);
      }
    }
    APSS_STAMP(1)
    if (w0.totch > WIN) {  // wave-uniform: more chunks than the register window holds
      if (ln == 0) ctr[4 + 2 * par] = 1;
      for (int c0 = WIN; c0 < w0.totch; c0 += GPW) {
        const uint32_t c = (uint32_t)(c0 + ln / kChunk);
        uint32_t st = 0, cn = 0;
        float wq_ = 0.f;
        for (int m = 0; m < w0.tw; ++m) {
          const uint32_t em = (uint32_t)__builtin_amdgcn_readlane((int)w0.excl, m);
          const uint32_t lm = (uint32_t)__builtin_amdgcn_readlane((int)w0.len, m);
          const uint32_t sm = (uint32_t)__builtin_amdgcn_readlane((int)w0.s, m);
          const float wm = fxs * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w0.w), m));
          const uint32_t nm = (lm + kChunk - 1) / kChunk;
          const bool sel = c >= em && c < em + nm;
          const uint32_t k = c - em;
          st = sel ? sm + k * kChunk : st;
          cn = sel ? min((uint32_t)kChunk, lm - k * kChunk) : cn;
          wq_ = sel ? wm : wq_;
        }
        if (lo < cn) visit(__builtin_amdgcn_raw_buffer_load_b64(rs_po, (st + lo) * 8u, 0, 0), wq_);
      }
    }
    const uint32_t n_long = min(n_long_next, (uint32_t)LONGCAP);
    for (uint32_t j = 0; j < n_long; ++j) {
      const uint2 sgm = longs[l3 * LONGCAP + j];
      const float wq_ = fxs * long_w[l3 * LONGCAP + j];
      uint32_t k = tid;
      for (; k + 3 * BLOCK < sgm.y; k += 4 * BLOCK) {  // four loads in flight per lane
        const apss_u32x2 p0 = __builtin_amdgcn_raw_buffer_load_b64(rs_po, (sgm.x + k) * 8u, 0, 0),
                         p1 = __builtin_amdgcn_raw_buffer_load_b64(rs_po, (sgm.x + k + BLOCK) * 8u, 0, 0),
                         p2 = __builtin_amdgcn_raw_buffer_load_b64(rs_po, (sgm.x + k + 2 * BLOCK) * 8u, 0, 0),
                         p3 = __builtin_amdgcn_raw_buffer_load_b64(rs_po, (sgm.x + k + 3 * BLOCK) * 8u, 0, 0);
        visit(p0, wq_);
        visit(p1, wq_);
        visit(p2, wq_);
        visit(p3, wq_);
      }
      for (; k < sgm.y; k += BLOCK) visit(__builtin_amdgcn_raw_buffer_load_b64(rs_po, (sgm.x + k) * 8u, 0, 0), wq_);
    }
    APSS_STAMP(2)
    __syncthreads();  // every add of round q has landed
    APSS_STAMP(3)

    // ---- threshold prune + compaction (IWA:93-95) ----
    const uint2 fl = *reinterpret_cast<const uint2 *>(&ctr[4 + 2 * par]);  // {clear-whole-tile flag, survivors}
    n_long_next = ctr[l3 == 2 ? 0 : l3 + 1];  // complete since round q-1's pushes; same LDS wait as the flags
    const uint32_t n_surv = fl.y;
    const bool full_zero = n_long > 0 || fl.x != 0 || n_surv > (uint32_t)SURVCAP;
    if (n_surv > 0) {
      const int64_t qext = a.q_ext[q];
      if (n_surv <= (uint32_t)SURVCAP) {
        for (uint32_t i = tid; i < (n_surv + kWave - 1) / kWave * kWave; i += BLOCK) {
          bool ok = false;
          uint32_t c = 0;
          float sc = 0.f;
          if (i < n_surv) {
            c = surv[i];
            sc = (float)acc[c] * fxinv;
            const int64_t gs = tile_row0 + c;
            ok = a.ext_id[gs] != qext && (!SHARD || sc >= a.theta * qs * a.c_scale[gs] * 0.999999f);
          }
          const uint64_t o = wave_append(ok, &a.counters[kCtrResults]);
          if (ok && o < a.res_cap) {
            a.res_q[o] = q;
            a.res_c[o] = (int32_t)(tile_row0 + c);
            a.res_s[o] = sc;
          }
        }
      } else {
        for (int i = tid; i < (cb + BLOCK - 1) / BLOCK * BLOCK; i += BLOCK) {
          bool ok = false;
          float sc = 0.f;
          if (i < cb) {
            const uint32_t raw = acc[i];
            sc = (float)raw * fxinv;
            const int64_t gs = tile_row0 + i;
            ok = raw >= thr_fx && gs < a.n_rows && a.ext_id[gs] != qext &&
                 (!SHARD || sc >= a.theta * qs * a.c_scale[gs] * 0.999999f);
          }
          const uint64_t o = wave_append(ok, &a.counters[kCtrResults]);
          if (ok && o < a.res_cap) {
            a.res_q[o] = q;
            a.res_c[o] = (int32_t)(tile_row0 + i);
            a.res_s[o] = sc;
          }
        }
      }
      __syncthreads();  // final scores read before they are cleared
    }
    APSS_STAMP(4)

    // ---- re-zero exactly the slots the round touched ----
    if (full_zero) {
      for (int i = tid * 4; i < cb; i += BLOCK * 4) *reinterpret_cast<uint4 *>(acc + i) = make_uint4(0u, 0u, 0u, 0u);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (w0.wq[u] != 0.0f) acc[w0.pc[u].x] = 0u;
    }
    APSS_STAMP(5)
    if (tid == 0) {
      // this round's long list was consumed before the barrier above and is refilled (for round q+3) after the
      // barrier below; the OTHER parity's flags were last read before the closing barrier of round q-1 and are
      // next written after the closing barrier of this round: both can be cleared here without a race
      ctr[l3] = 0;
      ctr[4 + 2 * (par ^ 1)] = 0;
      ctr[5 + 2 * (par ^ 1)] = 0;
    }
    __syncthreads();  // clears done before any add of round q+1
    APSS_STAMP(6)

    P2 = P3;
    I3 = I4;
    R1 = R2;
    R2 = R3;
    R3 = R4;
    R4 = R5;
  };
  for (int q = q0; q < q1; q += 3) {
    round(wfa, wfc, q, 0);
    if (q + 1 >= q1) break;
    round(wfb, wfa, q + 1, 1);
    if (q + 2 >= q1) break;
    round(wfc, wfb, q + 2, 2);
  }
#undef APSS_STAMP
  if (DIAG && ln == 0 && a.dbg)
    for (int k = 0; k < 8; ++k) atomicAdd(&a.dbg[k], tsum[k]);
  __syncthreads();
  if (tid < 3) stat[tid] = 0;
  __syncthreads();
  atomicAdd(&stat[0], my_visits);
  atomicAdd(&stat[1], (unsigned long long)my_cands);
  __syncthreads();
  if (tid == 0) {
    atomicAdd(&a.counters[kCtrVisits], stat[0]);
    atomicAdd(&a.counters[kCtrCands], stat[1]);
  }
}

// virtual-row table of a query batch: row r has max(1, ceil(nnz / part)) parts
__global__ void k_vrow_count(const int64_t *rowptr, int64_t n, int part, int64_t *nparts) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) {
    const int64_t len = rowptr[r + 1] - rowptr[r];
    nparts[r] = len <= part ? 1 : (len + part - 1) / part;
  }
}
__global__ void k_vrow_fill(const int64_t *rowptr, int64_t n, int part, const int64_t *first, int32_t *vq_first,
                            int64_t *vrow_ptr, int32_t *vrow_q) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n) return;
  if (r == n) {  // sentinels
    vq_first[n] = (int32_t)first[n];
    vrow_ptr[first[n]] = rowptr[n];
    vrow_q[first[n]] = (int32_t)n;
    return;
  }
  const int64_t b = rowptr[r], e = rowptr[r + 1], f = first[r], np = first[r + 1] - f;
  vq_first[r] = (int32_t)f;
  for (int64_t p = 0; p < np; ++p) {
    vrow_ptr[f + p] = b + p * part < e ? b + p * part : e;
    vrow_q[f + p] = (int32_t)r;
  }
}

// ---------------------------------------------------------------------------------------------------------
// k_probe_coarse: FILTER pass of the two-pass exact join.  Same structure as k_probe_wave, but it reads the coarse
// index (4-B postings {u16 slot, fp16 weight}, 32 per 128-B line) and sums into 16-bit LDS accumulators (two
// candidates per 32-bit word), so a tile holds 32768 candidates in 64 KB: half the (query, tile) rounds of the exact
// kernel at two workgroups per CU, and about half the memory-side bytes per posting visit.
// It reports every pair whose coarse sum reaches  floor(theta*S*(1 - 2^-11 - 1e-6)) - 2  (S = cx_scale).
// No true pair is lost.  A stored weight c_i becomes fp16: round-to-nearest errs by at most 2^-11 RELATIVE while the
// value is a normal fp16 (>= 2^-14) and by at most 2^-25 ABSOLUTE below that (subnormals; a weight under fp16's range is
// clamped UP to the smallest subnormal by pack_coarse), so  fp16(c_i) >= c_i (1 - 2^-11) - 2^-25.  Each product
// S q_i fp16(c_i) is rounded UP to an integer (floor(x) + 1) and integer sums are exact, hence
//     coarse >= S * true * (1 - 2^-11)  -  S * 2^-25 * sum_i q_i .
// The first term is the factor in the threshold.  The second is at most S 2^-25 sqrt(nnz_q) |q| units: 0.05 units for
// a unit-norm query of 2,247 terms at S = 2^15, and never more for un-normalised rows (S |q||c| < 2^16 caps S |q|);
// the fp32 rounding of the scaled query weight S q_i is another ~2^-9 units.  Both sit inside the 2 units subtracted
// from the threshold (tests/test_gpu_soundness.py attacks exactly these corners).  The survivors (a few per thousand
// more than the true pairs) are re-scored from the fp32 store by k_rescore and pruned at theta.  A sum exceeds S*q.c by at
// most one unit per shared term; the host picks S so that S*|q||c| + min(nnz_q, nnz_c) stays below 2^16 (no carry).
// __launch_bounds__(512, 4): two workgroups of 8 waves per CU = 4 waves per SIMD = at most 128 VGPRs.  The kernel
// sits right at that edge; without the bound a small edit tipped it to 130 VGPRs = one workgroup per CU = 1.6x slower.
template <int BLOCK, int U, int LONGCAP, int SURVCAP, bool SHARD, int CHUNK = 16, bool VROWS = false, bool SIGNED = false, bool LONGPF = false,
          bool ACC8 = false>
__global__ __launch_bounds__(BLOCK, BLOCK <= 512 ? 2 * BLOCK / 256 : BLOCK / 256) void k_probe_coarse(const ProbeArgs a) {
  constexpr int NW = BLOCK / kWave;
  // SLOT2: the posting's slot field is the LDS BYTE OFFSET of the candidate's accumulator: slot * 2 for 16-bit accumulators
  // (tiles of <= 32768 rows, see pack_coarse), the slot itself for 8-bit ones (ACC8: 65536 rows in the same 64 KB).
  // ACC8 serves thin rounds (term shards: a dozen terms per row and shard): twice the candidates per round at the same two
  // workgroups per CU, i.e. half the rounds.  Sums are S * (normalised partial) + one unit per shared term, S <= 2^7; the host
  // picks S so that they stay below 2^8 (no carry into the neighbour's byte).
  constexpr bool SLOT2 = BLOCK <= 512;
  constexpr uint32_t ABITS = ACC8 ? 8u : 16u;  // accumulator width
  // WIDE (ACC8 in the 1024-thread kernel): 131072-row tiles, one workgroup per CU, postings slot << 15 | fp15 weight
  // (pack_coarse_wide).  The sparse regime (C5: 13 postings per (65536-row tile, term)) is bound by the traffic of
  // half-empty 128-B lines -- one for the segment descriptor, one for the postings -- and twice the rows per tile
  // halve both per posting visit.
  constexpr bool WIDE = ACC8 && BLOCK > 512;
  constexpr int CH = CHUNK;              // postings per chunk: LPC lanes x 2 postings (8 B per lane)
  constexpr int LPC = CH / 2;            // lanes per chunk
  constexpr int GPW = kWave / LPC;       // chunks per wave step
  constexpr int WIN = GPW * U;
  static_assert(WIN <= kWave, "the window strip is cleared by one store per lane");
  // STATIC LDS (sized for the largest tile this instantiation serves): the compiler then knows every LDS address and
  // folds the array bases into the instructions' offset fields; through a dynamic `extern __shared__` base every LDS
  // access of the hot loop paid a VALU add of the (link-time) base.
  constexpr int kLongLen = kLongLenW;  // (128 / 512 for the prefetched sweeps measured slower: 710 / 884 vs 614 ms, C3 with Zipf(1))
  constexpr int CBMAX = WIDE ? 131072 : (BLOCK <= 512 && !ACC8 ? 32768 : 65536);
  constexpr int APW = 32 / (int)ABITS;  // accumulators per LDS word
  __shared__ __attribute__((aligned(16))) uint32_t acc[CBMAX / APW + kWave];  // two u16 / four u8 accumulators per word (+ slack)
  __shared__ uint2 items[NW * WIN + kWave];  // [NW][WIN] {byte offset of the chunk's first posting, weight bits} (+ one spare entry per lane)
  __shared__ uint2 longs[3 * LONGCAP];     // [3][LONGCAP]
  __shared__ float long_w[3 * LONGCAP];    // [3][LONGCAP]
  __shared__ uint32_t surv[SURVCAP];
  __shared__ uint32_t ctr[16];
  __shared__ unsigned long long stat[4];
  unsigned char *const smem_raw = reinterpret_cast<unsigned char *>(acc);

  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid / kWave), ln = tid % kWave;
  const int cb = a.cb;
  const int tile = a.tile0 + blockIdx.x / a.n_chunks;
  const int chunk = blockIdx.x % a.n_chunks;
  const int q0 = chunk * a.q_chunk;
  const int q1 = min(a.nq, q0 + a.q_chunk);
  const int qtile = q0 / a.cb;  // (symmetric joins: the chunk lies inside one tile)
  if (a.tri && tile > qtile) return;  // the mirrored half: the workgroup (qtile, the chunks of this tile's rows) finds these pairs
  // rounds run over VIRTUAL rows: a query of more than BLOCK terms is cut into parts of <= BLOCK terms that share
  // the accumulators (the adds of all parts land before the query's candidates are cleared); without a table a
  // virtual row is a query
  constexpr bool vrows = VROWS;  // compiled out of the common kernel: it sits at the 128-VGPR edge
  const int v0 = vrows ? a.vq_first[q0] : q0;
  const int v1 = vrows ? a.vq_first[q1] : q1;
  const int nv = vrows ? a.vq_first[a.nq] : a.nq;
  const int64_t tile_row0 = (int64_t)tile * cb;
  const int kterm = ln * NW + wv;
  const uint32_t lo = (uint32_t)(ln % LPC);  // this lane handles postings 2*lo and 2*lo + 1 of its chunk
  uint2 *wl = items + wv * WIN;
  const float cxs = a.cx_scale;

  const int64_t qbase = a.q_rowptr[q0], qend = a.q_rowptr[q1];
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

  for (int i = tid * 4; i < cb / APW + kWave; i += BLOCK * 4) *reinterpret_cast<uint4 *>(acc + i) = make_uint4(0u, 0u, 0u, 0u);
  if (tid < 16) ctr[tid] = 0;
  unsigned long long my_visits = 0;
  uint32_t wave_cands = 0;  // first touches of the register window (uniform control flow: ballots on the scalar unit)
  uint32_t my_cands = 0;    // first touches of the sweeps (divergent control flow: per lane)
  uint32_t n_long_next = 0;

  // part extent, its query, is it the query's last part; shard rule: 1 / (|q_g| / |q|), the query's half of the normalisation
  struct RowExt { int qb; int nnz; int q; bool last; float inv_qs; };
  struct TermW { uint32_t term; float w; bool valid; };
  struct Seg { uint32_t s, len; float w; };
  struct WaveWork {
    uint32_t s, len, excl;
    float w;
    int totch, tw;
    apss_u32x2 pc[U];   // two coarse postings per lane and step
    float wq[U];        // query weight x cx_scale of the step's chunk
  };
  auto load_R = [&](int v) {
    RowExt r;
    const int vv = min(v, nv - 1);
    int64_t b, e;
    if (vrows) {
      b = a.vrow_ptr[vv];
      e = a.vrow_ptr[vv + 1];
      r.q = a.vrow_q[vv];
      r.last = a.vrow_q[vv + 1] != r.q;
    } else {
      b = uniform_load(a.q_rowptr + vv);
      e = uniform_load(a.q_rowptr + vv + 1);
      r.q = vv;
      r.last = true;
    }
    r.qb = (int)(b - qbase);
    r.nnz = v < v1 ? (int)(e - b) : 0;
    r.inv_qs = 1.0f;
    if (SHARD) {
      const float qsv = uniform_load(a.q_scale + r.q);
      r.inv_qs = qsv > 0.f ? 1.0f / qsv : 0.f;
    }
    return r;
  };
  auto load_I = [&](const RowExt &r) {
    TermW t;
    t.valid = kterm < r.nnz;
    const uint32_t off = t.valid ? (uint32_t)(r.qb + kterm) * 4u : kOob;
    t.term = __builtin_amdgcn_raw_buffer_load_b32(rs_qi, off, 0, 0);
    t.w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_qv, off, 0, 0));
    return t;
  };
  auto load_P = [&](const TermW &t) {
    Seg g;
    const apss_u32x2 sg = __builtin_amdgcn_raw_buffer_load_b64(rs_tp, t.term * 8u, 0, 0);
    g.s = sg.x;
    g.len = t.valid ? sg.y : 0u;
    g.w = t.w;
    return g;
  };
  auto flatten = [&](WaveWork &f, const Seg &g, const RowExt &r, int li) {
    uint32_t len = g.len;
    my_visits += len;
    if (len > (uint32_t)kLongLen) {
      const uint32_t k = atomicAdd(&ctr[li], 1u);
      if (k < (uint32_t)LONGCAP) {
        longs[li * LONGCAP + k] = make_uint2(g.s, len);
        long_w[li * LONGCAP + k] = g.w * (SHARD ? r.inv_qs : 1.0f);
        len = 0;
      }
    }
    const uint32_t nch = (len + CH - 1) / CH;
    const uint32_t incl = wave_incl_scan(nch);
    const uint32_t excl = incl - nch;
    f.s = g.s;
    f.len = len;
    f.excl = excl;
    f.w = g.w * (SHARD ? r.inv_qs : 1.0f);
    f.tw = wv < r.nnz ? (r.nnz - wv + NW - 1) / NW : 0;
    f.totch = __builtin_amdgcn_readlane((int)incl, kWave - 1);
    const uint32_t wbits = __float_as_uint(cxs * f.w);
    auto put = [&](const uint32_t k) {
      if (k < nch && excl + k < (uint32_t)WIN)
        wl[excl + k] = make_uint2((g.s + k * CH) * 4u, wbits);  // {byte offset of the chunk's first posting, weight bits}
    };
    // the first chunks of every segment without exec masks: a lane with no such chunk writes its own spare entry (a select
    // on the address; a mask is a trip through the scalar unit, ~64 cycles of the wave's serial issue each)
    auto put_always = [&](const uint32_t k) {
      uint2 *const dst = k < nch && excl + k < (uint32_t)WIN ? wl + excl + k : items + NW * WIN + ln;
      *dst = make_uint2((g.s + k * CH) * 4u, wbits);
    };
    constexpr uint32_t kPuts = CH == 16 ? 3u : 5u;  // covers segments of up to 48 / 40 postings without the loop
#pragma unroll
    for (uint32_t k = 0; k < kPuts; ++k) put_always(k);
    if (__any(nch > kPuts))
      for (uint32_t k = kPuts; __any(k < nch && excl + k < (uint32_t)WIN); ++k) put(k);
  };
  // second half of the staging: every LPC lanes take one chunk of the strip and start its posting load
  auto flatten_loads = [&](WaveWork &f) {
    uint2 it[U];
#pragma unroll
    for (int u = 0; u < U; ++u) it[u] = wl[u * GPW + ln / LPC];
#pragma unroll
    for (int u = 0; u < U; ++u) asm volatile("" : "+v"(it[u].x), "+v"(it[u].y));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // a posting word of zero is no posting: the padding after a segment's last posting is zero-filled by the build,
      // and an empty slot of the strip holds an out-of-range offset, which reads as zero
      // a strip slot past the wave's last chunk holds a stale descriptor: its load is sent out of range (-> zero)
      f.wq[u] = __uint_as_float(it[u].y);
      f.pc[u] = __builtin_amdgcn_raw_buffer_load_b64(rs_po, u * GPW + ln / LPC < f.totch ? it[u].x + lo * 8u : kOob, 0, 0);
    }
  };

  RowExt R1 = load_R(v0 + 1), R2 = load_R(v0 + 2), R3 = load_R(v0 + 3);
  TermW I2, I3;
  Seg P1, P2;
  WaveWork wfa, wfb;
  {
    const RowExt R0 = load_R(v0);
    const TermW I0 = load_I(R0), I1 = load_I(R1);
    I2 = load_I(R2);
    const Seg P0 = load_P(I0);
    P1 = load_P(I1);
    __syncthreads();
    flatten(wfa, P0, R0, 0);
    flatten_loads(wfa);
  }
  __syncthreads();
  n_long_next = ctr[0];

  bool multi = false;    // the current query has earlier parts whose slots are no longer in registers
  bool rescan = false;   // an earlier part overflowed the survivor list: scan the accumulators at the last part
  auto round = [&](WaveWork &w0, WaveWork &w2, const int v, const int l3, const RowExt cur) {  // (l3: ring slot, 0/1)
    const int par = (v - v0) & 1;
    const int q = cur.q;
    // coarse threshold: products are rounded UP (add16), so a coarse sum is never below S * sum(q_i * fp16(c_i)); the 2
    // units cover the fp32 rounding of the scaled query weight.  Shard rule (p_g >= theta r_q r_c, r_x = |x_g| / |x|): the
    // postings hold c_i / r_c (index build) and the query weights are divided by r_q (flatten), so a sum is
    // p_g / (r_q r_c) and the threshold is the same theta for every query and candidate.
    constexpr int slack = 2;
    const int thr_c = (int)a.cx_theta - slack;
    const uint32_t thr1 = (uint32_t)max(thr_c, 1) - 1u;

    const RowExt R4 = load_R(v + 4);  // (scalar loads: uniform_load)

    auto crossed = [&](const uint32_t slot, const uint32_t p, const uint32_t old16) {
      if (thr1 - old16 < p) {
        const uint32_t k = atomicAdd(&ctr[5 + 2 * par], 1u);
        if (k < (uint32_t)SURVCAP) surv[k] = slot;
      }
    };
    // one coarse posting: 16-bit add into the candidate's half of its word; the returning atomic gives the old WORD
    // (the half is extracted later, so that the round's atomics are all in flight before the first wait)
    auto slot_of = [&](const uint32_t pcw) { return WIDE ? pcw >> 15 : (SLOT2 && !ACC8 ? (pcw & 0xffffu) >> 1 : pcw & 0xffffu); };
    // the product of one posting, rounded up: floor(x) + 1 is never below x and >= 1, so that a touch always shows
    // SIGNED (weights of either sign, theta > 0): only the positive products count and a negative one adds a single
    // unit, so a sum is an upper bound of S * score that still grows with every touch -- sound for a filter (a pair
    // with score >= theta has at least that much positive mass); the exact pass applies the signs
    auto prod = [&](const uint32_t pcw, const float wqs) {
      const float x = __builtin_fmaf(wqs, __half2float(__ushort_as_half((unsigned short)(WIDE ? pcw & 0x7fffu : pcw >> 16))), 1.0f);
      return SIGNED ? (uint32_t)max((int)x, 1) : (uint32_t)x;
    };
    // ds_add_rtn_u32 on the word that holds the candidate's half; halves cannot carry (bounded scores)
    auto add16 = [&](const uint32_t pcw, const uint32_t p) -> uint32_t {
      if (WIDE) return atomicAdd(reinterpret_cast<uint32_t *>(smem_raw + ((pcw >> 15) & 0x1fffcu)), p << ((pcw >> 12) & 24u));
      if (SLOT2) return atomicAdd(reinterpret_cast<uint32_t *>(smem_raw + (pcw & 0xfffcu)), p << ((pcw << 3) & 31u));
      const uint32_t slot = pcw & 0xffffu;
      return atomicAdd(&acc[slot >> 1], p << ((slot & 1u) << 4));
    };
    auto half_of = [&](const uint32_t old_word, const uint32_t pcw) {
      // SLOT2: the low bits of the byte offset select the field; v_bfe_u32 takes the offset modulo 32
      if (WIDE) return __builtin_amdgcn_ubfe(old_word, (pcw >> 12) & 24u, 8u);
      return SLOT2 ? __builtin_amdgcn_ubfe(old_word, pcw << 3, ABITS) : (old_word >> ((pcw & 1u) << 4)) & 0xffffu;
    };
    auto visit = [&](const uint32_t pcw, const float wqs) {
      const uint32_t p = prod(pcw, wqs);
      const uint32_t old16 = half_of(add16(pcw, p), pcw);
      my_cands += old16 == 0u ? 1u : 0u;
      crossed(slot_of(pcw), p, old16);
    };
    // the register window, BATCH steps (2 x BATCH atomics) at a time: enough LDS atomics in flight to cover their
    // latency, few enough live registers to keep two workgroups on the CU.  The staging of round v + 1 (its own LDS
    // round trip and its posting loads) runs between the first batch's adds and their tests.
    constexpr int BATCH = 3;
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
        crossed(slot_of(x), px, ox);
        crossed(slot_of(y), py, oy);
      }
    };
    auto issue_batch = [&](const int u0, uint32_t (&p0)[BATCH], uint32_t (&p1)[BATCH], uint32_t (&o0)[BATCH], uint32_t (&o1)[BATCH]) {
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const int u = u0 + j;
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
        if (u < U) {
          o0[j] = w0.pc[u].x ? half_of(o0[j], w0.pc[u].x) : thr1 + 1u;
          o1[j] = w0.pc[u].y ? half_of(o1[j], w0.pc[u].y) : thr1 + 1u;
          my_cands += (o0[j] == 0u ? 1u : 0u) + (o1[j] == 0u ? 1u : 0u);
          n_cross += (thr1 - o0[j] < p0[j] ? 1u : 0u) + (thr1 - o1[j] < p1[j] ? 1u : 0u);
        }
      }
      if (n_cross) {
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
          const int u = u0 + j;
          if (u < U) {
            crossed(slot_of(w0.pc[u].x), p0[j], o0[j]);
            crossed(slot_of(w0.pc[u].y), p1[j], o1[j]);
          }
        }
      }
    };
    {
      uint32_t p0[BATCH], p1[BATCH], o0[BATCH], o1[BATCH];
      issue_batch(0, p0, p1, o0, o1);
      I3 = load_I(R3);
      P2 = load_P(I2);
      flatten(w2, P1, R1, l3 ^ 1);
      check_batch(0, p0, p1, o0, o1);
    }
    if (U <= BATCH) flatten_loads(w2);
#pragma unroll
    for (int u0 = BATCH; u0 < U; u0 += BATCH) {
      uint32_t p0[BATCH], p1[BATCH], o0[BATCH], o1[BATCH];
      issue_batch(u0, p0, p1, o0, o1);
      if (u0 == BATCH) flatten_loads(w2);  // the strip round trip and the loads' issue overlap the second batch's adds
      check_batch(u0, p0, p1, o0, o1);
    }
    if (w0.totch > WIN) {
      if (ln == 0) ctr[4 + 2 * par] = 1;
      for (int c0 = WIN; c0 < w0.totch; c0 += GPW) {
        const uint32_t c = (uint32_t)(c0 + ln / LPC);
        uint32_t st = 0, cn = 0;
        float wq_ = 0.f;
        for (int m = 0; m < w0.tw; ++m) {
          const uint32_t em = (uint32_t)__builtin_amdgcn_readlane((int)w0.excl, m);
          const uint32_t lm = (uint32_t)__builtin_amdgcn_readlane((int)w0.len, m);
          const uint32_t sm = (uint32_t)__builtin_amdgcn_readlane((int)w0.s, m);
          const float wm = cxs * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w0.w), m));
          const uint32_t nm = (lm + CH - 1) / CH;
          const bool sel = c >= em && c < em + nm;
          const uint32_t k = c - em;
          st = sel ? sm + k * CH : st;
          cn = sel ? min((uint32_t)CH, lm - k * CH) : cn;
          wq_ = sel ? wm : wq_;
        }
        const apss_u32x2 two = __builtin_amdgcn_raw_buffer_load_b64(rs_po, st * 4u + lo * 8u, 0, 0);
        if (2u * lo < cn) visit(two.x, wq_);
        if (2u * lo + 1u < cn) visit(two.y, wq_);
      }
    }
    const uint32_t n_long = min(n_long_next, (uint32_t)LONGCAP);
    if constexpr (LONGPF) {
      // LONGPF: the instantiation for the sparse half of a handle with a dense-head block -- a skewed tail, a dozen long
      // segments per round.  They are swept by the whole workgroup, two postings (8 B) per lane and pass, and the first
      // pass of the NEXT TWO segments is requested before the current one is added: a segment's load -> atomic -> test
      // chain otherwise runs start to finish before the next begins (a round then costs n_long memory round trips:
      // C3 with Zipf(1) terms, 753 -> 615 ms).  A zero word is no posting (padding; out-of-range reads return 0).
      // (Its own instantiation: with no long segment in sight this code measured 7 - 10 % slower, uniform C3 and term shards.)
      if (n_long > 0) {
        // the workgroup splits into LG groups of BLOCK / LG threads; group g sweeps segments g, g + LG, ...: a typical long
        // segment (a few hundred postings) fills a quarter of the workgroup, not all of it
        constexpr uint32_t LG = 4, GT = BLOCK / LG, PASS = 2u * GT;
        const uint32_t grp = (uint32_t)tid / GT, gt = (uint32_t)tid % GT;
        auto first_pass = [&](const uint32_t i) {
          const uint32_t j = i * LG + grp;
          const uint2 sg = longs[l3 * LONGCAP + min(j, n_long - 1u)];
          return __builtin_amdgcn_raw_buffer_load_b64(rs_po, j < n_long && 2u * gt < sg.y ? (sg.x + 2u * gt) * 4u : kOob, 0, 0);
        };
        apss_u32x2 n1 = first_pass(0), n2 = first_pass(1);
        for (uint32_t i = 0; i * LG < n_long; ++i) {
          const uint32_t j = min(i * LG + grp, n_long - 1u);
          const bool mine = i * LG + grp < n_long;
          const uint2 sgm = longs[l3 * LONGCAP + j];
          const float wq_ = cxs * long_w[l3 * LONGCAP + j];
          const apss_u32x2 cur = n1;
          n1 = n2;
          n2 = first_pass(i + 2u);
          visit2(cur.x, cur.y, wq_);
          if (mine)
            for (uint32_t k = 2u * gt + PASS; k < sgm.y; k += PASS) {  // further passes of a segment of > PASS postings
              const apss_u32x2 a0 = __builtin_amdgcn_raw_buffer_load_b64(rs_po, (sgm.x + k) * 4u, 0, 0);
              visit2(a0.x, a0.y, wq_);
            }
        }
      }
    } else {
      for (uint32_t j = 0; j < n_long; ++j) {
        const uint2 sgm = longs[l3 * LONGCAP + j];
        const float wq_ = cxs * long_w[l3 * LONGCAP + j];
        // two postings (8 B) per lane and load, two loads in flight per lane
        uint32_t k = 2u * tid;
        for (; k + 2u * BLOCK < sgm.y; k += 4u * BLOCK) {
          const apss_u32x2 a0 = __builtin_amdgcn_raw_buffer_load_b64(rs_po, (sgm.x + k) * 4u, 0, 0);
          const apss_u32x2 a1 = __builtin_amdgcn_raw_buffer_load_b64(rs_po, (sgm.x + k + 2u * BLOCK) * 4u, 0, 0);
          visit2(a0.x, k + 1u < sgm.y ? a0.y : 0u, wq_);  // (a zero word is no posting: spare-word add, no mask)
          visit2(a1.x, k + 2u * BLOCK + 1u < sgm.y ? a1.y : 0u, wq_);
        }
        for (; k < sgm.y; k += 2u * BLOCK) {
          const apss_u32x2 a0 = __builtin_amdgcn_raw_buffer_load_b64(rs_po, (sgm.x + k) * 4u, 0, 0);
          visit2(a0.x, k + 1u < sgm.y ? a0.y : 0u, wq_);
        }
      }
    }
    __syncthreads();

    const uint2 fl = *reinterpret_cast<const uint2 *>(&ctr[4 + 2 * par]);
    n_long_next = ctr[l3 ^ 1];
    const uint32_t n_surv = fl.y;
    // a part that is not the query's last only adds; reporting and clearing wait for the last part
    // a multi-part query reports once, at its last part, by scanning the accumulators (no duplicates across parts)
    if (n_surv > (uint32_t)SURVCAP || multi || !cur.last) rescan = true;
    const bool full_zero = cur.last && (multi || rescan || n_long > 0 || fl.x != 0);
    if ((n_surv > 0 && !rescan) || (cur.last && rescan)) {
      const int64_t qext = a.q_ext[q];
      const unsigned short *acc16p = reinterpret_cast<const unsigned short *>(acc);
      const unsigned char *acc8p = reinterpret_cast<const unsigned char *>(acc);
      auto acc_at = [&](const int c) -> int { return ACC8 ? (int)acc8p[c] : (int)acc16p[c]; };
      if (!rescan) {
        for (uint32_t i = tid; i < (n_surv + kWave - 1) / kWave * kWave; i += BLOCK) {
          bool ok = false;
          uint32_t c = 0;
          if (i < n_surv) {
            c = surv[i];
            ok = a.ext_id[tile_row0 + c] != qext;
          }
          const uint64_t o = wave_append(ok, &a.counters[kCtrResults]);
          if (ok && o < a.res_cap) {
            a.res_q[o] = q;
            a.res_c[o] = (int32_t)(tile_row0 + c);
            a.res_s[o] = (float)acc_at((int)c) / cxs;  // coarse score, replaced by k_rescore
          }
        }
      } else if (cur.last) {
        // some part had more crossings than the list holds: report everything at or above the threshold, once
        for (int i = tid; i < (cb + BLOCK - 1) / BLOCK * BLOCK; i += BLOCK) {
          bool ok = false;
          if (i < cb) {
            const int64_t gs = tile_row0 + i;
            ok = acc_at(i) >= max(thr_c, 1) && gs < a.n_rows && a.ext_id[gs] != qext;
          }
          const uint64_t o = wave_append(ok, &a.counters[kCtrResults]);
          if (ok && o < a.res_cap) {
            a.res_q[o] = q;
            a.res_c[o] = (int32_t)(tile_row0 + i);
            a.res_s[o] = (float)acc_at(i) / cxs;
          }
        }
      }
      __syncthreads();
    }

    if (full_zero) {
      for (int i = tid * 4; i < cb / APW; i += BLOCK * 4) *reinterpret_cast<uint4 *>(acc + i) = make_uint4(0u, 0u, 0u, 0u);
    } else if (cur.last) {
      unsigned short *acc16w = reinterpret_cast<unsigned short *>(acc);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // unconditional: an idle lane (zero word) clears slot 0, which is zero at the end of a query either way
        if (WIDE) {
          smem_raw[w0.pc[u].x >> 15] = 0;  // ds_write_b8
          smem_raw[w0.pc[u].y >> 15] = 0;
        } else if (SLOT2 && ACC8) {
          smem_raw[w0.pc[u].x & 0xffffu] = 0;  // ds_write_b8
          smem_raw[w0.pc[u].y & 0xffffu] = 0;
        } else if (SLOT2) {
          *reinterpret_cast<unsigned short *>(smem_raw + (w0.pc[u].x & 0xffffu)) = 0;  // ds_write_b16
          *reinterpret_cast<unsigned short *>(smem_raw + (w0.pc[u].y & 0xffffu)) = 0;
        } else {
          acc16w[w0.pc[u].x & 0xffffu] = 0;
          acc16w[w0.pc[u].y & 0xffffu] = 0;
        }
      }
    }
    multi = !cur.last;  // the next part (if any) belongs to the same query
    if (cur.last) rescan = false;
    if (tid == 0) {
      ctr[l3] = 0;
      ctr[4 + 2 * (par ^ 1)] = 0;
      ctr[5 + 2 * (par ^ 1)] = 0;
    }
    __syncthreads();

    P1 = P2;
    I2 = I3;
    R1 = R2;
    R2 = R3;
    R3 = R4;
  };
  // the extent of round v is R0 of that round; track it alongside (R1 is round v+1 at the top of round v)
  RowExt cur = load_R(v0);
  for (int v = v0; v < v1; v += 2) {
    RowExt nx = R1;
    round(wfa, wfb, v, 0, cur);
    cur = nx;
    if (v + 1 >= v1) break;
    nx = R1;
    round(wfb, wfa, v + 1, 1, cur);
    cur = nx;
  }
  __syncthreads();
  if (tid < 3) stat[tid] = 0;
  __syncthreads();
  atomicAdd(&stat[0], my_visits);
  atomicAdd(&stat[1], (unsigned long long)my_cands + (ln == 0 ? wave_cands : 0u));
  __syncthreads();
  if (tid == 0) {
    const unsigned long long twice = a.tri && tile < qtile ? 2ull : 1ull;  // (both counts are symmetric in the two tiles)
    atomicAdd(&a.counters[kCtrVisits], stat[0] * twice);
    atomicAdd(&a.counters[kCtrCands], stat[1] * twice);
    atomicAdd(&a.counters[kCtrDevVisits], stat[0]);
  }
}



// ---------------------------------------------------------------------------------------------------------
// exact pass of the two-pass join: full fp32 dot product (CU:98-117) of every pair the filter let through, then the
// `>= theta` prune (IWA:93) with wavefront compaction into the final result list.  16 lanes per pair.
struct RescoreArgs {
  int64_t n_pairs;
  const int32_t *q_row;
  const int32_t *c_slot;
  const int64_t *q_rowptr;
  const int32_t *q_idx;
  const float *q_val;
  const int64_t *c_rowptr;
  const int32_t *c_idx;
  const float *c_val;
  float theta;
  int32_t *out_q;
  int32_t *out_c;
  float *out_s;
  unsigned long long *out_count;
  // chained launch (small batches: no host round trip between the filter and this pass): the number of pairs is what the
  // filter's counter says when this kernel runs, capped at n_pairs (= the candidate list's capacity); null: n_pairs as given
  const unsigned long long *n_pairs_dev;
};

// (The kept pairs of a workgroup's trips are staged in LDS and appended with ONE global atomic per workgroup: a wave-level
// append is an atomic per wave and trip on one address, and same-address atomics are serialised at the memory side -- see
// k_shard_prune, where the same change took 295 -> 47 us.)
constexpr int kRescoreBlock = 512, kRescoreTrips = 1, kRescorePairs = kRescoreBlock / kGroup;  // pairs per workgroup and trip (long rows: a trip is a long chain of dependent loads -- one per workgroup, all pairs in flight at once)
__global__ __launch_bounds__(kRescoreBlock) void k_rescore(RescoreArgs a) {
  __shared__ int32_t st_q[kRescorePairs * kRescoreTrips], st_c[kRescorePairs * kRescoreTrips];
  __shared__ float st_s[kRescorePairs * kRescoreTrips];
  __shared__ uint32_t st_n;
  __shared__ unsigned long long st_base;
  const int tid = threadIdx.x, gl = tid % kGroup, grp = tid / kGroup;
  const int64_t n = a.n_pairs_dev ? min((int64_t)*a.n_pairs_dev, a.n_pairs) : a.n_pairs;
  const int64_t stride = (int64_t)gridDim.x * kRescorePairs;
  // (a grid sized for an upper bound strides over what the counter holds; the usual launch takes one round of trips)
  for (int64_t base = (int64_t)blockIdx.x * kRescorePairs; base < n; base += stride * kRescoreTrips) {  // uniform per workgroup
    if (tid == 0) st_n = 0u;
    __syncthreads();
    for (int trip = 0; trip < kRescoreTrips; ++trip) {
      const int64_t pair = base + (int64_t)trip * stride + grp;
      const bool live = pair < n;
      float s = 0.f;
      int32_t qr = 0, cs = 0;
      if (live) {
        qr = a.q_row[pair];
        cs = a.c_slot[pair];
        const int64_t qb = a.q_rowptr[qr], qe = a.q_rowptr[qr + 1];
        const int64_t cb = a.c_rowptr[cs], ce = a.c_rowptr[cs + 1];
        for (int64_t k = qb + gl; k < qe; k += kGroup) {
          const int32_t t = a.q_idx[k];
          int64_t lo = cb, hi = ce;
          while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (a.c_idx[mid] < t) lo = mid + 1; else hi = mid;
          }
          if (lo < ce && a.c_idx[lo] == t) s += a.q_val[k] * a.c_val[lo];
        }
      }
      for (int o = kGroup / 2; o; o >>= 1) s += __shfl_xor(s, o, kGroup);
      if (live && gl == 0 && s >= a.theta) {
        const uint32_t at = atomicAdd(&st_n, 1u);  // (LDS)
        st_q[at] = qr;
        st_c[at] = cs;
        st_s[at] = s;
      }
    }
    __syncthreads();
    const uint32_t kept = st_n;
    if (tid == 0 && kept) st_base = atomicAdd(a.out_count, (unsigned long long)kept);
    __syncthreads();
    for (uint32_t i = tid; i < kept; i += kRescoreBlock) {
      a.out_q[st_base + i] = st_q[i];
      a.out_c[st_base + i] = st_c[i];
      a.out_s[st_base + i] = st_s[i];
    }
    __syncthreads();  // (the staging area is reused by the next trips)
  }
}

// ---------------------------------------------------------------------------------------------------------
// k_tail_score: the rows a stream of small batches appended since the index was last extended (the TAIL: at most a few
// thousand) are not in the tile index yet -- folding every single-vector message into it costs a pass over a dim-wide
// segment table (2 ms at vectorDim 2^20, the reference's production shape).  A query batch scores them directly: every
// (query, tail row) pair, 16 lanes per pair, exact fp32 dot over the two sorted rows (CU:98-117), `>= theta` (IWA:93),
// self-exclusion by external id (IWA:91); a pair counts as a candidate iff it shares a term (IWA:86: it is reached through
// a posting list), and its shared terms count as posting visits.  Appends to the final result list.
struct TailArgs {
  int64_t nq, n_tail;
  const int64_t *q_rowptr;  // absolute offsets into q_idx / q_val
  const int32_t *q_idx;
  const float *q_val;
  const int64_t *q_ext;
  const int64_t *c_rowptr;  // the store's rowptr; tail rows are slots tail0 .. tail0 + n_tail
  const int32_t *c_idx;
  const float *c_val;
  const int64_t *c_ext;
  int64_t tail0;
  float theta;
  int32_t *out_q, *out_c;
  float *out_s;
  int64_t out_base;               // results already in the list
  unsigned long long *counters;   // [kCtrResults] appended here, [kCtrVisits] shared terms, [kCtrCands] pairs sharing a term
};

__global__ void k_tail_score(TailArgs a) {
  const int64_t pair = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
  const int gl = threadIdx.x % kGroup;
  const bool live = pair < a.nq * a.n_tail;
  float s = 0.f;
  int matches = 0;
  int64_t qr = 0, cs = 0;
  if (live) {
    qr = pair / a.n_tail;
    cs = a.tail0 + pair % a.n_tail;
    const int64_t qb = a.q_rowptr[qr], qe = a.q_rowptr[qr + 1];
    const int64_t cb = a.c_rowptr[cs], ce = a.c_rowptr[cs + 1];
    for (int64_t k = qb + gl; k < qe; k += kGroup) {
      const int32_t t = a.q_idx[k];
      int64_t lo = cb, hi = ce;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a.c_idx[mid] < t) lo = mid + 1; else hi = mid;
      }
      if (lo < ce && a.c_idx[lo] == t) {
        s += a.q_val[k] * a.c_val[lo];
        matches++;
      }
    }
  }
  for (int o = kGroup / 2; o; o >>= 1) {
    s += __shfl_xor(s, o, kGroup);
    matches += __shfl_xor(matches, o, kGroup);
  }
  const bool shares = live && gl == 0 && matches > 0;  // reached through a posting list (its own row included, as in the index)
  const bool lead = shares && a.q_ext[qr] != a.c_ext[cs];
  const bool keep = lead && s >= a.theta;
  if (__ballot(shares)) {
    int v = shares ? matches : 0;
    for (int o = kWave / 2; o; o >>= 1) v += __shfl_xor(v, o);
    const unsigned long long touched = __ballot(lead);
    if (__lane_id() == 0) {
      atomicAdd(&a.counters[kCtrVisits], (unsigned long long)v);
      if (touched) atomicAdd(&a.counters[kCtrCands], (unsigned long long)__popcll(touched));
    }
  }
  const uint64_t o = wave_append(keep, &a.counters[kCtrResults]);
  if (keep) {
    a.out_q[a.out_base + o] = (int32_t)qr;
    a.out_c[a.out_base + o] = (int32_t)cs;
    a.out_s[a.out_base + o] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------
// exact partial score of named pairs over the shard's dims (two sorted index lists, 16 lanes per pair:
// each lane binary-searches its share of the query's entries in the candidate row)
struct PartialArgs {
  int64_t n_pairs;
  const int32_t *q_row;
  const int32_t *c_slot;
  const int64_t *q_rowptr;
  const int32_t *q_idx;
  const float *q_val;
  const int64_t *c_rowptr;
  const int32_t *c_idx;
  const float *c_val;
  float *out;
  int64_t nq, n_rows;  // bounds of q_row / c_slot: a pair outside them scores 0 instead of faulting
};

__global__ void k_partial_scores(PartialArgs a) {
  const int64_t pair = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
  const int gl = threadIdx.x % kGroup;
  if (pair >= a.n_pairs) return;
  const int64_t qr = a.q_row[pair], cs = a.c_slot[pair];
  if (qr < 0 || qr >= a.nq || cs < 0 || cs >= a.n_rows) {  // caller error: never turn it into a wild read
    if (gl == 0) a.out[pair] = 0.f;
    return;
  }
  const int64_t qb = a.q_rowptr[qr], qe = a.q_rowptr[qr + 1];
  const int64_t cb = a.c_rowptr[cs], ce = a.c_rowptr[cs + 1];
  float s = 0.f;
  for (int64_t k = qb + gl; k < qe; k += kGroup) {
    const int32_t t = a.q_idx[k];
    int64_t lo = cb, hi = ce;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (a.c_idx[mid] < t) lo = mid + 1; else hi = mid;
    }
    if (lo < ce && a.c_idx[lo] == t) s += a.q_val[k] * a.c_val[lo];
  }
  for (int o = kGroup / 2; o; o >>= 1) s += __shfl_xor(s, o, kGroup);
  if (gl == 0) a.out[pair] = s;
}

// ---------------------------------------------------------------------------------------------------------
// MERGED rounds of the thin-round filter (apss_even.hpp): M = 2^m neighbouring query rows share a round.
// k_prenorm_rows: the staged weights divided by their row's shard factor (the kernel otherwise divides per round, by ONE factor);
// 16 lanes per row.
__global__ void k_prenorm_rows(const int64_t *rowptr, const float *val, const float *row_scale, int64_t n_rows, float *out) {
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
  const int gl = threadIdx.x % kGroup;
  if (row >= n_rows) return;
  const float sc = row_scale[row];
  const float inv = sc > 0.f ? 1.0f / sc : 0.f;
  const int64_t e = rowptr[row + 1];
  for (int64_t k = rowptr[row] + gl; k < e; k += kGroup) out[k] = val[k] * inv;
}

// k_expand_merged: the filter reported (round V, candidate c): the SUM of the round's M filter sums crossed the threshold at c.
// Every query row of the round becomes a survivor (M V + j, c), j < M, except c's own row (self-exclusion by external id,
// IWA:91, which the kernel left to this pass) -- out of place, into a second list with its own counter.  n_in: the rounds
// reported (the filter's counter); if it exceeded the list, `over` (kCtrOver) is raised to the capacity that holds all of them
// expanded, so that the caller's overflow path grows the lists and runs the probe again.
__global__ void k_expand_merged(const int32_t *in_q, const int32_t *in_c, const float *in_s, const unsigned long long *n_in, uint64_t cap,
                                int32_t mlog, int64_t nq_rows, const int64_t *q_ext, const int64_t *c_ext, int32_t *out_q, int32_t *out_c,
                                float *out_s, unsigned long long *counter, unsigned long long *over) {
  const uint64_t n_rep = (uint64_t)*n_in, n = min(n_rep, cap);
  if (blockIdx.x == 0 && threadIdx.x == 0 && n_rep > cap) atomicMax(over, (unsigned long long)(n_rep << mlog));
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < n; base += stride) {  // (uniform per workgroup: wave_append)
    const uint64_t i = base + threadIdx.x;
    int64_t q0 = 0;
    int32_t c = 0;
    float sc = 0.f;
    int64_t cext = 0;
    bool one = false;  // a pair already: reported by a workgroup that ran its rows one per round (self-exclusion done there)
    if (i < n) {
      const int32_t rq = in_q[i];
      one = (rq & kUnmergedBit) != 0;
      q0 = one ? (int64_t)(rq & ~kUnmergedBit) : (int64_t)rq << mlog;
      c = in_c[i];
      sc = in_s[i];
      cext = c_ext[c];
    }
    for (int j = 0; j < (1 << mlog); ++j) {
      const int64_t q = q0 + j;
      const bool ok = i < n && q < nq_rows && (one ? j == 0 : q_ext[q] != cext);
      const uint64_t o = wave_append(ok, counter);
      if (ok && o < cap) {
        out_q[o] = (int32_t)q;
        out_c[o] = c;
        out_s[o] = sc;
      }
    }
  }
}

// (one launch instead of a device-to-device copy and a fill: counters[to] = counters[from], counters[from] = 0)
__global__ void k_counter_move(unsigned long long *counters, int from, int to) {
  counters[to] = counters[from];
  counters[from] = 0ull;
}

// k_shard_prune: behind k_expand_merged on a term shard.  A crossing of a merged round stands for every row of the round, so
// half of the expanded pairs (M = 2) are there for their neighbour's sake.  Each pair's EXACT partial score over the shard's
// rows (two sorted index lists, 16 lanes per pair, as k_partial_scores) is tested against the shard rule itself,
//   p_g >= theta * r_q * r_c   (r = |x_g| / |x|: what the filter's normalised sums are an upper bound of),
// with a relative slack of 1e-4 for the fp32 sum (a pair >= theta overall satisfies the rule in at least one shard --
// Cauchy-Schwarz over the shards -- passes that shard's filter, and stays here).  What the shard hands to the exchange is then
// no longer than without merging.  Out of place, as k_expand_merged; an overflowed input is reported through `over`.
struct ShardPruneArgs {
  const int32_t *in_q, *in_c;
  const float *in_s;
  const unsigned long long *n_in;
  uint64_t cap;
  const int64_t *q_rowptr;  // the staged query rows (the shard's slice; the tail view on a shard with a dense-head block)
  const int32_t *q_idx;
  const float *q_val;
  const float *q_sub;
  const int64_t *c_rowptr;  // the indexed rows, by slot
  const int32_t *c_idx;
  const float *c_val;
  const float *c_sub;
  float theta;
  int32_t *out_q, *out_c;
  float *out_s;
  unsigned long long *counter;
  unsigned long long *over;
};

// (One global atomic per WORKGROUP and sixteen trips, not one per wave and trip: every wave that keeps a pair appends through the
// same address, and same-address atomics are serialised at the memory side -- rocprofv3 on a T = 8 shard of C3: 295 us for
// 197k pairs with a wave-level append, i.e. 49k atomics of ~6 ns.  The kept pairs of up to 16 trips are staged in LDS.)
constexpr int kPruneBlock = 256, kPruneTrips = 16, kPrunePairs = kPruneBlock / kGroup;  // pairs per workgroup and trip
__global__ __launch_bounds__(kPruneBlock) void k_shard_prune(ShardPruneArgs a) {
  __shared__ int32_t st_q[kPrunePairs * kPruneTrips], st_c[kPrunePairs * kPruneTrips];
  __shared__ float st_s[kPrunePairs * kPruneTrips];
  __shared__ uint32_t st_n;
  __shared__ unsigned long long st_base;
  const int tid = threadIdx.x, gl = tid % kGroup, grp = tid / kGroup;
  const uint64_t n_rep = (uint64_t)*a.n_in, n = min(n_rep, a.cap);
  if (blockIdx.x == 0 && tid == 0 && n_rep > a.cap) atomicMax(a.over, (unsigned long long)n_rep);
  const uint64_t stride = (uint64_t)gridDim.x * kPrunePairs;
  for (uint64_t base = (uint64_t)blockIdx.x * kPrunePairs; base < n; base += stride * kPruneTrips) {  // (uniform per workgroup)
    if (tid == 0) st_n = 0u;
    __syncthreads();
    for (int trip = 0; trip < kPruneTrips; ++trip) {
      const uint64_t pair = base + (uint64_t)trip * stride + (uint64_t)grp;
      const bool live = pair < n;
      float s = 0.f, bound = 0.f, sc = 0.f;
      int32_t qr = 0, cs = 0;
      if (live) {
        qr = a.in_q[pair];
        cs = a.in_c[pair];
        sc = a.in_s[pair];
        bound = a.theta * (1.0f - 1e-4f) * a.q_sub[qr] * a.c_sub[cs];
        const int64_t qb = a.q_rowptr[qr], qe = a.q_rowptr[qr + 1];
        const int64_t cb = a.c_rowptr[cs], ce = a.c_rowptr[cs + 1];
        for (int64_t k = qb + gl; k < qe; k += kGroup) {
          const int32_t t = a.q_idx[k];
          int64_t lo = cb, hi = ce;
          while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (a.c_idx[mid] < t) lo = mid + 1; else hi = mid;
          }
          if (lo < ce && a.c_idx[lo] == t) s += a.q_val[k] * a.c_val[lo];
        }
      }
      for (int o = kGroup / 2; o; o >>= 1) s += __shfl_xor(s, o, kGroup);
      if (live && gl == 0 && s > 0.f && s >= bound) {
        const uint32_t at = atomicAdd(&st_n, 1u);  // (LDS)
        st_q[at] = qr;
        st_c[at] = cs;
        st_s[at] = sc;
      }
    }
    __syncthreads();
    const uint32_t kept = st_n;
    if (tid == 0 && kept) st_base = atomicAdd(a.counter, (unsigned long long)kept);
    __syncthreads();
    for (uint32_t i = tid; i < kept; i += kPruneBlock) {
      const uint64_t o = st_base + i;
      if (o < a.cap) {
        a.out_q[o] = st_q[i];
        a.out_c[o] = st_c[i];
        a.out_s[o] = st_s[i];
      }
    }
    __syncthreads();  // (the staging area is reused by the next sixteen trips)
  }
}

// ---------------------------------------------------------------------------------------------------------
// k_mirror_survivors: the second half of a SYMMETRIC whole-store join (ProbeArgs::tri).  The filter kernels skipped every
// workgroup whose candidate tile lies above its queries' tile; a survivor (q, c) with c's tile BELOW q's tile stands for the
// pair (c, q) too -- the filter sums are upper bounds in either direction, so a true pair survives in the direction that was
// run, and both directions are re-scored exactly afterwards (k_rescore / the shard's phase 2), each in its own order of
// summation, exactly as without the symmetry.  Pairs inside one tile were found in both directions by the kernels themselves.
// n_before: the survivor count when the filter launches ended (a copy: `counter` grows while this kernel appends).
__global__ void k_mirror_survivors(int32_t *res_q, int32_t *res_c, float *res_s, const unsigned long long *n_before, uint64_t cap,
                                   int32_t cb, unsigned long long *counter) {
  const uint64_t n = min((uint64_t)*n_before, cap);
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < n; base += stride) {  // (uniform per workgroup: wave_append)
    const uint64_t i = base + threadIdx.x;
    int32_t q = 0, c = 0;
    float sc = 0.f;
    bool m = false;
    if (i < n) {
      q = res_q[i];
      c = res_c[i];
      sc = res_s[i];
      m = c / cb < q / cb;
    }
    const uint64_t o = wave_append(m, counter);
    if (m && o < cap) {
      res_q[o] = c;
      res_c[o] = q;
      res_s[o] = sc;
    }
  }
}

}  // namespace apss
