// apss_head.hpp -- dense-head path of the probe (gfx950 / CDNA4 MFMA).
//
// Under a skewed term distribution (TF-IDF is Zipfian) a few terms occur in a large fraction of the vectors: their
// posting lists are N long and the inverted-index probe (IndexingWorkerActor.scala:101-109) visits df_t^2 postings for
// each of them.  Those terms are taken OUT of the inverted index and kept as a dense block instead: row c of
//     W [rows x KH]  (bf16),   w_c = c_H * |c| / |c_H|     (c_H = the row restricted to the KH head terms)
// so that  w_q . w_c = cos_H(q, c) * |q| |c|.  Layout in HBM: tiles of 64 rows; inside a tile CHUNK-major --
// W[tile][chunk k/8][row % 64][k % 8], 16-B chunks -- so that a tile is copied into LDS linearly (LDS-DMA writes
// wave-uniform base + lane * 16) and the MFMA fragment of k-step kk is 32 consecutive rows of one chunk: a contiguous,
// bank-conflict-free LDS read at an immediate offset (no swizzle, no padding, no address registers).  The arithmetic is the reference's dot product
// (CommonUtils.scala:110-115) restricted to the head dims, as a [queries x KH] x [KH x candidates] bf16 contraction
// on the matrix cores.
//
// How the two halves are joined (exact, no pair lost).  Split every vector into head and tail part.  For a pair with
// q.c >= theta:  q.c = p_H + p_T,  p_g <= |q_g||c_g|,  |q_H||c_H| + |q_T||c_T| <= |q||c|  (Cauchy-Schwarz twice), hence
//     p_H >= theta |q_H||c_H| / (|q||c|)   or   p_T >= theta |q_T||c_T| / (|q||c|)
// (if both failed, the sum would stay below theta).  The first test is  w_q . w_c >= theta  -- k_head_gemm, a FILTER in
// bf16 whose threshold is lowered by the rounding bound below; the second is the term-shard rule of the sparse filter
// (k_probe_coarse<SHARD> with the ratios |x_T|/|x| as scales).  The union of both candidate lists (k_pair_dedup) is
// re-scored exactly over the FULL rows by k_rescore (CommonUtils.scala:98-117) and pruned at theta
// (IndexingWorkerActor.scala:93), exactly as the survivors of the plain two-pass join are.
//
// Wider heads = FOLDED columns.  A head of more than 256 terms is still ONE block of 256 columns: the `exact` (128) most
// frequent terms keep a column each, every further term i adds into column  exact + (i - exact) mod (256 - exact).  The row is
//     w_c = c~ * |c| / |c_H|,   c~ = the mixed row (own columns + folded sums),  |c_H| = the TRUE norm of the head entries,
// and the test is the same  w_q . w_c >= theta.  With non-negative weights the mixed dot product is an UPPER bound of the
// true partial dot over H --
//     sum_col (sum_{t in col} q_t) (sum_{t in col} c_t)  >=  sum_{t in col} q_t c_t        (the cross terms are >= 0)
// -- so no pair the exact test would pass is lost; what it passes in excess (two rows holding DIFFERENT terms of one
// column) is bounded by the rows' other entries: a row holds m ~ 30-60 folded terms, a chance pair collides in ~m^2/128
// columns, each worth ~1/m^2-ish of the head's cosine -- 0.1-0.2 in all, far below theta; every candidate is re-scored exactly
// anyway.  Cost: the contraction of a 256-term head, WHATEVER the number of terms -- while the posting visits the inverted
// index keeps fall like 1 / K under a Zipfian term distribution, and with them the long segments the sparse filter is slowest
// on.  profiles/microbench/mixed_block_model.py models the pass counts of the splits (128 + 128, 64 + 192 and even 0 + 256
// pass only the true near-duplicates of a 30,000-row sample; 192 + 64 and every 128-column block pass 10x .. 1e5x more).
// Round 3 went there in steps, each measured: several EXACT blocks of 256 under the "or" rule (blocks 2..4
// hold 2-6 terms of a row, rows with a single term of a block pair up at cosine 1: 4e7 .. 1.5e9 chance pairs), then 256
// exact columns + a SECOND block of 256 or 128 folded columns with its own test (APSS_DEBUG=fold_w=256|128 still builds
// that form: k_head_gemm's `chunk0` / `kt` arguments address a block inside a wider row), then the one mixed block: ONE test
// over the whole head is also more selective than the "or" of two.
//
// Rounding bound of the contraction: bf16 keeps 8 significant bits, round-to-nearest errs by <= 2^-8 relative, a
// product of two rounded factors by <= 2^-7 + 2^-16, and sum_i |a_i b_i| <= |a||b| <= B (B = the largest |q||c| of the
// call), so |bf16 dot - exact| <= 0.00783 B; the fp32 accumulation of <= 256 products adds < 2e-5 B.  The host
// passes thr = theta - 0.0080 B.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "apss_kernels.hpp"

namespace apss {

typedef __attribute__((ext_vector_type(8))) __bf16 apss_bf16x8;
typedef __attribute__((ext_vector_type(16))) float apss_f32x16;
typedef __attribute__((ext_vector_type(4))) int apss_i32x4;     // 16 int8 weights: one fragment of v_mfma_i32_32x32x32_i8
typedef __attribute__((ext_vector_type(16))) int apss_i32x16;

// ---- INT8 rendering of W (round 4).  The block is a FILTER for non-negative weights, so its rows may be rounded UP:
//     u_c = ceil(w_c * S)  (S = 127 / the largest row norm the handle has seen when the block is first packed; 0 stays 0)
// and  sum_k u_q[k] u_c[k] >= S^2 w_q . w_c  holds EXACTLY (integer products, int32 accumulation: no rounding bound at all),
// so the test  acc >= floor(S^2 (theta - slack))  loses no pair the exact head test passes.  What it passes in excess: every
// NON-ZERO element is over-estimated by less than one unit of 1/S, i.e. a pair's dot by < (sum of q over the shared columns
// + sum of c over them) / S + shared / S^2 -- ~0.03 for a chance pair of 60-term rows sharing ~14 columns, against a
// threshold of 0.9.  v_mfma_i32_32x32x32_i8 multiplies twice the K of the bf16 form in the same cycles (MI355X_MICROARCH.md,
// matrix cores) and a row is 256 B instead of 512 B: half the MFMAs, half the tile stream, half the W panel per candidate.
// Same 16-B chunk layout (a chunk holds 16 columns instead of 8), same LDS-DMA copy, same fragment reads: A and B take their
// 16 bytes from the same chunk of their rows, so whatever order the instruction sums the 32 columns of a k-step in, column k
// of the query meets column k of the candidate.

constexpr int kHeadQBlock = 512;   // query slots per workgroup (8 waves x 64)
constexpr int kHeadCTile = 64;     // candidate rows per LDS tile
// candidate rows per LDS tile of k_head_gemm
__host__ __device__ constexpr int head_tile_rows(int) { return kHeadCTile; }

__device__ __forceinline__ uint16_t f32_to_bf16_rn(float f) {
  const uint32_t u = __float_as_uint(f);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);  // finite inputs only (ingest rejects the rest)
}

// ---------------------------------------------------------------------------------------------------------
// sampled document frequencies of the store (row r counts iff r % stride == 0): the head-term policy only needs the
// shape of the distribution, and a full histogram would be 1e8 same-line global atomics per C3 step
__global__ void k_df_sample(const int64_t *rowptr, const int32_t *idx, int64_t n_rows, int64_t stride, uint32_t *df) {
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const int lane = threadIdx.x % kWave;
  const int64_t row = wave * stride;
  if (row >= n_rows) return;
  const int64_t b = rowptr[row], e = rowptr[row + 1];
  for (int64_t k = b + lane; k < e; k += kWave) atomicAdd(&df[idx[k]], 1u);
}

// ---------------------------------------------------------------------------------------------------------
// k_head_pack: one wave per row of a CSR batch -> its row of W and its tail ratio |x_T| / |x| (the scale of the sparse
// filter's shard rule).
// element offset of (row, chunk) in a tiled W of width kh
// (kh: the TOTAL width of a W row -- 64 | 128 | 256, or 256 B for B blocks; chunk counts across the blocks)
__host__ __device__ __forceinline__ int64_t head_chunk_off(int64_t row, int chunk, int kh) {
  return (((row >> 6) * (kh / 8) + chunk) * kHeadCTile + (row & 63)) * 8;
}
// BYTE offset of (row, 16-B chunk) in a tiled W whose rows are `cpr` chunks wide (bf16: kh / 8 chunks, int8: kh / 16)
__host__ __device__ __forceinline__ int64_t head_chunk_byte_off(int64_t row, int chunk, int cpr) {
  return (((row >> 6) * cpr + chunk) * kHeadCTile + (row & 63)) * 16;
}
constexpr int kHeadBlock = 256;      // columns per block of a two-block head
constexpr int kHeadMaxBlocks = 2;    // (the two-block experiment form: 256 columns + a block of folded columns with its own norm)
constexpr int kHeadMaxFold = 127;    // heads of up to 256 * 128 = 32768 terms

struct HeadPackArgs {
  const int64_t *rowptr;   // absolute offsets into idx / val
  const int32_t *idx;
  const float *val;
  int64_t row0, row1;      // rows [row0, row1) of that CSR are packed; W rows [w_row0 + row1 - row0, w_pad) are zeroed
  const int32_t *head_pos; // [dim] position of a term in the dense block, -1 = tail term
  int32_t kh;              // total width of a W row: 64 | 128 | 256, or 512 (two blocks of 256, each scaled by its own norm; several
                           // terms may share a column of the second: head_pos maps them to the same position, their values add)
  uint16_t *W;             // tiled; CSR row r lands in W row w_row0 + r - row0
  int64_t w_row0, w_pad;   // w_pad: a multiple of 64 (the GEMM reads whole tiles)
  float *ratio_t;          // [..] |x_T| / |x| per row, indexed like the W rows
  unsigned int *head_nonempty;  // += rows with at least one head entry
  // a term shard packs W from the batch as the caller handed it in (whole rows; its store keeps its term range only):
  const float *row_inv;    // [rows of the CSR] or null: factor of every value (APSS_FLAG_NORMALIZE, k_ingest_count)
  float prune_above;       // an entry counts iff value * row_inv > prune_above (APSS_FLAG_VALUE_PRUNE; -inf: every entry)
  int32_t part, n_parts;   // head_nonempty counts the rows of the W tiles t % n_parts == part only (n_parts <= 1: every row)
  int32_t fold_from;       // columns >= fold_from are FOLDED (several terms add into them); below: one term per column
  float i8_scale;          // > 0: the INT8 rendering -- element = ceil(w * i8_scale), one byte per column (W is then a byte array
                           //   of kh-byte rows); 0: bf16
  unsigned int *overflow;  // |= 1 when an element would exceed 127 (the handle then falls back to bf16 for good)
};

// one wave per W row, 8 rows (one 128-B line per chunk) per workgroup; the workgroup covers W rows [8 g, 8 g + 8)
__global__ __launch_bounds__(512) void k_head_pack(HeadPackArgs a) {
  __shared__ __attribute__((aligned(16))) float rowbuf[8][kHeadBlock * kHeadMaxBlocks];  // fp32: a folded column is a sum
  __shared__ unsigned int nz;
  if (threadIdx.x == 0) nz = 0;
  __syncthreads();
  const int wv = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const int64_t g0 = (a.w_row0 / 8 + blockIdx.x) * 8;  // first W row of this group
  const int64_t wr = g0 + wv;
  const int64_t row = a.row0 + (wr - a.w_row0);         // CSR row of this wave
  const bool real = wr >= a.w_row0 && row < a.row1;
  for (int i = lane; i < a.kh; i += kWave) rowbuf[wv][i] = 0.f;
  if (real) {
    const int64_t b = a.rowptr[row], e = a.rowptr[row + 1];
    float full2 = 0.f, t2 = 0.f, h2b[kHeadMaxBlocks] = {0.f, 0.f};
    const float inv = a.row_inv ? a.row_inv[row] : 1.0f;
    for (int64_t k = b + lane; k < e; k += kWave) {
      const float v = a.val[k] * inv;
      const int32_t t = a.idx[k];
      const int32_t hp = a.head_pos[t];
      if (!(v > a.prune_above)) continue;
      full2 += v * v;
      if (hp < 0) t2 += v * v;
#pragma unroll
      for (int j = 0; j < kHeadMaxBlocks; ++j) h2b[j] += (hp >= 0 && (hp >> 8) == j) ? v * v : 0.f;  // block of a term: position / 256
    }
    for (int o = kWave / 2; o; o >>= 1) {
      full2 += __shfl_xor(full2, o);
      t2 += __shfl_xor(t2, o);
#pragma unroll
      for (int j = 0; j < kHeadMaxBlocks; ++j) h2b[j] += __shfl_xor(h2b[j], o);
    }
    float scale[kHeadMaxBlocks];
    float h2 = 0.f;
#pragma unroll
    for (int j = 0; j < kHeadMaxBlocks; ++j) {
      scale[j] = h2b[j] > 0.f ? sqrtf(full2 / h2b[j]) : 0.f;  // every block is normalised by ITS norm: w^b = x_{H_b} |x| / |x_{H_b}|
      h2 += h2b[j];
    }
    // (LDS operations of one wave execute in order: the zero fill above lands before these entries)
    for (int64_t k = b + lane; k < e; k += kWave) {
      const int32_t hp = a.head_pos[a.idx[k]];
      const float v = a.val[k] * inv;
      if (hp >= 0 && v > a.prune_above) {
        // (block 0: a column belongs to one term, a plain store; block 1: the terms of a column ADD -- |x_{H_2}| above is the
        // norm of the entries themselves, not of the folded row)
        // a FOLDED column holds the L2 norm of its terms (squares add here, the root is taken at write-out): by Cauchy-Schwarz
        // per column  sum_{t in col} q_t c_t <= sqrt(sum q_t^2) sqrt(sum c_t^2), an upper bound of the column's share of the
        // dot product like the plain sum round 3 used, but a TIGHTER one (sqrt(a^2 + b^2) <= a + b), and it keeps the row's norm:
        // |c~| = |c_H|, so no element of w_c exceeds |c| -- which is what lets the INT8 rendering fix its scale from the row norms
        if (hp < a.fold_from) rowbuf[wv][hp] = v * scale[0];
        else {
          const float x = v * (hp < kHeadBlock ? scale[0] : scale[1]);
          atomicAdd(&rowbuf[wv][hp], x * x);
        }
      }
    }
    if (lane == 0) {
      // rounded DOWN a hair: the sparse filter divides the row's tail weights by it (errs on the side of reporting more)
      if (a.ratio_t) a.ratio_t[wr] = full2 > 0.f ? fminf(1.0f, sqrtf(t2 / full2)) * 0.999999f : 0.f;
      if (h2 > 0.f && (a.n_parts <= 1 || (int32_t)((wr >> 6) % a.n_parts) == a.part)) atomicAdd(&nz, 1u);
    }
  }
  __syncthreads();
  for (int u = threadIdx.x; u < 8 * a.kh; u += blockDim.x) {  // folded columns: sum of squares -> L2 norm
    const int j = u / a.kh, col = u % a.kh;
    if (col >= a.fold_from) rowbuf[j][col] = sqrtf(rowbuf[j][col]);
  }
  __syncthreads();
  if (a.i8_scale > 0.f) {
    // INT8: 16 columns per 16-B chunk, rounded UP (a sound filter for the non-negative weights the block is built from)
    const int cpr8 = a.kh / 16;
    bool ovf = false;
    for (int u = threadIdx.x; u < cpr8 * 8; u += blockDim.x) {
      const int c = u >> 3, j = u & 7;
      const int64_t w = g0 + j;
      if (w >= a.w_row0 && w < a.w_pad) {
        const float *f = &rowbuf[j][c * 16];
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          uint32_t word = 0;
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const float x = f[4 * e + b];
            int q = x > 0.f ? (int)ceilf(x * a.i8_scale) : 0;
            if (q > 127) {
              ovf = true;
              q = 127;
            }
            word |= (uint32_t)q << (8 * b);
          }
          o[e] = word;
        }
        *reinterpret_cast<uint4 *>(reinterpret_cast<unsigned char *>(a.W) + head_chunk_byte_off(w, c, cpr8)) = make_uint4(o[0], o[1], o[2], o[3]);
      }
    }
    if (ovf && a.overflow) atomicOr(a.overflow, 1u);
    if (threadIdx.x == 0 && nz && a.head_nonempty) atomicAdd(a.head_nonempty, nz);
    return;
  }
  // write-out: unit u = (chunk c, row j of the group): 8 rows x 16 B = one 128-B line per chunk
  const int cpr = a.kh / 8;
  for (int u = threadIdx.x; u < cpr * 8; u += blockDim.x) {
    const int c = u >> 3, j = u & 7;
    const int64_t w = g0 + j;
    if (w >= a.w_row0 && w < a.w_pad) {
      const float *f = &rowbuf[j][c * 8];
      uint4 o;
      o.x = (uint32_t)f32_to_bf16_rn(f[0]) | (uint32_t)f32_to_bf16_rn(f[1]) << 16;
      o.y = (uint32_t)f32_to_bf16_rn(f[2]) | (uint32_t)f32_to_bf16_rn(f[3]) << 16;
      o.z = (uint32_t)f32_to_bf16_rn(f[4]) | (uint32_t)f32_to_bf16_rn(f[5]) << 16;
      o.w = (uint32_t)f32_to_bf16_rn(f[6]) | (uint32_t)f32_to_bf16_rn(f[7]) << 16;
      *reinterpret_cast<uint4 *>(a.W + head_chunk_off(w, c, a.kh)) = o;
    }
  }
  if (threadIdx.x == 0 && nz && a.head_nonempty) atomicAdd(a.head_nonempty, nz);
}

// INT8 rendering: a row with a larger norm than any before needs a smaller scale S' for the WHOLE block (one threshold for
// every pair).  The rows already packed are re-quantised in place: u' = ceil(u * S' / S) >= w S' still holds (u >= w S), so
// the filter stays sound without the original rows (a term shard no longer has them); num / den = S' / S as a fraction of
// 16-bit integers rounded UP.
__global__ void k_head_rescale(unsigned char *W, int64_t n_bytes, uint32_t num, uint32_t den) {
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n_bytes; i += (int64_t)gridDim.x * blockDim.x * 16) {
    uint4 v = *reinterpret_cast<const uint4 *>(W + i);
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      uint32_t o = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const uint32_t u = (w[e] >> (8 * b)) & 0xffu;
        o |= ((u * num + den - 1) / den) << (8 * b);
      }
      w[e] = o;
    }
    *reinterpret_cast<uint4 *>(W + i) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// TAIL VIEW of a CSR batch: the same rows without the entries the dense block holds -- what the inverted index is built
// from and what the sparse filter's rounds stage.  With a deep folded block four fifths of a row's entries are head terms;
// left in the rows they would be staged every round only to find an empty posting segment.  Two passes around a scan, like
// the ingest: count per row, then an ordered compaction by 16-lane groups.
struct TailViewArgs {
  const int64_t *rowptr;    // source CSR (absolute offsets), rows [row0, row1)
  const int32_t *idx;
  const float *val;
  int64_t row0, row1;
  const int32_t *head_pos;  // [dim] >= 0: the term lives in the dense block
  int64_t *cnt;             // [row1 - row0] tail entries per row (pass 1 out, pass 2 in as its exclusive scan `off`)
  const int64_t *off;       // [row1 - row0 + 1]
  unsigned int *summary;    // [0] max tail entries of a row, [1] rows with at least one
  int64_t dst_row0, dst_nnz0;  // destination: row r lands in view row dst_row0 + r - row0, its entries from dst_nnz0 + off[r - row0]
  int64_t *o_rowptr;
  int32_t *o_idx;
  float *o_val;
  uint32_t *o_erow;         // view row of every entry (may be null)
};

__global__ void k_tailv_count(TailViewArgs a) {
  __shared__ unsigned sh[2];
  if (threadIdx.x < 2) sh[threadIdx.x] = 0;
  __syncthreads();
  const int64_t row = a.row0 + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
  const int gl = threadIdx.x % kGroup;
  int c = 0;
  if (row < a.row1)
    for (int64_t k = a.rowptr[row] + gl, e = a.rowptr[row + 1]; k < e; k += kGroup) c += a.head_pos[a.idx[k]] < 0 ? 1 : 0;
  for (int o = kGroup / 2; o; o >>= 1) c += __shfl_xor(c, o, kGroup);
  if (gl == 0 && row < a.row1) {
    a.cnt[row - a.row0] = c;
    atomicMax(&sh[0], (unsigned)c);
    if (c) atomicAdd(&sh[1], 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (sh[0]) atomicMax(a.summary, sh[0]);
    if (sh[1]) atomicAdd(a.summary + 1, sh[1]);
  }
}

__global__ void k_tailv_write(TailViewArgs a) {
  const int64_t row = a.row0 + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
  const int gl = threadIdx.x % kGroup;
  if (row >= a.row1) return;
  const int64_t b = a.rowptr[row], e = a.rowptr[row + 1];
  const int64_t dr = a.dst_row0 + (row - a.row0);
  int64_t out = a.dst_nnz0 + a.off[row - a.row0];
  for (int64_t k0 = b; k0 < e; k0 += kGroup) {
    const int64_t k = k0 + gl;
    int32_t t = 0;
    bool keep = false;
    if (k < e) {
      t = a.idx[k];
      keep = a.head_pos[t] < 0;
    }
    const unsigned long long m = __ballot(keep);
    const unsigned gm = (unsigned)((m >> (__lane_id() & ~(kGroup - 1))) & 0xffffu);
    if (keep) {
      const int64_t o = out + __popc(gm & ((1u << gl) - 1u));
      a.o_idx[o] = t;
      a.o_val[o] = a.val[k];
      if (a.o_erow) a.o_erow[o] = (uint32_t)dr;
    }
    out += __popc(gm);
  }
  if (gl == 0) a.o_rowptr[dr + 1] = a.dst_nnz0 + a.off[row - a.row0 + 1];
}

// ---------------------------------------------------------------------------------------------------------
// k_head_gemm: the dense-head filter.  D[q][c] = w_q . w_c for a block of 512 query slots against a panel of candidate
// rows, v_mfma_f32_32x32x16_bf16; nothing of D is stored: every element is compared with thr in registers and the few
// that pass are appended to the candidate list (wavefront ballot + prefix count, one global atomic per wave).
//
// Shape.  One workgroup = 8 waves = 512 query slots; wave w keeps the A fragments of its 64 slots (2 x KH/16 fragments
// of 8 bf16 = 128 VGPRs at KH = 256) in registers for the whole kernel and the panel's candidate rows stream past it:
// tiles of 64 candidates x KH (32 KB at KH = 256) are copied global -> LDS by LDS-DMA (buffer -> lds, 16 B per lane,
// 1 KiB per wave-instruction) into two buffers, tile t + 1 in flight while the 64 MFMAs of tile t run; one barrier per
// tile.  The tile's chunk-major layout (top of this file) makes the copy linear and every fragment read contiguous.
// Grid: blockIdx -> (panel = blockIdx % P, query block descending); panel p = the tiles t with t % P == p.  Hardware
// sends consecutive workgroups to consecutive XCDs, so with P a multiple of 8 every workgroup of an XCD streams panels
// of one residue class: a panel's tiles are read from HBM once per XCD and then served by that XCD's L2 to the other
// workgroups sweeping it.
// Stored queries (self-join, insert-and-query): D is symmetric over the batch, so candidate tiles ABOVE a query block
// are skipped and an element strictly below it reports both (q, c) and (c, q).
struct HeadGemmArgs {
  const uint16_t *Wq;   // query rows (tiled): indexed by query SLOT when the batch is stored (Wq == Wc), else by query row
  const uint16_t *Wc;   // candidate rows by slot (tiled), zero rows up to the tile boundary
  int64_t wq_rows;      // rows of Wq that may be read
  int64_t n_rows;       // candidate slots
  int64_t q_slot_base;  // slot of query row 0 when the batch is stored in the index, else -1
  int32_t nq;
  int32_t n_qblocks, n_panels, n_ctiles;  // n_ctiles: candidate tiles of head_tile_rows(KH) rows
  int32_t part, n_parts;  // this launch multiplies the candidate tiles t with t % n_parts == part (the block of a term-sharded
                          // join is cut over the GPUs by candidate row: 64-row tiles dealt round-robin; 0, 1: every tile)
  int32_t kt, chunk0;     // total width of a W row (= KH for a single block) and the first 16-B chunk, inside a row, of the block of
                          // KH columns this launch multiplies (0 for the first block, 32 for the folded block behind 256 columns)
  int64_t qblock0;      // first query block's first slot (a multiple of 512)
  const int64_t *q_ext, *c_ext;
  float thr;
  int32_t thr_i;        // INT8 rendering: the threshold in units of 1 / S^2 ...
  float inv_s2;         // ... and 1 / S^2 (a reported filter score = acc * inv_s2)
  int32_t *res_q, *res_c;
  float *res_s;
  uint64_t res_cap;
  unsigned long long *counters;    // [kCtrResults] shared with the sparse filter
  unsigned long long *head_pairs;  // += elements with a positive dot (pairs sharing a head term), self pairs included
  unsigned long long *clk;         // diagnostic kernel of the microbenchmark only: per workgroup {shader cycles, 100-MHz ticks}
};

// the two renderings behind one kernel body: fragment and accumulator types, the MFMA, the threshold
template <bool I8> struct HeadMma;
template <> struct HeadMma<false> {
  using frag = apss_bf16x8;
  using acc = apss_f32x16;
  using elem = float;
  static __device__ __forceinline__ acc mma(const frag &a, const frag &b, const acc &c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ elem thr(const HeadGemmArgs &a) { return a.thr; }
  static __device__ __forceinline__ float score(const HeadGemmArgs &, elem v) { return v; }
  static __device__ __forceinline__ elem vmax(elem x, elem y) { return fmaxf(x, y); }
  static __device__ __forceinline__ elem vmax3(elem x, elem y, elem z) { return __builtin_fmaxf(__builtin_fmaxf(x, y), z); }
};
template <> struct HeadMma<true> {
  using frag = apss_i32x4;
  using acc = apss_i32x16;
  using elem = int;
  static __device__ __forceinline__ acc mma(const frag &a, const frag &b, const acc &c) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ elem thr(const HeadGemmArgs &a) { return a.thr_i; }
  static __device__ __forceinline__ float score(const HeadGemmArgs &a, elem v) { return (float)v * a.inv_s2; }
  static __device__ __forceinline__ elem vmax(elem x, elem y) { return max(x, y); }
  static __device__ __forceinline__ elem vmax3(elem x, elem y, elem z) { return max(max(x, y), z); }
};

template <int KH, bool COUNT = true, int NBUF = 3, int NW = 8, bool I8 = false>
__global__ __launch_bounds__(64 * NW, 2) void k_head_gemm(const HeadGemmArgs a) {
  using MM = HeadMma<I8>;
  using frag_t = typename MM::frag;
  using acc_t = typename MM::acc;
  using elem_t = typename MM::elem;
  constexpr bool PIPE3 = NBUF >= 3;           // NBUF tile buffers in LDS: 2 = barrier at the tile boundary, >= 3 = barrier in mid-tile
  constexpr int QB = 64 * NW;                 // query slots per workgroup (kHeadQBlock in the library; 256 in an experiment of the microbenchmark)
  constexpr int ROWB = I8 ? KH : KH * 2;      // bytes per row
  constexpr int KS = ROWB / 32;               // k-steps: two 16-B chunks each (32x32x16 bf16 / 32x32x32 int8)
  constexpr int SPK = 16 / KS;                // epilogue scan steps (2 accumulators each) per k-step of the other half
  static_assert(SPK * KS == 16, "a row is 128, 256 or 512 bytes");
  constexpr int CPR = ROWB / 16;              // 16-B chunks per row of this block
  constexpr int SUB = 1;                      // 64-row sub-tiles per LDS tile (32-KB tiles at KH < 256, SUB = 256 / KH, measured
  constexpr int CT = kHeadCTile * SUB;        //   slower: narrow blocks are bound by the epilogue, not by the barrier)
  constexpr int NB = CT / 32;                 // 32-candidate column blocks per tile
  constexpr int SUBB = kHeadCTile * ROWB;     // bytes per sub-tile
  constexpr int TILEB = CT * ROWB;            // bytes per tile (contiguous in HBM: consecutive 64-row tiles of W)
  constexpr int PIECES = TILEB / 1024;        // 1-KiB DMA pieces per tile
  constexpr int PPW = PIECES / NW;            // pieces per wave
  constexpr int PF = 2;                       // B fragments requested ahead of the MFMAs that use them
  static_assert(PIECES % NW == 0, "every wave copies the same number of pieces");
  __shared__ __attribute__((aligned(1024))) unsigned char ldsb[NBUF * TILEB];
  __shared__ elem_t scratch[NW * 16 * kWave];  // reporting path only: one 32 x 32 accumulator block per wave

  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid / kWave), ln = tid % kWave;
  const int r = ln & 31, hh = ln >> 5;
  const int panel = blockIdx.x % a.n_panels;
  const int qb = a.n_qblocks - 1 - (int)(blockIdx.x / a.n_panels);
  const bool stored = a.q_slot_base >= 0;
  const int64_t B0 = a.qblock0 + (int64_t)qb * QB;  // first query slot of this block
  const int64_t qs0 = stored ? a.q_slot_base : 0;            // slot of query row 0

  // panel p = candidate tiles p, p + P, p + 2P, ...: interleaved, so that the triangle of a stored batch (tiles above
  // the query block are skipped: the block that owns them reports the mirrored pairs) is cut evenly over the panels,
  // hence over the XCDs
  const int n_parts = max(a.n_parts, 1);  // (a zero-initialised argument block must not make the tile loop stand still)
  const int t_lo = a.part + n_parts * panel, t_step = n_parts * a.n_panels;
  int t_hi = a.n_ctiles;
  if (stored) t_hi = min(t_hi, (int)((B0 + QB) / CT));
  if (t_lo >= t_hi) return;

  // ---- A fragments: lane (r, hh) of block m holds W[slot][16 kk + 8 hh .. + 8) = chunk 2 kk + hh of its row; the 32
  // lanes of a half read 32 consecutive rows of one chunk: 512 contiguous bytes ----
  frag_t af[2][KS];
  const int64_t wslot0 = B0 + 64 * wv;  // a multiple of 64: the wave's 64 slots are one tile of Wq
  bool wave_live = false;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int64_t s = wslot0 + 32 * m + r;
    const int64_t row = s - qs0;
    const bool ok = row >= 0 && row < a.nq && s < a.wq_rows;
    wave_live |= ok;
    const uint4 *src = reinterpret_cast<const uint4 *>(a.Wq) + ((ok ? wslot0 : 0) / kHeadCTile) * (int64_t)(a.kt * (I8 ? 1 : 2) / 16 * kHeadCTile) +
                       (int64_t)a.chunk0 * kHeadCTile + 32 * m + r;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (ok) v = src[(2 * kk + hh) * kHeadCTile];
      af[m][kk] = __builtin_bit_cast(frag_t, v);
    }
  }
  wave_live = __any(wave_live);  // a wave without a query row copies tiles but computes nothing

  // ---- tile copy by LDS-DMA: a tile is TILEB contiguous bytes in HBM and lands in LDS as it is; wave w moves the
  // 1-KiB pieces (= chunks) w * PPW .. , lane l the 16 bytes of row l ----
  auto copy_tile = [&](const int t, const int buf) {
    const unsigned char *tsrc = reinterpret_cast<const unsigned char *>(a.Wc) + (int64_t)t * ((int64_t)kHeadCTile * a.kt * (I8 ? 1 : 2)) +
                                (int64_t)a.chunk0 * (kHeadCTile * 16) + ln * 16;
#pragma unroll
    for (int p = 0; p < PPW; ++p) {
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void *)(tsrc + (wv * PPW + p) * 1024),
          (__attribute__((address_space(3))) void *)(ldsb + buf * TILEB + (wv * PPW + p) * 1024), 16, 0, 0);
    }
  };
  // fragment reads: lane (r, hh) of column block n reads row 32 n + r of chunk 2 kk + hh: one address per lane, the
  // k-step and the column block are immediate offsets
  const uint32_t rd_lane = (uint32_t)(hh * 1024 + r * 16);

  unsigned long long n_pos = 0;  // positive elements seen by this lane
  elem_t *const sc = scratch + wv * (16 * kWave);
  const elem_t thr = MM::thr(a);

  // ---- the epilogue of one 64 x 32 half (column block n of a tile): nothing of D is stored.  `scan` folds two of its
  // 32 accumulators per call into the running maximum and the count of positive elements; it is called between the
  // MFMAs of the OTHER half, whose matrix-core time hides it.  `finish` closes the half: add the count (an element
  // strictly below a stored query block stands for (q, c) and, when c is a query of the batch too, for (c, q)) and,
  // when some element reached the threshold, report.
  struct Half {
    elem_t mx;
    uint32_t pos;
  };
  auto scan = [&](Half &hf, const acc_t (&ac)[2], const int step) {  // step 0..15: elements 2 step, 2 step + 1 of 32
#ifdef APSS_GEMM_NOEPI  // (microbenchmark experiment only: the tile stream without its epilogue; results are garbage)
    if (step != 0) return;
#endif
    // two accumulator elements per call.  The block's weights are non-negative, so is every accumulator, and "positive" is "bit
    // pattern != 0" in either rendering: min(bits, 1) twice and ONE three-operand add count the pair in 1.5 vector instructions
    // per element (compare + add-with-carry: 2), v_max3 takes the pair's maximum in one.  (Round 4, SQ counters on power-law C5:
    // with the INT8 rendering a half carries half the MFMAs over the same 32 accumulators and the epilogue's 5.9 vector
    // instructions per MFMA had become what the matrix pipe waits for.  Counting per WAVE on the scalar unit -- ballot,
    // s_bcnt1, s_add -- measured slower again: 88 -> 104 ms on C3-Zipf(1), as in round 2.)
    const elem_t v0 = ac[(2 * step) >> 4][(2 * step) & 15], v1 = ac[(2 * step + 1) >> 4][(2 * step + 1) & 15];
    hf.mx = MM::vmax3(hf.mx, v0, v1);
    if constexpr (COUNT && !I8) {
      hf.pos += (v0 > (elem_t)0 ? 1u : 0u) + (v1 > (elem_t)0 ? 1u : 0u);  // (bf16 form: twice the MFMAs per element, the compare form is no slower)
    } else if constexpr (COUNT) {
      // (spelled in assembly: the compiler folds min(x, 1) back into compare + conditional add)
      uint32_t t0, t1;
      asm("v_min_u32 %0, 1, %1" : "=v"(t0) : "v"(__builtin_bit_cast(uint32_t, v0)));
      asm("v_min_u32 %0, 1, %1" : "=v"(t1) : "v"(__builtin_bit_cast(uint32_t, v1)));
      asm("v_add3_u32 %0, %1, %2, %3" : "=v"(hf.pos) : "v"(t0), "v"(t1), "v"(hf.pos));
    }
  };
  auto finish = [&](Half &hf, const acc_t (&ac)[2], const int64_t cb_row0) {  // cb_row0: the half's first candidate row
    const bool below = stored && cb_row0 + 32 <= B0;
    const int64_t c = cb_row0 + r;
    n_pos += (below && c >= qs0) ? 2u * hf.pos : hf.pos;
    if (__any(hf.mx >= thr)) {
      // rare (a tile holding a near-duplicate, or the block's own diagonal): the accumulators go through a wave-private
      // LDS scratch one 32 x 32 block at a time, so that the reporting loop is a real loop
      const int64_t cext = c < a.n_rows ? a.c_ext[c] : 0;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sc[i * kWave + ln] = ac[m][i];
#pragma unroll 1
        for (int i = 0; i < 16; ++i) {
          const elem_t v = sc[i * kWave + ln];
          if (!__any(v >= thr)) continue;
          const int64_t s = wslot0 + 32 * m + (i & 3) + 8 * (i >> 2) + 4 * hh;
          const int64_t qrow = s - qs0;
          bool ok = v >= thr && qrow >= 0 && qrow < a.nq && c < a.n_rows;
          if (ok) ok = a.q_ext[qrow] != cext;  // self-exclusion by external id (IWA:91)
          const uint64_t o = wave_append(ok, &a.counters[kCtrResults]);
          if (ok && o < a.res_cap) {
            a.res_q[o] = (int32_t)qrow;
            a.res_c[o] = (int32_t)c;
            a.res_s[o] = MM::score(a, v);  // filter score; k_rescore replaces it
          }
          const bool ok2 = ok && below && c >= qs0;  // the mirrored pair: c as the query, q's slot as the candidate
          const uint64_t o2 = wave_append(ok2, &a.counters[kCtrResults]);
          if (ok2 && o2 < a.res_cap) {
            a.res_q[o2] = (int32_t)(c - qs0);
            a.res_c[o2] = (int32_t)s;
            a.res_s[o2] = MM::score(a, v);
          }
        }
      }
    }
    hf.mx = (elem_t)0;
    hf.pos = 0;
  };
  // the 32 MFMAs of one half: D[64 slots][32 candidates of column block n] over the whole K.  The B fragments come
  // from LDS PF k-steps ahead of their MFMAs (a read that an MFMA waits for exposes its whole latency);
  // `between(kk)` runs after the two MFMAs of every k-step
  auto ldfrag = [&](const unsigned char *tb, const int n, const int kk) {
    // column block n of the LDS tile: 64-row sub-tile n / 2 (SUBB bytes each), rows 32 (n % 2) .. of it
    return __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4 *>(tb + rd_lane + kk * 2048 + (n >> 1) * SUBB + (n & 1) * 512));
  };
  auto mma_half = [&](acc_t (&ac)[2], const unsigned char *tb, const int n, auto &&between) {
    frag_t bf[PF];
#pragma unroll
    for (int j = 0; j < PF && j < KS; ++j) bf[j] = ldfrag(tb, n, j);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const frag_t b = bf[kk % PF];
      if (kk + PF < KS) bf[kk % PF] = ldfrag(tb, n, kk + PF);
      if (kk == 0) {
        const acc_t zero = {};
        ac[0] = MM::mma(af[0][kk], b, zero);
        ac[1] = MM::mma(af[1][kk], b, zero);
      } else {
        ac[0] = MM::mma(af[0][kk], b, ac[0]);
        ac[1] = MM::mma(af[1][kk], b, ac[1]);
      }
      between(kk);
    }
  };

  // Software pipeline by halves, no second accumulator set: the 32-candidate column blocks of a tile alternate between
  // two accumulator pairs; while the matrix cores work on block b the vector unit scans block b - 1 (the last block of
  // tile t - 1 when b = 0).
  acc_t acc0[2], acc1[2];
  Half h0{(elem_t)0, 0u}, h1{(elem_t)0, 0u};
  int64_t pend = -1;  // first candidate row of the column block whose scan is pending in acc1 / h1 (-1: none)
  if constexpr (PIPE3) {
    // THREE tile buffers and the workgroup barrier in the MIDDLE of a tile.  With the barrier at the tile boundary every
    // wave of the workgroup -- both waves of every SIMD -- stopped at the same point with no MFMA left to issue and, once
    // released, first had to wait for its B fragments from LDS: the matrix pipe idled through arrival skew + an LDS round
    // trip per tile (SQ counters: MFMA busy 0.67 of the kernel's cycles at a measured in-kernel clock of 2.39 GHz, i.e. no
    // DVFS give-back to blame).  Here a wave runs from the second half of tile t straight into the first half of tile
    // t + 1: tile t + 1 has been complete in LDS since the barrier in the middle of tile t, so its first fragments are
    // requested during tile t's last MFMAs (a rolling prefetch queue across halves and tiles), and the barrier a wave meets
    // in mid-tile finds the fragments of the MFMAs behind it already in registers.  That barrier says "tile t + 1 has
    // landed for everyone, and everyone is past tile t - 1"; the copy of tile t + 2 into tile t - 1's buffer follows it.
    frag_t bq[PF];
    auto mma_run = [&](acc_t (&ac)[2], const unsigned char *tb, const int n, const unsigned char *tb_next, const int n_next,
                       auto &&between) {
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        const frag_t b = bq[kk % PF];
        bq[kk % PF] = kk + PF < KS ? ldfrag(tb, n, kk + PF) : ldfrag(tb_next, n_next, kk + PF - KS);
        if (kk == 0) {
          const acc_t zero = {};
          ac[0] = MM::mma(af[0][kk], b, zero);
          ac[1] = MM::mma(af[1][kk], b, zero);
        } else {
          ac[0] = MM::mma(af[0][kk], b, ac[0]);
          ac[1] = MM::mma(af[1][kk], b, ac[1]);
        }
        between(kk);
      }
    };
    static_assert(KS % PF == 0 && KS >= PF, "the prefetch queue's slots line up across halves");
    // prologue: the first NBUF - 1 tiles are requested; the first must have landed (loads complete in issue order)
    copy_tile(t_lo, 0);
    int ahead = 0;  // tiles requested beyond the current one
#pragma unroll
    for (int j = 1; j < NBUF - 1; ++j)
      if (t_lo + j * t_step < t_hi) {
        copy_tile(t_lo + j * t_step, j);
        ++ahead;
      }
    if (ahead == NBUF - 2 && NBUF > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int bi = 0;  // buffer of the current tile
    auto buf_of = [&](const int k) { return ldsb + ((bi + k) % NBUF) * TILEB; };
    const unsigned char *tb = buf_of(0), *tbn = buf_of(1);
    if (wave_live) {
#pragma unroll
      for (int j = 0; j < PF; ++j) bq[j] = ldfrag(tb, 0, j);
    }
    for (int t = t_lo; t < t_hi; t += t_step) {
      const int64_t t_row0 = (int64_t)t * CT;
      if (wave_live) {
        if (pend >= 0) {
          mma_run(acc0, tb, 0, tb, 1, [&](const int kk) {
#pragma unroll
            for (int j = 0; j < SPK; ++j) scan(h1, acc1, kk * SPK + j);
          });
          finish(h1, acc1, pend);
        } else {
          mma_run(acc0, tb, 0, tb, 1, [&](const int) {});
        }
      }
      // this wave's pieces of tile t + 1 have landed (the tiles requested after it may still be in flight: counted wait)
      if (NBUF > 3 && t + (NBUF - 2) * t_step < t_hi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF > 3 ? NBUF - 3 : 0) * PPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                   // ... and everyone's; nobody reads tile t - 1 any more
      if (t + (NBUF - 1) * t_step < t_hi) copy_tile(t + (NBUF - 1) * t_step, (bi + NBUF - 1) % NBUF);  // into tile t - 1's buffer
      if (wave_live) {
        mma_run(acc1, tb, 1, tbn, 0, [&](const int kk) {
#pragma unroll
          for (int j = 0; j < SPK; ++j) scan(h0, acc0, kk * SPK + j);
        });
        finish(h0, acc0, t_row0);
        pend = t_row0 + 32;
      }
      bi = (bi + 1) % NBUF;
      tb = buf_of(0);
      tbn = buf_of(1);
    }
  } else {
  copy_tile(t_lo, 0);
  int buf = 0;
  for (int t = t_lo; t < t_hi; t += t_step, buf ^= 1) {    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile t have landed
    __syncthreads();                                   // ... and everyone's; the other buffer's readers are done
    if (t + t_step < t_hi) copy_tile(t + t_step, buf ^ 1);
    if (!wave_live) continue;
    const unsigned char *tb = ldsb + buf * TILEB;
    const int64_t t_row0 = (int64_t)t * CT;
#pragma unroll
    for (int b = 0; b < NB; b += 2) {
      if (b > 0 || pend >= 0) {
        mma_half(acc0, tb, b, [&](const int kk) {
#pragma unroll
          for (int j = 0; j < SPK; ++j) scan(h1, acc1, kk * SPK + j);
        });
        finish(h1, acc1, b > 0 ? t_row0 + 32 * (b - 1) : pend);
      } else {
        mma_half(acc0, tb, b, [&](const int) {});
      }
      mma_half(acc1, tb, b + 1, [&](const int kk) {
#pragma unroll
        for (int j = 0; j < SPK; ++j) scan(h0, acc0, kk * SPK + j);
      });
      finish(h0, acc0, t_row0 + 32 * b);
    }
    pend = t_row0 + 32 * (NB - 1);
  }
  }
  if (wave_live && pend >= 0) {
#pragma unroll
    for (int st = 0; st < 16; ++st) scan(h1, acc1, st);
    finish(h1, acc1, pend);
  }
  // positive elements seen by this wave -> one atomic
  unsigned long long tot = n_pos;
  for (int o = kWave / 2; o; o >>= 1) tot += __shfl_xor(tot, o);
  if (ln == 0 && tot) atomicAdd(a.head_pairs, tot);
}

// ---------------------------------------------------------------------------------------------------------
// k_head_gemv: the same filter for a query batch too small to fill 32-row MFMA blocks (single-vector messages of the
// latency path, benchmark/LoadGenerator.scala:58-74): HBM-bound sweep of W.  One wave per candidate tile, lane = row:
// every load is one contiguous KiB (a chunk of the tile), every lane keeps its own row's dot products with the (<= 8
// per pass) query vectors, which sit in LDS as fp32.
struct HeadGemvArgs {
  const uint16_t *Wq;
  const uint16_t *Wc;
  int64_t n_rows;
  int64_t q_slot_base;
  int32_t nq;
  int32_t kh;             // width of the block this launch multiplies (<= 256)
  int32_t kt, chunk0;     // total width of a W row, first chunk of the block (as in HeadGemmArgs)
  int32_t part, n_parts;  // as in HeadGemmArgs
  const int64_t *q_ext, *c_ext;
  float thr;
  int32_t *res_q, *res_c;
  float *res_s;
  uint64_t res_cap;
  unsigned long long *counters;
  unsigned long long *head_pairs;
  int32_t i8;             // W is the INT8 rendering: sums are in units of 1 / S^2 (exact in fp32: < 2^24), thr is given in those units
  float inv_s2;           // ... and a reported score = sum * inv_s2
};

constexpr int kGemvQ = 8;  // queries per pass over W

__global__ __launch_bounds__(256) void k_head_gemv(const HeadGemvArgs a) {
  __shared__ float qv[kGemvQ][256];
  __shared__ unsigned long long npos;
  const int tid = threadIdx.x, ln = tid % kWave;
  const int64_t qs0 = a.q_slot_base >= 0 ? a.q_slot_base : 0;
  const int cpr = a.i8 ? a.kh / 16 : a.kh / 8;       // 16-B chunks of the block this launch multiplies
  const int cpr_row = a.i8 ? a.kt / 16 : a.kt / 8;   // ... of a whole W row
  const int64_t n_tiles = (a.n_rows + kHeadCTile - 1) / kHeadCTile;
  if (tid == 0) npos = 0;
  unsigned long long my_pos = 0;
  for (int q0 = 0; q0 < a.nq; q0 += kGemvQ) {
    const int nqq = min(kGemvQ, a.nq - q0);
    __syncthreads();
    for (int i = tid; i < nqq * a.kh; i += blockDim.x) {
      const int qq = i / a.kh, k = i % a.kh;
      if (a.i8) {
        const unsigned char *wq = reinterpret_cast<const unsigned char *>(a.Wq);
        qv[qq][k] = (float)wq[head_chunk_byte_off(qs0 + q0 + qq, a.chunk0 + (k >> 4), cpr_row) + (k & 15)];
      } else {
        const uint16_t b = a.Wq[head_chunk_off(qs0 + q0 + qq, a.chunk0 + (k >> 3), a.kt) + (k & 7)];
        qv[qq][k] = __uint_as_float((uint32_t)b << 16);
      }
    }
    __syncthreads();
    for (int64_t tl = (int64_t)blockIdx.x * 4 + tid / kWave; a.part + tl * max(a.n_parts, 1) < n_tiles; tl += (int64_t)gridDim.x * 4) {
      const int64_t t = a.part + tl * max(a.n_parts, 1);
      const int64_t c = t * kHeadCTile + ln;
      float s[kGemvQ];
#pragma unroll
      for (int qq = 0; qq < kGemvQ; ++qq) s[qq] = 0.f;
      const uint4 *tp = reinterpret_cast<const uint4 *>(a.Wc) + t * (int64_t)(cpr_row * kHeadCTile) + (int64_t)a.chunk0 * kHeadCTile + ln;
      for (int ch = 0; ch < cpr; ++ch) {
        const uint4 v = tp[ch * kHeadCTile];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        if (a.i8) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
              const float x = (float)((w[e] >> (8 * b)) & 0xffu);
#pragma unroll
              for (int qq = 0; qq < kGemvQ; ++qq)
                if (qq < nqq) s[qq] += x * qv[qq][ch * 16 + 4 * e + b];
            }
          }
          continue;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
#pragma unroll
          for (int qq = 0; qq < kGemvQ; ++qq)
            if (qq < nqq) s[qq] += lo * qv[qq][ch * 8 + 2 * e] + hi * qv[qq][ch * 8 + 2 * e + 1];
        }
      }
#pragma unroll
      for (int qq = 0; qq < kGemvQ; ++qq) {
        if (qq >= nqq) break;
        const bool in = c < a.n_rows;
        if (in && s[qq] > 0.f) my_pos++;
        bool ok = in && s[qq] >= a.thr;
        if (ok) ok = a.q_ext[q0 + qq] != a.c_ext[c];
        const uint64_t o = wave_append(ok, &a.counters[kCtrResults]);
        if (ok && o < a.res_cap) {
          a.res_q[o] = q0 + qq;
          a.res_c[o] = (int32_t)c;
          a.res_s[o] = a.i8 ? s[qq] * a.inv_s2 : s[qq];
        }
      }
    }
  }
  if (my_pos) atomicAdd(&npos, my_pos);
  __syncthreads();
  if (tid == 0 && npos) atomicAdd(a.head_pairs, npos);
}

// ---------------------------------------------------------------------------------------------------------
// k_pair_dedup: the two filters report a pair that passes both tests twice; keep one.  Open-addressing hash set of
// 64-bit keys (q << 32 | c) in global memory, one atomicCAS per probe; the table holds >= 2x the pairs, so a probe
// sequence always ends at an empty slot.
__global__ void k_pair_dedup(const int32_t *in_q, const int32_t *in_c, const float *in_s, int64_t n,
                             unsigned long long *table, uint64_t mask, int32_t *out_q, int32_t *out_c, float *out_s,
                             unsigned long long *out_count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool keep = false;
  int32_t q = 0, c = 0;
  if (i < n) {
    q = in_q[i];
    c = in_c[i];
    const unsigned long long key = ((unsigned long long)(uint32_t)q << 32) | (uint32_t)c;  // never ~0: q < 2^31
    unsigned long long x = key * 0x9E3779B97F4A7C15ull;
    x ^= x >> 29;
    uint64_t hpos = x & mask;
    for (uint64_t step = 0; step <= mask; ++step) {
      const unsigned long long old = atomicCAS(&table[hpos], ~0ull, key);
      if (old == ~0ull) {
        keep = true;
        break;
      }
      if (old == key) break;
      hpos = (hpos + 1) & mask;
    }
  }
  const uint64_t o = wave_append(keep, out_count);
  if (keep) {
    out_q[o] = q;
    out_c[o] = c;
    out_s[o] = in_s[i];
  }
}

}  // namespace apss
