// apss_hip.hip -- C ABI (include/apss.h) over the HIP kernels of apss_kernels.hpp.
//
// One handle == one IndexingWorkerActor's state (vectorsStore + invertedIndex + similarityThreshold,
// IndexingWorkerActor.scala:21-25) resident in the HBM of one MI355X.  No CPU compute path exists here: every
// entry point either runs the HIP kernels or fails with APSS_E_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/apss.h"
#include "apss_kernels.hpp"

using namespace apss;

namespace {

thread_local std::string g_create_error;

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;  // elements
};

}  // namespace

struct apss_handle {
  apss_config cfg{};
  int dev = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::string err;
  bool sharded = false;
  bool nonneg = true;  // every stored / queried weight so far is >= 0
  int64_t store_max_nnz = 0, q_max_nnz = 0;  // longest row of the store / of the staged query batch
  float store_max_norm2 = 0.f, q_max_norm2 = 0.f;  // largest squared row norm (bounds every partial score)
  int64_t store_nonempty = 0;  // stored rows with at least one indexed entry (each touches itself in a self-join)
  int64_t last_batch_nonempty = 0;
  int32_t cb = 16384;

  // store (CSR) -- vectorsStore, IWA:22
  int64_t n_rows = 0, nnz = 0;
  DevBuf<int64_t> rowptr, ext;
  DevBuf<int32_t> idx;
  DevBuf<uint32_t> erow;  // store row of every entry
  DevBuf<float> val, sub;  // sub: shard sub-norm per row (sharded only)
  // index (tile-major CSC) -- invertedIndex, IWA:25.  Two renderings of the same posting lists:
  //   ex: exact, 8-B postings, `cb` rows per tile      (k_probe_wave, k_probe, shard mode)
  //   cx: coarse, 4-B postings, up to 2*cb rows per tile (k_probe_coarse, the filter of the two-pass join)
  struct IndexSet {
    int32_t cb = 0;
    int32_t align = 0;
    bool coarse = false;
    int64_t n_tiles = 0, post_used = 0;
    DevBuf<uint2> seg;
    DevBuf<Posting> post;
    DevBuf<uint32_t> post_c;
    DevBuf<int64_t> base, total;
    std::vector<int64_t> h_base;
    double build_ms = 0;
  };
  IndexSet ex, cx;
  bool use_coarse = false;  // build and use the coarse index (non-sharded handles without APSS_FLAG_EXACT_ACCUM)
  int64_t n_tiles = 0;      // tiles of the exact rendering (apss_stats)
  int64_t ex_built_rows = 0;     // the exact rendering covers rows [0, ex_built_rows)
  DevBuf<float> tile_min, tile_min_c;  // shard mode: min positive sub-norm per exact / coarse tile
  // query staging (apss_query: batch not stored)
  DevBuf<int64_t> q_rowptr, q_ext;
  DevBuf<int32_t> q_idx;
  DevBuf<float> q_val, q_sub;
  // ingest scratch
  DevBuf<int64_t> s_keep, s_cnt, s_rowdst, s_nnzdst, in_rowptr, in_ext;
  DevBuf<int32_t> in_idx;
  DevBuf<float> s_inv, s_sub, in_val;
  DevBuf<double> in_val64;
  DevBuf<int32_t> vq_first, vrow_q;  // virtual-row table of the last query batch (queries of > 512 terms)
  DevBuf<int64_t> vrow_ptr, vrow_np, vrow_first;
  // results of the last query-type call
  DevBuf<int32_t> res_q, res_c, fin_q, fin_c;
  DevBuf<float> res_s, fin_s;
  const int32_t *out_q = nullptr, *out_c = nullptr;  // where the last call's results live (res_* or fin_*)
  const float *out_s = nullptr;
  int64_t n_res = -1;
  const int64_t *res_q_ext = nullptr;      // ext ids of the last query batch (device)
  const int64_t *last_q_rowptr = nullptr;  // last query batch CSR (device), for apss_partial_scores_dev
  const int32_t *last_q_idx = nullptr;
  const float *last_q_val = nullptr;
  int64_t last_nq = 0;
  DevBuf<unsigned long long> counters, dbg;
  DevBuf<unsigned int> flagword;
  // stats
  apss_stats st{};
  size_t bytes_reserved = 0;
};

namespace {

#define HIPCHK(h, expr)                                                                              \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess) {                                                                          \
      (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                                  \
      return e_ == hipErrorOutOfMemory ? APSS_E_NOMEM : APSS_E_DEVICE;                               \
    }                                                                                                \
  } while (0)

#define APSS_TRY(expr)          \
  do {                          \
    int32_t rc_ = (expr);       \
    if (rc_ != APSS_OK) return rc_; \
  } while (0)

int32_t fail(apss_handle *h, int32_t rc, const std::string &msg) {
  h->err = msg;
  return rc;
}

// grow-only device array; keeps the first `keep` elements
template <class T>
int32_t ensure(apss_handle *h, DevBuf<T> &b, size_t n, size_t keep = 0, bool exact = false) {
  if (n <= b.cap && b.p) return APSS_OK;
  size_t ncap = exact ? n : std::max(n, b.cap + b.cap / 2);
  ncap = std::max<size_t>(ncap, 64);
  T *np = nullptr;
  HIPCHK(h, hipMalloc((void **)&np, ncap * sizeof(T)));
  if (keep && b.p) HIPCHK(h, hipMemcpyAsync(np, b.p, keep * sizeof(T), hipMemcpyDeviceToDevice, h->stream));
  if (b.p) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipFree(b.p));
    h->bytes_reserved -= b.cap * sizeof(T);
  }
  b.p = np;
  b.cap = ncap;
  h->bytes_reserved += ncap * sizeof(T);
  return APSS_OK;
}

template <class T>
void release(DevBuf<T> &b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

int32_t enter(apss_handle *h) {
  if (!h) return APSS_E_INVALID;
  hipError_t e = hipSetDevice(h->dev);
  if (e != hipSuccess) return fail(h, APSS_E_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
  return APSS_OK;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- ingest: validate (+ optional normalise / admission / value prune / term-range filter) and append ----
// Source arrays are device pointers (batch-relative rowptr).  Destination: the store (to_store) or the query
// staging buffers.  *n_out / *nnz_out: rows / entries that survived.
int32_t ingest(apss_handle *h, int64_t n, int64_t nnz, const int64_t *d_rowptr, const int32_t *d_idx,
               const float *d_val, const int64_t *d_ext, bool to_store, int64_t *n_out, int64_t *nnz_out) {
  *n_out = 0;
  *nnz_out = 0;
  if (n == 0) return APSS_OK;
  const bool transform = h->sharded || (h->cfg.flags & (APSS_FLAG_VALUE_PRUNE | APSS_FLAG_ADMISSION | APSS_FLAG_NORMALIZE));
  APSS_TRY(ensure(h, h->s_keep, (size_t)n + 1));
  APSS_TRY(ensure(h, h->s_cnt, (size_t)n + 1));
  APSS_TRY(ensure(h, h->s_inv, (size_t)n));
  APSS_TRY(ensure(h, h->s_sub, (size_t)n));
  APSS_TRY(ensure(h, h->flagword, 4));
  HIPCHK(h, hipMemsetAsync(h->flagword.p, 0, 4 * sizeof(unsigned int), h->stream));

  IngestArgs a{};
  a.n = n;
  a.nnz = nnz;
  a.rowptr = d_rowptr;
  a.idx = d_idx;
  a.val = d_val;
  a.dim = h->cfg.dim;
  a.term_lo = h->cfg.term_lo;
  a.term_hi = h->cfg.term_hi;
  a.flags = h->cfg.flags;
  a.theta = (float)h->cfg.theta;
  a.index_threshold = (float)h->cfg.index_threshold;
  a.row_keep = h->s_keep.p;
  a.row_cnt = h->s_cnt.p;
  a.row_inv = h->s_inv.p;
  a.row_sub = h->s_sub.p;
  a.flags_out = h->flagword.p;
  const int threads = 256;
  const int64_t blocks = ceil_div(n * kGroup, threads);
  // (the workgroups loop over their rows: few workgroups = few same-address atomics on the batch summaries)
  hipLaunchKernelGGL(k_ingest_count<kGroup>, dim3((unsigned)std::min<int64_t>(blocks, 4096)), dim3(threads), 0, h->stream, a);
  HIPCHK(h, hipGetLastError());

  // destination
  int64_t dst_row0 = to_store ? h->n_rows : 0, dst_nnz0 = to_store ? h->nnz : 0;
  int64_t kept_rows = n, kept_nnz = nnz;
  unsigned int flags_host[4] = {0, 0, 0, 0};
  if (transform) {
    APSS_TRY(ensure(h, h->s_rowdst, (size_t)n + 1));
    APSS_TRY(ensure(h, h->s_nnzdst, (size_t)n + 1));
    hipLaunchKernelGGL(k_scan_i64, dim3(1), dim3(1024), 0, h->stream, (const int64_t *)h->s_keep.p, h->s_rowdst.p, n);
    hipLaunchKernelGGL(k_scan_i64, dim3(1), dim3(1024), 0, h->stream, (const int64_t *)h->s_cnt.p, h->s_nnzdst.p, n);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(&kept_rows, h->s_rowdst.p + n, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(&kept_nnz, h->s_nnzdst.p + n, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
  }
  HIPCHK(h, hipMemcpyAsync(flags_host, h->flagword.p, 4 * sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (flags_host[0] & 1u)
    return fail(h, APSS_E_INVALID, "malformed vector: indices must be strictly increasing and in [0, dim) "
                                   "(SparseVector.scala:75; vectorDim mismatch is the require of CommonUtils.scala:99)");
  if (flags_host[0] & 2u) return fail(h, APSS_E_INVALID, "non-finite value in a vector");
  if (flags_host[0] & 4u) h->nonneg = false;
  float norm2;
  std::memcpy(&norm2, &flags_host[2], sizeof(float));
  if (to_store) {
    h->store_max_nnz = std::max<int64_t>(h->store_max_nnz, flags_host[1]);
    h->store_max_norm2 = std::max(h->store_max_norm2, norm2);
    h->store_nonempty += flags_host[3];
    h->last_batch_nonempty = flags_host[3];
  } else {
    h->q_max_nnz = flags_host[1];
    h->q_max_norm2 = norm2;
  }

  DevBuf<int64_t> &o_rowptr = to_store ? h->rowptr : h->q_rowptr;
  DevBuf<int64_t> &o_ext = to_store ? h->ext : h->q_ext;
  DevBuf<int32_t> &o_idx = to_store ? h->idx : h->q_idx;
  DevBuf<float> &o_val = to_store ? h->val : h->q_val;
  DevBuf<float> &o_sub = to_store ? h->sub : h->q_sub;
  APSS_TRY(ensure(h, o_rowptr, (size_t)(dst_row0 + kept_rows + 1), (size_t)(to_store ? dst_row0 + 1 : 0)));
  APSS_TRY(ensure(h, o_ext, (size_t)(dst_row0 + kept_rows), (size_t)dst_row0));
  APSS_TRY(ensure(h, o_idx, (size_t)(dst_nnz0 + kept_nnz), (size_t)dst_nnz0));
  APSS_TRY(ensure(h, o_val, (size_t)(dst_nnz0 + kept_nnz), (size_t)dst_nnz0));
  if (to_store) APSS_TRY(ensure(h, h->erow, (size_t)(dst_nnz0 + kept_nnz), (size_t)dst_nnz0));
  if (h->sharded) APSS_TRY(ensure(h, o_sub, (size_t)(dst_row0 + kept_rows), (size_t)dst_row0));
  if (dst_row0 == 0) HIPCHK(h, hipMemsetAsync(o_rowptr.p, 0, sizeof(int64_t), h->stream));

  IngestWriteArgs w{};
  w.in = a;
  w.dst_row0 = dst_row0;
  w.dst_nnz0 = dst_nnz0;
  w.o_rowptr = o_rowptr.p;
  w.o_idx = o_idx.p;
  w.o_val = o_val.p;
  w.o_ext = o_ext.p;
  w.o_sub = h->sharded ? o_sub.p : nullptr;
  w.o_erow = to_store ? h->erow.p : nullptr;
  w.ext = d_ext;
  if (transform) {
    w.row_dst = h->s_rowdst.p;
    w.nnz_dst = h->s_nnzdst.p;
  } else {
    // identity placement: row r -> r, entry k -> k (the batch rowptr is the entry scan)
    w.row_dst = nullptr;
    w.nnz_dst = d_rowptr;
  }
  hipLaunchKernelGGL(k_ingest_write, dim3((unsigned)blocks), dim3(threads), 0, h->stream, w);
  HIPCHK(h, hipGetLastError());
  *n_out = kept_rows;
  *nnz_out = kept_nnz;
  return APSS_OK;
}

// ---- index build for rows [row0, n_rows): rebuild every tile of `ix` that contains one of them ----
int32_t build_tiles(apss_handle *h, apss_handle::IndexSet &ix, int64_t row0) {
  const int64_t cb = ix.cb;
  const int64_t tile0 = row0 / cb;
  const int64_t n_tiles = ceil_div(h->n_rows, cb);
  const int64_t stride = (int64_t)h->cfg.dim;
  ix.n_tiles = n_tiles;
  if (n_tiles == tile0) return APSS_OK;
  APSS_TRY(ensure(h, ix.seg, (size_t)(n_tiles * stride), (size_t)(tile0 * stride)));
  APSS_TRY(ensure(h, ix.base, (size_t)n_tiles + 1, (size_t)tile0 + 1));
  APSS_TRY(ensure(h, ix.total, (size_t)n_tiles, 0));
  DevBuf<float> &tmin = ix.coarse ? h->tile_min_c : h->tile_min;
  if (h->sharded) APSS_TRY(ensure(h, tmin, (size_t)n_tiles, (size_t)tile0));
  if (ix.h_base.empty()) ix.h_base.push_back(0);
  ix.h_base.resize((size_t)tile0 + 1);  // bases of the tiles that stay
  const int64_t r0 = tile0 * cb;
  // small dims: counters and cursors of a (tile, term range) live in one workgroup's LDS (no global atomics)
  const int32_t n_ranges = (int32_t)ceil_div(h->cfg.dim, kBuildRange);
  // (one workgroup per (tile, range): with fewer than ~48 of them -- a small batch, a rebuilt tail tile, a 1/8 candidate
  // range -- the row-parallel global-atomic kernels are faster; APSS_BUILD_LDS / APSS_BUILD_ATOMIC force either)
  const bool lds_build = n_ranges <= kBuildMaxRanges && !getenv("APSS_BUILD_ATOMIC") &&
                         ((n_tiles - tile0) * n_ranges >= 48 || getenv("APSS_BUILD_LDS"));
  if (!lds_build)
    HIPCHK(h, hipMemsetAsync(ix.seg.p + tile0 * stride, 0, (size_t)((n_tiles - tile0) * stride) * sizeof(uint2), h->stream));
  BuildArgs b{};
  b.rowptr = h->rowptr.p;
  b.idx = h->idx.p;
  b.val = h->val.p;
  b.row0 = r0;
  b.row1 = h->n_rows;
  b.cb = (int32_t)cb;
  b.dim = h->cfg.dim;
  b.tile_seg = ix.seg.p;
  b.seg_stride = stride;
  b.coarse = ix.coarse ? 1 : 0;
  b.seg_align = ix.align;
  b.erow = h->erow.p;
  b.coarse_shift = ix.coarse && ix.cb <= 32768 ? 1 : 0;  // must agree with k_probe_coarse's SLOT2 (the 512-thread kernels)
  const int threads = 256;
  const int64_t blocks = ceil_div((h->n_rows - r0) * kWave, threads);
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  const dim3 lds_grid((unsigned)((n_tiles - tile0) * n_ranges));
  if (lds_build) hipLaunchKernelGGL(k_tile_hist_lds, lds_grid, dim3(1024), 0, h->stream, b, tile0, n_ranges);
  else hipLaunchKernelGGL(k_tile_hist, dim3((unsigned)blocks), dim3(threads), 0, h->stream, b);
  hipLaunchKernelGGL(k_tile_scan, dim3((unsigned)(n_tiles - tile0)), dim3(1024), 0, h->stream, ix.seg.p, stride, h->cfg.dim,
                     tile0, ix.total.p, (uint32_t)ix.align, lds_build ? 1u : 0u);
  HIPCHK(h, hipGetLastError());
  // padded posting counts -> tile bases (host prefix sum: a handful of values), then reserve the posting array
  std::vector<int64_t> tot((size_t)(n_tiles - tile0));
  HIPCHK(h, hipMemcpyAsync(tot.data(), ix.total.p + tile0, tot.size() * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (size_t i = 0; i < tot.size(); ++i) ix.h_base.push_back(ix.h_base.back() + tot[i]);
  HIPCHK(h, hipMemcpyAsync(ix.base.p + tile0, ix.h_base.data() + tile0, (size_t)(n_tiles - tile0 + 1) * sizeof(int64_t),
                           hipMemcpyHostToDevice, h->stream));
  ix.post_used = ix.h_base.back();
  if (ix.coarse) APSS_TRY(ensure(h, ix.post_c, (size_t)ix.post_used + 64, (size_t)ix.h_base[(size_t)tile0]));
  else APSS_TRY(ensure(h, ix.post, (size_t)ix.post_used + 64, (size_t)ix.h_base[(size_t)tile0]));
  if (ix.coarse)  // the probe reads a zero word as "no posting": clear the padding between segments
    HIPCHK(h, hipMemsetAsync(ix.post_c.p + ix.h_base[(size_t)tile0], 0,
                             (size_t)(ix.post_used - ix.h_base[(size_t)tile0] + 64) * sizeof(uint32_t), h->stream));
  b.tile_post_base = ix.base.p;
  b.post = ix.post.p;
  b.post_c = ix.post_c.p;
  if (lds_build) hipLaunchKernelGGL(k_tile_scatter_lds, lds_grid, dim3(1024), 0, h->stream, b, tile0, n_ranges);
  else hipLaunchKernelGGL(k_tile_scatter, dim3((unsigned)blocks), dim3(threads), 0, h->stream, b);
  if (h->sharded)
    hipLaunchKernelGGL(k_tile_min_sub, dim3((unsigned)(n_tiles - tile0)), dim3(1024), 0, h->stream,
                       (const float *)h->sub.p, h->n_rows, (int32_t)cb, tmin.p, tile0);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  HIPCHK(h, hipEventSynchronize(h->ev1));
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  ix.build_ms = ms;
  return APSS_OK;
}

// Which renderings of the index a handle keeps: the coarse one (two-pass speed path) is built at insert time when the
// handle may use it; the exact one is built at insert time otherwise, or lazily by the first probe that needs it.
int32_t ensure_exact_index(apss_handle *h) {
  if (h->ex_built_rows < h->n_rows || h->ex.n_tiles == 0) {
    APSS_TRY(build_tiles(h, h->ex, std::min(h->ex_built_rows, h->n_rows)));
    h->ex_built_rows = h->n_rows;
    h->st.build_ms += h->ex.build_ms;
  }
  return APSS_OK;
}

int32_t build_index(apss_handle *h, int64_t row0) {
  h->st.build_ms = 0;
  if (h->use_coarse) {
    if (h->cx.n_tiles == 0 && h->cfg.tile_rows == 0 && !getenv("APSS_CX_TILE") && h->n_rows > 0) {
      // a round's cost is mostly fixed, so what matters is how many postings a (tile, term) segment holds:
      // rows_per_tile * nnz_per_row / dim.  Below ~16 at 32768 rows (C5 shape: 6.5) the 65536-row tile with one
      // 1024-thread workgroup per CU wins (C5 shape at N=2M: 647 vs 790 ms); at C3 (33) two workgroups per CU win.
      const double seg32 = 32768.0 * ((double)h->nnz / (double)h->n_rows) / (double)h->cfg.dim;
      h->cx.cb = seg32 < 16.0 && !h->sharded ? 65536 : 32768;  // (the 1024-thread kernel has no shard variant)
    }
    APSS_TRY(build_tiles(h, h->cx, row0));
    h->st.build_ms += h->cx.build_ms;
    h->ex_built_rows = std::min(h->ex_built_rows, row0 / h->ex.cb * h->ex.cb);  // exact tiles from here on are stale
  } else {
    h->ex_built_rows = std::min(h->ex_built_rows, row0);
    APSS_TRY(ensure_exact_index(h));
  }
  h->n_tiles = ceil_div(h->n_rows, h->ex.cb);
  return APSS_OK;
}

template <int MODE, bool FX>
int32_t launch_probe(apss_handle *h, const ProbeArgs &a, size_t lds) {
  auto kern = k_probe<MODE, kProbeBlock, FX>;
  HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)a.n_tiles * a.n_chunks)), dim3(kProbeBlock), lds, h->stream, a);
  HIPCHK(h, hipGetLastError());
  return APSS_OK;
}

// ---- probe the whole index with a query batch resident on the device ----
int32_t probe(apss_handle *h, int64_t nq, const int64_t *q_rowptr, const int32_t *q_idx, const float *q_val,
              const int64_t *q_ext, const float *q_sub, int64_t q_slot_base, int64_t q_max_nnz, float q_max_norm2,
              int64_t q_nnz_end, int64_t *n_results) {
  h->res_q_ext = q_ext;
  h->last_q_rowptr = q_rowptr;
  h->last_q_idx = q_idx;
  h->last_q_val = q_val;
  h->last_nq = nq;
  h->n_res = 0;
  h->st.posting_visits = h->st.candidate_pairs = h->st.result_pairs = 0;
  h->st.probe_ms = 0;
  h->st.probe_launches = 0;
  if (n_results) *n_results = 0;
  APSS_TRY(ensure(h, h->counters, kCtrCount));
  h->st.filter_survivors = 0;
  h->st.rescore_ms = 0;
  if (nq == 0 || h->n_rows == 0 || q_nnz_end <= 0 || h->nnz == 0) return APSS_OK;  // nothing can share a term
  if (nq > 0x7fffffffLL) return fail(h, APSS_E_INVALID, "query batch too large");

  const double theta = h->cfg.theta;
  int mode;
  if (!(theta > 0.0)) mode = 2;
  else if (!h->nonneg || (h->cfg.flags & APSS_FLAG_FORCE_SCAN)) mode = 1;
  else mode = 0;

  // every partial score must fit the fixed-point accumulators: by Cauchy-Schwarz a partial sum of non-negative
  // products is at most |q| * |c| <= the product of the largest row norms
  const double bound = std::sqrt((double)q_max_norm2) * std::sqrt((double)h->store_max_norm2) * 1.0001 + 1e-6;
  const double fx_scale = bound < 3.9 ? 1073741824.0 : (bound < 15.6 ? 268435456.0 : 0.0);
  const bool forced_general = (h->cfg.flags & APSS_FLAG_FORCE_GENERAL) || getenv("APSS_FORCE_GENERAL");

  // ---- path 1: two-pass join (coarse filter + exact rescoring) ----
  // accumulator units per 1.0: a 16-bit sum holds S * |q||c| (fp16 weights: + 2^-11) plus one unit per shared term
  // S = the largest power of two that fits (2^15 for unit-norm input); un-normalised input gets a smaller one as long as
  // the threshold in units stays well above the round-up bias (one unit per shared term), else the filter passes too much
  const double cx_shared = (double)std::min<int64_t>(q_max_nnz, h->store_max_nnz);
  const double cx_room = 65535.0 - cx_shared;
  double cx_scale = 0.0;
  for (int k = 15; k >= 4 && cx_scale == 0.0; --k)
    if (bound * 1.0005 * std::ldexp(1.0, k) < cx_room) cx_scale = std::ldexp(1.0, k);
  const double cx_theta = std::floor(theta * cx_scale * (1.0 - 1.0 / 2048 - 1e-6));
  const bool cx_selective = cx_scale >= 16384.0 || cx_theta >= 4.0 * cx_shared;
  const bool cx_fp16_ok = std::sqrt((double)h->store_max_norm2) < 60000.0;  // a weight never exceeds its row's norm
  // (shard mode scales the threshold down per query and tile: the kernel clamps it at 1, which only admits more)
  // weights of either sign with theta > 0 go through the filter too (it then sums positive products only); term shards
  // (and long queries over 65536-row tiles) have no such instantiation and keep the general kernel
  const bool cx_signed = mode == 1 && !(h->cfg.flags & APSS_FLAG_FORCE_SCAN) && !h->sharded &&
                         (h->cx.cb <= 32768 || q_max_nnz <= 512) &&
                         !getenv("APSS_CX_CHUNK8") && !getenv("APSS_CX_U3") && !getenv("APSS_CX_U4");
  const bool coarse_path = h->use_coarse && (mode == 0 || cx_signed) && cx_selective && cx_fp16_ok && !forced_general && nq < (1LL << 30) &&
                           !(h->sharded && h->cx.cb > 32768) &&
                           (q_max_nnz <= 512 || !h->sharded) &&
                           !getenv("APSS_EXACT_ACCUM") && cx_scale > 0 && cx_theta < 65000.0 && (h->sharded || cx_theta - 2 >= 1.0) &&
                           std::min(h->store_max_nnz * (int64_t)h->cx.cb, h->nnz) + (int64_t)kSegAlignC * h->cfg.dim < (1LL << 27);

  ProbeArgs a{};
  a.seg_stride = (int64_t)h->cfg.dim;
  a.ext_id = h->ext.p;
  a.c_scale = h->sharded ? h->sub.p : nullptr;
  a.n_rows = h->n_rows;
  a.q_rowptr = q_rowptr;
  a.q_idx = q_idx;
  a.q_val = q_val;
  a.q_ext = q_ext;
  a.q_scale = h->sharded ? q_sub : nullptr;
  a.nq = (int32_t)nq;
  a.q_nnz_end = q_nnz_end;
  // ~2 workgroups per CU per tile in flight at once, tiles swept one after another (tile-major grid) so the
  // chip works on one tile's postings at a time and they stay in L2 / Infinity Cache
  const int64_t want_chunks = getenv("APSS_CHUNKS") ? atoi(getenv("APSS_CHUNKS")) : 1024;  // (env: tuning hook)
  a.q_chunk = (int32_t)std::max<int64_t>(1, ceil_div(nq, want_chunks));
  a.n_chunks = (int32_t)ceil_div(nq, a.q_chunk);
  a.q_slot_base = q_slot_base;
  a.theta = (float)theta;
  a.counters = h->counters.p;
  a.cx_scale = (float)cx_scale;
  a.cx_theta = (float)cx_theta;

  apss_handle::IndexSet &ix = coarse_path ? h->cx : h->ex;
  if (!coarse_path) APSS_TRY(ensure_exact_index(h));
  a.tile_seg = ix.seg.p;
  a.post = h->ex.post.p;
  a.post_c = h->cx.post_c.p;
  a.tile_post_base = ix.base.p;
  a.tile_scale = h->sharded ? (coarse_path ? h->tile_min_c.p : h->tile_min.p) : nullptr;
  a.cb = ix.cb;
  a.n_tiles = (int32_t)ix.n_tiles;

  // ---- path 2: single-pass exact speed kernel (k_probe_wave); kernel shapes: A = 8 waves x 64-chunk window, one
  // workgroup per CU at 32768-row tiles; C = 8 waves x 40-chunk window, TWO workgroups per CU at <= 16384-row tiles
  const char variant = h->ex.cb <= 16384 ? 'C' : 'A';
  const int wave_block = 512;
  const int wave_u = variant == 'A' ? 8 : 5;
  const int wave_longcap = variant == 'A' ? 256 : 128;
  const int wave_survcap = variant == 'A' ? 1024 : 512;
  // chunk descriptors pack (first posting * 8 + count - 1) into 32 bits: a tile's postings must number < 2^28
  const bool wave_path = !coarse_path && mode == 0 && fx_scale > 0 && q_max_nnz <= wave_block && !forced_general &&
                         h->store_max_nnz * (int64_t)h->ex.cb + (int64_t)kSegAlign * h->cfg.dim < (1LL << 28);
  // ---- path 3: general kernel (k_probe): signed fixed point when the norms are bounded, else fp32 atomics ----
  const bool gen_fx = !coarse_path && !wave_path && fx_scale > 0;
  const double scale_used = wave_path ? fx_scale : fx_scale / 2;
  a.fx_scale = (float)scale_used;
  a.theta_fx = (uint32_t)std::min(4294967295.0, std::max(1.0, std::ceil(theta * scale_used)));
  a.theta_fxi = (int32_t)std::max(-2147483647.0, std::min(2147483647.0, std::ceil(theta * scale_used)));
  const bool cx_big = h->cx.cb > 32768;  // experiment: one 1024-thread workgroup per CU over a 65536-row tile
  // the sparse regime's wave holds few, short segments: a 3-step register window (24 chunks per wave) wastes fewer idle
  // steps than the 5-step one as long as a wave's expected chunks stay well inside it (C5 shape: 458 vs 604 ms at N=2M)
  const double cx_seg = (double)h->cx.cb * ((double)h->nnz / (double)h->n_rows) / (double)h->cfg.dim;
  const double cx_wave_chunks = ((double)q_nnz_end / (double)nq / 16.0) * std::max(1.0, cx_seg / 16.0 + 0.5);
  const bool cx_big_u3 = cx_big && cx_wave_chunks <= 17.0 && !getenv("APSS_CX_BIG_U5");
  const bool cx8 = getenv("APSS_CX_CHUNK8") != nullptr;  // experiment: 8-posting chunks, 4 steps
  const int vrow_part = 512;
  // (the filter kernels keep their LDS in static arrays: no dynamic allocation)
  const size_t lds = coarse_path ? 0
                     : wave_path ? probe_wave_lds_bytes(h->ex.cb, wave_block, wave_u, wave_longcap, wave_survcap)
                                 : probe_lds_bytes(h->ex.cb, kProbeBlock, mode);
  auto launch_wave = [&](bool diag) -> int32_t {
    const dim3 grid((unsigned)((int64_t)a.n_tiles * a.n_chunks));
#define APSS_LAUNCH_WAVE1(B, UU, LC, SC, SH, DG)                                                                         \
    do {                                                                                                                   \
      auto kern = k_probe_wave<B, UU, LC, SC, SH, DG>;                                                                     \
      HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      hipLaunchKernelGGL(kern, grid, dim3(B), lds, h->stream, a);                                                          \
    } while (0)
#define APSS_LAUNCH_WAVE(B, UU, LC, SC)                                                                                  \
    do {                                                                                                                   \
      if (h->sharded) { if (diag) APSS_LAUNCH_WAVE1(B, UU, LC, SC, true, true); else APSS_LAUNCH_WAVE1(B, UU, LC, SC, true, false); } \
      else { if (diag) APSS_LAUNCH_WAVE1(B, UU, LC, SC, false, true); else APSS_LAUNCH_WAVE1(B, UU, LC, SC, false, false); }          \
    } while (0)
    if (variant == 'A') APSS_LAUNCH_WAVE(512, 8, 256, 1024);
    else APSS_LAUNCH_WAVE(512, 5, 128, 512);
#undef APSS_LAUNCH_WAVE1
#undef APSS_LAUNCH_WAVE
    HIPCHK(h, hipGetLastError());
    return APSS_OK;
  };

  if (coarse_path && q_max_nnz > vrow_part) {
    // queries of more terms than a round takes: cut them into parts that share the accumulators
    APSS_TRY(ensure(h, h->vrow_np, (size_t)nq + 1));
    APSS_TRY(ensure(h, h->vrow_first, (size_t)nq + 2));
    APSS_TRY(ensure(h, h->vq_first, (size_t)nq + 1));
    hipLaunchKernelGGL(k_vrow_count, dim3((unsigned)ceil_div(nq, 256)), dim3(256), 0, h->stream, q_rowptr, nq, vrow_part, h->vrow_np.p);
    hipLaunchKernelGGL(k_scan_i64, dim3(1), dim3(1024), 0, h->stream, (const int64_t *)h->vrow_np.p, h->vrow_first.p, nq);
    HIPCHK(h, hipGetLastError());
    int64_t nv = 0;
    HIPCHK(h, hipMemcpyAsync(&nv, h->vrow_first.p + nq, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    APSS_TRY(ensure(h, h->vrow_ptr, (size_t)nv + 1));
    APSS_TRY(ensure(h, h->vrow_q, (size_t)nv + 1));
    hipLaunchKernelGGL(k_vrow_fill, dim3((unsigned)ceil_div(nq + 1, 256)), dim3(256), 0, h->stream, q_rowptr, nq, vrow_part,
                       (const int64_t *)h->vrow_first.p, h->vq_first.p, h->vrow_ptr.p, h->vrow_q.p);
    HIPCHK(h, hipGetLastError());
    a.vq_first = h->vq_first.p;
    a.vrow_ptr = h->vrow_ptr.p;
    a.vrow_q = h->vrow_q.p;
  }
  // one launch sweeps a group of tiles sized for roughly 2e11 posting visits (a few hundred ms): a very large join
  // becomes a sequence of launches of bounded duration instead of one kernel that runs for tens of seconds
  const int64_t total_tiles = ix.n_tiles;
  int64_t tiles_per_launch = std::max<int64_t>(1, total_tiles);
  if (total_tiles > 0) {
    const double per_tile = (double)q_nnz_end * ((double)h->nnz / (double)total_tiles / (double)h->cfg.dim) + 1.0;
    tiles_per_launch = (int64_t)std::min<double>((double)total_tiles, std::max(1.0, std::floor(2e11 / per_tile)));
    if (getenv("APSS_TILES_PER_LAUNCH")) tiles_per_launch = std::max(1, atoi(getenv("APSS_TILES_PER_LAUNCH")));  // test hook
    tiles_per_launch = std::min<int64_t>(tiles_per_launch, std::max<int64_t>(1, 2000000000LL / std::max(1, a.n_chunks)));
  }
  if (h->res_q.cap < (size_t)std::min<int64_t>(2 * nq, 1LL << 28)) {
    // room for two hits per query up front: growing means running the whole probe again
    const size_t cap0 = (size_t)std::max<int64_t>(1 << 20, std::min<int64_t>(2 * nq, 1LL << 28));
    APSS_TRY(ensure(h, h->res_q, cap0, 0, true));
    APSS_TRY(ensure(h, h->res_c, cap0, 0, true));
    APSS_TRY(ensure(h, h->res_s, cap0, 0, true));
  }
  for (int attempt = 0; attempt < 3; ++attempt) {
    a.dbg = nullptr;
    a.res_q = h->res_q.p;
    a.res_c = h->res_c.p;
    a.res_s = h->res_s.p;
    a.res_cap = h->res_q.cap;
    HIPCHK(h, hipMemsetAsync(h->counters.p, 0, kCtrCount * sizeof(unsigned long long), h->stream));
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    int64_t n_launches = 0;
    for (int64_t t0 = 0; t0 < total_tiles; t0 += tiles_per_launch, ++n_launches) {
    a.tile0 = (int32_t)t0;
    a.n_tiles = (int32_t)std::min<int64_t>(tiles_per_launch, total_tiles - t0);
    if (coarse_path) {
      if (a.vq_first && cx_big) {
        if (cx_big_u3) {
          auto kern = k_probe_coarse<1024, 3, 256, 1024, false, 16, true>;
          hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)a.n_tiles * a.n_chunks)), dim3(1024), lds, h->stream, a);
        } else {
          auto kern = k_probe_coarse<1024, 5, 256, 1024, false, 16, true>;
          hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)a.n_tiles * a.n_chunks)), dim3(1024), lds, h->stream, a);
        }
      } else if (a.vq_first && cx_signed && !cx_big) {
        auto kern = k_probe_coarse<512, 5, 128, 512, false, 16, true, true>;
        hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)a.n_tiles * a.n_chunks)), dim3(512), lds, h->stream, a);
      } else if (cx_signed && !cx_big) {
        auto kern = k_probe_coarse<512, 5, 128, 512, false, 16, false, true>;
        hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)a.n_tiles * a.n_chunks)), dim3(512), lds, h->stream, a);
      } else if (a.vq_first) {
        auto kern = k_probe_coarse<512, 5, 128, 512, false, 16, true>;
        hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)a.n_tiles * a.n_chunks)), dim3(512), lds, h->stream, a);
      } else if (cx_big && cx_signed) {
        if (cx_big_u3) {
          auto kern = k_probe_coarse<1024, 3, 256, 1024, false, 16, false, true>;
          hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)a.n_tiles * a.n_chunks)), dim3(1024), lds, h->stream, a);
        } else {
          auto kern = k_probe_coarse<1024, 5, 256, 1024, false, 16, false, true>;
          hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)a.n_tiles * a.n_chunks)), dim3(1024), lds, h->stream, a);
        }
      } else if (cx_big) {
        if (cx_big_u3) {
          auto kern = k_probe_coarse<1024, 3, 256, 1024, false>;
          hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)a.n_tiles * a.n_chunks)), dim3(1024), lds, h->stream, a);
        } else {
          auto kern = k_probe_coarse<1024, 5, 256, 1024, false>;
          hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)a.n_tiles * a.n_chunks)), dim3(1024), lds, h->stream, a);
        }
      } else if (getenv("APSS_CX_U4") && !h->sharded) {  // experiment hook: a 32-chunk window
        auto kern = k_probe_coarse<512, 4, 128, 512, false>;
        hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)a.n_tiles * a.n_chunks)), dim3(512), lds, h->stream, a);
      } else if (getenv("APSS_CX_U3") && !h->sharded) {  // test hook: a 24-chunk window, most rounds overflow it
        auto kern = k_probe_coarse<512, 3, 128, 512, false>;
        hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)a.n_tiles * a.n_chunks)), dim3(512), lds, h->stream, a);
      } else if (cx8 && !h->sharded) {
        auto kern = k_probe_coarse<512, 4, 128, 512, false, 8>;
        hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)a.n_tiles * a.n_chunks)), dim3(512), lds, h->stream, a);
      } else if (h->sharded) {
        auto kern = k_probe_coarse<512, 5, 128, 512, true>;
        hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)a.n_tiles * a.n_chunks)), dim3(512), lds, h->stream, a);
      } else {
        auto kern = k_probe_coarse<512, 5, 128, 512, false>;
        hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)a.n_tiles * a.n_chunks)), dim3(512), lds, h->stream, a);
      }
      HIPCHK(h, hipGetLastError());
    } else if (wave_path && getenv("APSS_DIAG")) {
      // diagnostic build: in-kernel cycle stamps per round segment (shares only; never a benchmark number)
      APSS_TRY(ensure(h, h->dbg, 8));
      HIPCHK(h, hipMemsetAsync(h->dbg.p, 0, 8 * sizeof(unsigned long long), h->stream));
      a.dbg = h->dbg.p;
      APSS_TRY(launch_wave(true));
      unsigned long long d[8];
      HIPCHK(h, hipMemcpyAsync(d, h->dbg.p, sizeof(d), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      unsigned long long tot = 0;
      for (int k = 0; k < 7; ++k) tot += d[k];
      fprintf(stderr, "[apss diag] stage+flatten %.1f%% | atomics %.1f%% | longs/rest %.1f%% | barrier G %.1f%% | survivors %.1f%% | "
                      "re-zero %.1f%% | barrier U %.1f%% | cycles/round/wave %.0f\n",
              100.0 * d[0] / tot, 100.0 * d[1] / tot, 100.0 * d[2] / tot, 100.0 * d[3] / tot, 100.0 * d[4] / tot,
              100.0 * d[5] / tot, 100.0 * d[6] / tot, (double)tot / ((double)a.n_tiles * a.nq * (wave_block / kWave)));
    } else if (wave_path) {
      APSS_TRY(launch_wave(false));
    } else if (gen_fx) {
      if (mode == 0) APSS_TRY((launch_probe<0, true>(h, a, lds)));
      else if (mode == 1) APSS_TRY((launch_probe<1, true>(h, a, lds)));
      else APSS_TRY((launch_probe<2, true>(h, a, lds)));
    } else {
      if (mode == 0) APSS_TRY((launch_probe<0, false>(h, a, lds)));
      else if (mode == 1) APSS_TRY((launch_probe<1, false>(h, a, lds)));
      else APSS_TRY((launch_probe<2, false>(h, a, lds)));
    }
    }  // tile groups
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    unsigned long long c[kCtrCount];
    HIPCHK(h, hipMemcpyAsync(c, h->counters.p, sizeof(c), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->st.probe_ms += ms;
    h->st.probe_launches += n_launches;
    h->st.posting_visits = (int64_t)c[kCtrVisits];
    h->st.candidate_pairs = (int64_t)c[kCtrCands];
    // the speed paths count a stored query's touch of its own slot; it is not a (q, c != q) pair
    if ((wave_path || coarse_path) && q_slot_base >= 0)
      h->st.candidate_pairs -= (q_slot_base == 0 && nq == h->n_rows) ? h->store_nonempty : h->last_batch_nonempty;
    h->st.result_pairs = (int64_t)c[kCtrResults];
    if (c[kCtrResults] > a.res_cap) {
      // the result list overflowed: grow to what the run asked for and repeat the (idempotent) probe
      const size_t need = (size_t)c[kCtrResults] + (size_t)c[kCtrResults] / 8 + 1024;
      APSS_TRY(ensure(h, h->res_q, need, 0, true));
      APSS_TRY(ensure(h, h->res_c, need, 0, true));
      APSS_TRY(ensure(h, h->res_s, need, 0, true));
      continue;
    }
    h->n_res = (int64_t)c[kCtrResults];
    h->out_q = h->res_q.p;
    h->out_c = h->res_c.p;
    h->out_s = h->res_s.p;
    if (coarse_path && !h->sharded) {
      // exact pass: re-score what the filter let through from the fp32 store and prune at theta
      const int64_t n_cand = h->n_res;
      h->st.filter_survivors = n_cand;
      h->n_res = 0;
      if (n_cand > 0) {
        APSS_TRY(ensure(h, h->fin_q, (size_t)n_cand, 0, true));
        APSS_TRY(ensure(h, h->fin_c, (size_t)n_cand, 0, true));
        APSS_TRY(ensure(h, h->fin_s, (size_t)n_cand, 0, true));
        HIPCHK(h, hipMemsetAsync(h->counters.p, 0, sizeof(unsigned long long), h->stream));
        RescoreArgs r{};
        r.n_pairs = n_cand;
        r.q_row = h->res_q.p;
        r.c_slot = h->res_c.p;
        r.q_rowptr = q_rowptr;
        r.q_idx = q_idx;
        r.q_val = q_val;
        r.c_rowptr = h->rowptr.p;
        r.c_idx = h->idx.p;
        r.c_val = h->val.p;
        r.theta = (float)theta;
        r.out_q = h->fin_q.p;
        r.out_c = h->fin_c.p;
        r.out_s = h->fin_s.p;
        r.out_count = h->counters.p;
        HIPCHK(h, hipEventRecord(h->ev0, h->stream));
        hipLaunchKernelGGL(k_rescore, dim3((unsigned)ceil_div(n_cand * kGroup, 256)), dim3(256), 0, h->stream, r);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipEventRecord(h->ev1, h->stream));
        unsigned long long nfin = 0;
        HIPCHK(h, hipMemcpyAsync(&nfin, h->counters.p, sizeof(nfin), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
        h->st.rescore_ms = ms;
        h->n_res = (int64_t)nfin;
      }
      h->out_q = h->fin_q.p;
      h->out_c = h->fin_c.p;
      h->out_s = h->fin_s.p;
      h->st.result_pairs = h->n_res;
    }
    if (n_results) *n_results = h->n_res;
    return APSS_OK;
  }
  return fail(h, APSS_E_STATE, "result buffer kept overflowing");
}

int32_t validate_host_csr(apss_handle *h, int64_t n, const int64_t *rowptr, const int32_t *indices,
                          const double *values, const int64_t *ext_ids) {
  if (n < 0) return fail(h, APSS_E_INVALID, "negative row count");
  if (n == 0) return APSS_OK;
  if (!rowptr || !ext_ids) return fail(h, APSS_E_INVALID, "null rowptr / ext_ids");
  if (rowptr[0] != 0) return fail(h, APSS_E_INVALID, "rowptr[0] must be 0");
  for (int64_t i = 0; i < n; ++i)
    if (rowptr[i + 1] < rowptr[i]) return fail(h, APSS_E_INVALID, "rowptr must be non-decreasing");
  if (rowptr[n] > 0 && (!indices || !values)) return fail(h, APSS_E_INVALID, "null indices / values");
  return APSS_OK;
}

// host CSR (double values, as SparkSparseVector.values) -> device input staging (float values).  The doubles are
// copied as they are and narrowed on the device: a host-side conversion loop cost more than the extra PCIe bytes.
__global__ void k_narrow_f64(const double *in, float *out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (float)in[i];
}

int32_t upload(apss_handle *h, int64_t n, const int64_t *rowptr, const int32_t *indices, const double *values,
               const int64_t *ext_ids) {
  const int64_t nnz = n ? rowptr[n] : 0;
  APSS_TRY(ensure(h, h->in_rowptr, (size_t)n + 1));
  APSS_TRY(ensure(h, h->in_ext, (size_t)std::max<int64_t>(n, 1)));
  APSS_TRY(ensure(h, h->in_idx, (size_t)std::max<int64_t>(nnz, 1)));
  APSS_TRY(ensure(h, h->in_val, (size_t)std::max<int64_t>(nnz, 1)));
  APSS_TRY(ensure(h, h->in_val64, (size_t)std::max<int64_t>(nnz, 1)));
  if (n) {
    HIPCHK(h, hipMemcpyAsync(h->in_rowptr.p, rowptr, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->in_ext.p, ext_ids, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
  }
  if (nnz) {
    HIPCHK(h, hipMemcpyAsync(h->in_idx.p, indices, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->in_val64.p, values, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_narrow_f64, dim3(2048), dim3(256), 0, h->stream, (const double *)h->in_val64.p, h->in_val.p, nnz);
    HIPCHK(h, hipGetLastError());
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));  // the caller's buffers may be reused on return
  return APSS_OK;
}

int32_t insert_dev_impl(apss_handle *h, int64_t n, int64_t nnz, const int64_t *d_rowptr, const int32_t *d_idx,
                        const float *d_val, const int64_t *d_ext, int64_t *first_new_row) {
  *first_new_row = h->n_rows;
  if (n == 0) return APSS_OK;
  if (h->n_rows + n > 0x7fffffffLL) return fail(h, APSS_E_INVALID, "more than 2^31 - 1 vectors in one handle");
  int64_t kept_rows = 0, kept_nnz = 0;
  APSS_TRY(ingest(h, n, nnz, d_rowptr, d_idx, d_val, d_ext, true, &kept_rows, &kept_nnz));
  const int64_t row0 = h->n_rows;
  h->n_rows += kept_rows;
  h->nnz += kept_nnz;
  APSS_TRY(build_index(h, row0));
  return APSS_OK;
}

int32_t query_dev_impl(apss_handle *h, int64_t n, int64_t nnz, const int64_t *d_rowptr, const int32_t *d_idx,
                       const float *d_val, const int64_t *d_ext, int64_t *n_results) {
  int64_t kept_rows = 0, kept_nnz = 0;
  APSS_TRY(ingest(h, n, nnz, d_rowptr, d_idx, d_val, d_ext, false, &kept_rows, &kept_nnz));
  return probe(h, kept_rows, h->q_rowptr.p, h->q_idx.p, h->q_val.p, h->q_ext.p, h->q_sub.p, -1, h->q_max_nnz, h->q_max_norm2,
               kept_nnz, n_results);
}

}  // namespace

// =====================================================================================================
extern "C" {

int32_t apss_create(const apss_config *cfg, apss_handle **out) {
  if (!cfg || !out) {
    g_create_error = "null argument";
    return APSS_E_INVALID;
  }
  *out = nullptr;
  if (cfg->struct_size != (int32_t)sizeof(apss_config)) {
    g_create_error = "apss_config.struct_size mismatch";
    return APSS_E_INVALID;
  }
  if (cfg->dim <= 0 || !std::isfinite(cfg->theta)) {
    g_create_error = "dim must be > 0 and theta finite";
    return APSS_E_INVALID;
  }
  apss_handle *h = new (std::nothrow) apss_handle();
  if (!h) return APSS_E_NOMEM;
  h->cfg = *cfg;
  if (h->cfg.term_hi == 0 && h->cfg.term_lo == 0) h->cfg.term_hi = cfg->dim;
  if (h->cfg.term_lo < 0 || h->cfg.term_hi > cfg->dim || h->cfg.term_lo >= h->cfg.term_hi) {
    g_create_error = "bad term range";
    delete h;
    return APSS_E_INVALID;
  }
  h->sharded = !(h->cfg.term_lo == 0 && h->cfg.term_hi == cfg->dim);
  // term-range shards use the coarse filter too (measured T=2: 130 ms vs 195 ms per shard with the single-pass kernel);
  // their survivors are the shard's candidates, scored exactly in phase 2.  APSS_SHARD_EXACT=1: single-pass kernel.
  h->use_coarse = !(cfg->flags & (APSS_FLAG_EXACT_ACCUM | APSS_FLAG_FORCE_GENERAL | APSS_FLAG_FORCE_SCAN)) &&
                  (!h->sharded || !getenv("APSS_SHARD_EXACT"));
  h->cb = cfg->tile_rows ? cfg->tile_rows : 16384;
  h->ex.cb = h->cb;
  h->ex.align = kSegAlign;
  h->cx.cb = std::min(2 * h->cb, 32768);
  if (getenv("APSS_CX_TILE")) h->cx.cb = atoi(getenv("APSS_CX_TILE"));  // experiment hook (multiple of 64, <= 65536)
  h->cx.align = kSegAlignC;
  h->cx.coarse = true;
  if (h->cb < 64 || h->cb > 32768 || (h->cb % 64)) {
    g_create_error = "tile_rows must be a multiple of 64 in [64, 32768]";
    delete h;
    return APSS_E_INVALID;
  }
  if (h->sharded && !(cfg->theta > 0.0)) {
    g_create_error = "term-range shards need theta > 0 (candidate test p_g >= theta*|q_g|*|c_g|)";
    delete h;
    return APSS_E_UNSUPPORTED;
  }
  h->dev = cfg->device_id;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0 || h->dev < 0 || h->dev >= ndev) {
    g_create_error = std::string("no usable HIP device (there is no CPU fallback): ") +
                     (e != hipSuccess ? hipGetErrorString(e) : "device ordinal out of range");
    delete h;
    return APSS_E_DEVICE;
  }
  if ((e = hipSetDevice(h->dev)) != hipSuccess || (e = hipStreamCreateWithFlags(&h->own_stream, hipStreamDefault)) != hipSuccess ||
      (e = hipEventCreate(&h->ev0)) != hipSuccess || (e = hipEventCreate(&h->ev1)) != hipSuccess) {
    g_create_error = std::string("HIP init failed: ") + hipGetErrorString(e);
    delete h;
    return APSS_E_DEVICE;
  }
  h->stream = h->own_stream;
  if (cfg->capacity_rows > 0) {
    if (ensure(h, h->rowptr, (size_t)cfg->capacity_rows + 1, 0, true) != APSS_OK ||
        ensure(h, h->ext, (size_t)cfg->capacity_rows, 0, true) != APSS_OK) {
      g_create_error = h->err;
      apss_destroy(h);
      return APSS_E_NOMEM;
    }
  }
  if (cfg->capacity_nnz > 0) {
    if (ensure(h, h->idx, (size_t)cfg->capacity_nnz, 0, true) != APSS_OK ||
        ensure(h, h->erow, (size_t)cfg->capacity_nnz, 0, true) != APSS_OK ||
        ensure(h, h->val, (size_t)cfg->capacity_nnz, 0, true) != APSS_OK ||
        (h->use_coarse ? ensure(h, h->cx.post_c, (size_t)cfg->capacity_nnz * 2 + 64, 0, true)
                       : ensure(h, h->ex.post, (size_t)cfg->capacity_nnz + (size_t)cfg->capacity_nnz / 2 + 64, 0, true)) != APSS_OK) {
      g_create_error = h->err;
      apss_destroy(h);
      return APSS_E_NOMEM;
    }
  }
  *out = h;
  return APSS_OK;
}

void apss_destroy(apss_handle *h) {
  if (!h) return;
  (void)hipSetDevice(h->dev);
  if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
  release(h->rowptr); release(h->ext); release(h->idx); release(h->erow); release(h->val); release(h->sub);
  for (apss_handle::IndexSet *s : {&h->ex, &h->cx}) { release(s->seg); release(s->post); release(s->post_c); release(s->base); release(s->total); }
  release(h->tile_min); release(h->tile_min_c); release(h->fin_q); release(h->fin_c); release(h->fin_s);
  release(h->q_rowptr); release(h->q_ext); release(h->q_idx); release(h->q_val); release(h->q_sub);
  release(h->s_keep); release(h->s_cnt); release(h->s_rowdst); release(h->s_nnzdst);
  release(h->in_rowptr); release(h->in_ext); release(h->in_idx); release(h->s_inv); release(h->s_sub); release(h->in_val); release(h->in_val64); release(h->vq_first); release(h->vrow_q); release(h->vrow_ptr); release(h->vrow_np); release(h->vrow_first);
  release(h->res_q); release(h->res_c); release(h->res_s); release(h->counters); release(h->flagword); release(h->dbg);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
}

const char *apss_last_error(const apss_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int32_t apss_set_stream(apss_handle *h, void *hip_stream, int32_t use_own) {
  APSS_TRY(enter(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->stream = use_own ? h->own_stream : (hipStream_t)hip_stream;  // NULL is the device's default stream
  return APSS_OK;
}

int32_t apss_insert(apss_handle *h, int64_t n, const int64_t *rowptr, const int32_t *indices, const double *values,
                    const int64_t *ext_ids) {
  APSS_TRY(enter(h));
  APSS_TRY(validate_host_csr(h, n, rowptr, indices, values, ext_ids));
  if (n == 0) return APSS_OK;
  APSS_TRY(upload(h, n, rowptr, indices, values, ext_ids));
  int64_t first = 0;
  return insert_dev_impl(h, n, rowptr[n], h->in_rowptr.p, h->in_idx.p, h->in_val.p, h->in_ext.p, &first);
}

int32_t apss_query(apss_handle *h, int64_t n, const int64_t *rowptr, const int32_t *indices, const double *values,
                   const int64_t *ext_ids, int64_t *n_results) {
  APSS_TRY(enter(h));
  APSS_TRY(validate_host_csr(h, n, rowptr, indices, values, ext_ids));
  APSS_TRY(upload(h, n, rowptr, indices, values, ext_ids));
  return query_dev_impl(h, n, n ? rowptr[n] : 0, h->in_rowptr.p, h->in_idx.p, h->in_val.p, h->in_ext.p, n_results);
}

int32_t apss_insert_and_query(apss_handle *h, int64_t n, const int64_t *rowptr, const int32_t *indices,
                              const double *values, const int64_t *ext_ids, int64_t *n_results) {
  APSS_TRY(enter(h));
  APSS_TRY(validate_host_csr(h, n, rowptr, indices, values, ext_ids));
  APSS_TRY(upload(h, n, rowptr, indices, values, ext_ids));
  return apss_insert_and_query_dev(h, n, n ? rowptr[n] : 0, h->in_rowptr.p, h->in_idx.p, h->in_val.p, h->in_ext.p,
                                   n_results);
}

int32_t apss_self_join(apss_handle *h, int64_t *n_results) {
  APSS_TRY(enter(h));
  return probe(h, h->n_rows, h->rowptr.p, h->idx.p, h->val.p, h->ext.p, h->sub.p, 0, h->store_max_nnz, h->store_max_norm2, h->nnz,
               n_results);
}

int32_t apss_result_count(const apss_handle *h, int64_t *n_results) {
  if (!h || !n_results) return APSS_E_INVALID;
  if (h->n_res < 0) return APSS_E_STATE;
  *n_results = h->n_res;
  return APSS_OK;
}

__global__ void k_gather_ids(const int32_t *res_q, const int32_t *res_c, const int64_t *q_ext, const int64_t *c_ext,
                             int64_t n, int64_t *out_q, int64_t *out_c) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    out_q[i] = q_ext[res_q[i]];
    out_c[i] = c_ext[res_c[i]];
  }
}

int32_t apss_fetch_results(apss_handle *h, int64_t offset, int64_t count, int64_t *out_q, int64_t *out_c,
                           float *out_score) {
  APSS_TRY(enter(h));
  if (h->n_res < 0) return fail(h, APSS_E_STATE, "no query has run on this handle");
  if (offset < 0 || count < 0 || offset + count > h->n_res) return fail(h, APSS_E_INVALID, "fetch range out of bounds");
  if (count == 0) return APSS_OK;
  if (!out_q || !out_c || !out_score) return fail(h, APSS_E_INVALID, "null output buffer");
  // map (query row, candidate slot) to external ids on the device, then copy out
  APSS_TRY(ensure(h, h->s_rowdst, (size_t)count));
  APSS_TRY(ensure(h, h->s_nnzdst, (size_t)count));
  hipLaunchKernelGGL(k_gather_ids, dim3((unsigned)ceil_div(count, 256)), dim3(256), 0, h->stream,
                     h->out_q + offset, h->out_c + offset, h->res_q_ext,
                     (const int64_t *)h->ext.p, count, h->s_rowdst.p, h->s_nnzdst.p);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(out_q, h->s_rowdst.p, (size_t)count * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(out_c, h->s_nnzdst.p, (size_t)count * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(out_score, h->out_s + offset, (size_t)count * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return APSS_OK;
}

int32_t apss_size(const apss_handle *h, int64_t *rows, int64_t *nnz) {
  if (!h) return APSS_E_INVALID;
  if (rows) *rows = h->n_rows;
  if (nnz) *nnz = h->nnz;
  return APSS_OK;
}

int32_t apss_stats_get(apss_handle *h, apss_stats *out) {
  if (!h || !out) return APSS_E_INVALID;
  h->st.rows = h->n_rows;
  h->st.nnz = h->nnz;
  h->st.tiles = h->use_coarse && h->ex_built_rows < h->n_rows ? h->cx.n_tiles : ceil_div(h->n_rows, h->ex.cb);
  h->st.hbm_bytes = (int64_t)h->bytes_reserved;
  *out = h->st;
  return APSS_OK;
}

int32_t apss_insert_dev(apss_handle *h, int64_t n, int64_t nnz, const int64_t *d_rowptr, const int32_t *d_indices,
                        const float *d_values, const int64_t *d_ext_ids) {
  APSS_TRY(enter(h));
  if (n < 0 || nnz < 0) return fail(h, APSS_E_INVALID, "negative size");
  if (n > 0 && (!d_rowptr || !d_ext_ids || (nnz > 0 && (!d_indices || !d_values))))
    return fail(h, APSS_E_INVALID, "null device pointer");
  int64_t first = 0;
  return insert_dev_impl(h, n, nnz, d_rowptr, d_indices, d_values, d_ext_ids, &first);
}

int32_t apss_query_dev(apss_handle *h, int64_t n, int64_t nnz, const int64_t *d_rowptr, const int32_t *d_indices,
                       const float *d_values, const int64_t *d_ext_ids, int64_t *n_results) {
  APSS_TRY(enter(h));
  if (n < 0 || nnz < 0) return fail(h, APSS_E_INVALID, "negative size");
  if (n > 0 && (!d_rowptr || !d_ext_ids || (nnz > 0 && (!d_indices || !d_values))))
    return fail(h, APSS_E_INVALID, "null device pointer");
  return query_dev_impl(h, n, nnz, d_rowptr, d_indices, d_values, d_ext_ids, n_results);
}

int32_t apss_insert_and_query_dev(apss_handle *h, int64_t n, int64_t nnz, const int64_t *d_rowptr,
                                  const int32_t *d_indices, const float *d_values, const int64_t *d_ext_ids,
                                  int64_t *n_results) {
  APSS_TRY(enter(h));
  if (n < 0 || nnz < 0) return fail(h, APSS_E_INVALID, "negative size");
  if (n > 0 && (!d_rowptr || !d_ext_ids || (nnz > 0 && (!d_indices || !d_values))))
    return fail(h, APSS_E_INVALID, "null device pointer");
  int64_t first = 0;
  APSS_TRY(insert_dev_impl(h, n, nnz, d_rowptr, d_indices, d_values, d_ext_ids, &first));
  const int64_t nq = h->n_rows - first;
  // the batch is now rows [first, n_rows) of the store: query it in place (rowptr offsets are absolute)
  return probe(h, nq, h->rowptr.p + first, h->idx.p, h->val.p, h->ext.p + first, h->sharded ? h->sub.p + first : nullptr,
               first, h->store_max_nnz, h->store_max_norm2, h->nnz, n_results);
}

int32_t apss_clear(apss_handle *h) {
  APSS_TRY(enter(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->n_rows = 0;
  h->nnz = 0;
  h->n_tiles = 0;
  for (apss_handle::IndexSet *s : {&h->ex, &h->cx}) { s->n_tiles = 0; s->post_used = 0; s->h_base.clear(); }
  h->ex_built_rows = 0;
  h->n_res = -1;
  h->nonneg = true;
  h->store_max_nnz = 0;
  h->store_max_norm2 = 0.f;
  h->store_nonempty = 0;
  return APSS_OK;
}

int32_t apss_results_dev(apss_handle *h, const int32_t **d_q_row, const int32_t **d_c_slot, const float **d_score,
                         int64_t *n_results) {
  if (!h) return APSS_E_INVALID;
  if (h->n_res < 0) return fail(h, APSS_E_STATE, "no query has run on this handle");
  if (d_q_row) *d_q_row = h->out_q;
  if (d_c_slot) *d_c_slot = h->out_c;
  if (d_score) *d_score = h->out_s;
  if (n_results) *n_results = h->n_res;
  return APSS_OK;
}

int32_t apss_partial_scores_dev(apss_handle *h, int64_t n_pairs, const int32_t *d_q_row, const int32_t *d_c_slot,
                                float *d_out_partial) {
  APSS_TRY(enter(h));
  if (n_pairs < 0) return fail(h, APSS_E_INVALID, "negative pair count");
  if (n_pairs == 0) return APSS_OK;
  if (!h->last_q_rowptr) return fail(h, APSS_E_STATE, "no query batch on this handle");
  if (!d_q_row || !d_c_slot || !d_out_partial) return fail(h, APSS_E_INVALID, "null device pointer");
  PartialArgs a{};
  a.n_pairs = n_pairs;
  a.q_row = d_q_row;
  a.c_slot = d_c_slot;
  a.q_rowptr = h->last_q_rowptr;
  a.q_idx = h->last_q_idx;
  a.q_val = h->last_q_val;
  a.c_rowptr = h->rowptr.p;
  a.c_idx = h->idx.p;
  a.c_val = h->val.p;
  a.out = d_out_partial;
  a.nq = h->last_nq;
  a.n_rows = h->n_rows;
  hipLaunchKernelGGL(k_partial_scores, dim3((unsigned)ceil_div(n_pairs * kGroup, 256)), dim3(256), 0, h->stream, a);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return APSS_OK;
}

}  // extern "C"
