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
#include "apss_head.hpp"
#include "apss_even.hpp"

using namespace apss;

namespace {

thread_local std::string g_create_error;

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;  // elements
};

// Test and experiment hooks: ONE environment word, APSS_DEBUG, read once when a handle is created -- a comma-separated
// list of the tokens below.  Nothing else in the library reads the environment.
struct DebugCfg {
  bool build_lds = false;      // build_lds      index build with LDS cursors even for a handful of tiles
  bool build_atomic = false;   // build_atomic   index build with global atomics even for small dims
  bool chunk8 = false;         // chunk8         filter kernel with 8-posting chunks (4-step window)
  bool shard_exact = false;    // shard_exact    term-range shards run the single-pass exact kernel
  bool diag = false;           // diag           in-kernel cycle stamps of the single-pass kernel (stderr)
  bool force_general = false;  // force_general  as APSS_FLAG_FORCE_GENERAL
  bool exact_accum = false;    // exact_accum    as APSS_FLAG_EXACT_ACCUM, decided per probe
  bool big_u5 = false;         // big_u5         1024-thread filter kernel keeps its 5-step window
  int cx_tile = 0;             // cx_tile=N      rows per tile of the coarse index (multiple of 64, <= 65536)
  int window = 0;              // window=U       register-window steps of the 512-thread filter kernel (2..5)
  int chunks = 0;              // chunks=N       query chunks per tile (grid shape)
  int tiles_per_launch = 0;    // tiles_per_launch=N
  bool no_tail = false;        // no_tail        every insert extends the tile index at once (no tail of waiting rows)
  bool no_acc8 = false;        // no_acc8        term shards keep 16-bit accumulators over 32768-row tiles
  int flat_group = -1;         // flat_group=L   k_probe_even: 2^L staging lanes per term (default: as many as keep the staging waves <= 1/4)
  int pad_lds = 0;             // pad_lds=N      N bytes of dynamic LDS on the filter launch (occupancy experiments)
  bool longpf = false;         // longpf         prefetched, group-parallel long-segment sweeps on a plain handle too (experiment)
  bool even_wide = false;      // even_wide      k_probe_even also with one staging lane per term and windows of up to 7 steps (experiment)
  bool no_even = false;        // no_even        the filter stages every round on all waves (k_probe_coarse), never on F of them (k_probe_even)
  int seg_align = 0;           // seg_align=N    postings per aligned unit of the coarse index (16 | 32)
  bool bank_order = false;     // bank_order     experiment: bank-aware posting order inside short segments (k_seg_bank_order)
  int fold_w = 0;              // fold_w=N       columns of the dense head's folded block (128 | 256)
  bool no_sym = false;         // no_sym         as APSS_FLAG_NO_SYMMETRY
  int mix = 0;                 // mix=E          ONE head block of 256 columns: E terms with a column each, the others folded into 256 - E
  bool no_chain = false;       // no_chain       small batches take the exact pass with its own host round trips (as large ones do)
  bool no_append = false;      // no_append      a batch that lands in a partly filled tile rebuilds the whole tile (never appends to it)
  bool no_bucket = false;      // no_bucket      large dims build with global atomics (never the bucketed LDS build)
  int bucket_range = 0;        // bucket_range=N terms per range of the bucketed LDS build (a power of two <= 16384; default 2048)
  int res_cap = 0;             // res_cap=N      initial capacity of the candidate list (tests of the overflow -> regrow -> re-run path)
  bool head_bf16 = false;      // head_bf16      the dense-head block keeps bf16 rows (v_mfma_f32_32x32x16_bf16), never the INT8 rendering
  int merge = -1;              // merge=L        a term shard's thin rounds: at most 2^L query rows share a round (0: never; default 1: two rows)
  bool no_merge_prune = false; // no_merge_prune merged rounds hand every expanded pair to the exchange (no exact shard-rule test behind k_expand_merged)
  int merge_u = 0;             // merge_u=U      ... as long as the merged round fits a window of U steps (default 7)
  int merge_single = 0;        // merge_single=U ... and a single row's round takes a window of at most U steps (default 4)
};

DebugCfg parse_debug_env() {
  DebugCfg d;
  const char *e = getenv("APSS_DEBUG");
  if (!e) return d;
  std::string str(e);
  size_t pos = 0;
  while (pos <= str.size()) {
    const size_t end = std::min(str.find(',', pos), str.size());
    const std::string tok = str.substr(pos, end - pos);
    pos = end + 1;
    const size_t eq = tok.find('=');
    const std::string key = tok.substr(0, eq);
    const int val = eq == std::string::npos ? 1 : atoi(tok.c_str() + eq + 1);
    if (key == "build_lds") d.build_lds = val != 0;
    else if (key == "build_atomic") d.build_atomic = val != 0;
    else if (key == "chunk8") d.chunk8 = val != 0;
    else if (key == "shard_exact") d.shard_exact = val != 0;
    else if (key == "diag") d.diag = val != 0;
    else if (key == "force_general") d.force_general = val != 0;
    else if (key == "exact_accum") d.exact_accum = val != 0;
    else if (key == "big_u5") d.big_u5 = val != 0;
    else if (key == "cx_tile") d.cx_tile = val;
    else if (key == "window") d.window = val;
    else if (key == "chunks") d.chunks = val;
    else if (key == "tiles_per_launch") d.tiles_per_launch = val;
    else if (key == "no_tail") d.no_tail = val != 0;
    else if (key == "no_acc8") d.no_acc8 = val != 0;
    else if (key == "no_even") d.no_even = val != 0;
    else if (key == "even_wide") d.even_wide = val != 0;
    else if (key == "longpf") d.longpf = val != 0;
    else if (key == "pad_lds") d.pad_lds = (int)val;
    else if (key == "flat_group") d.flat_group = (int)val;
    else if (key == "seg_align") d.seg_align = val;
    else if (key == "bank_order") d.bank_order = val != 0;
    else if (key == "fold_w") d.fold_w = val;
    else if (key == "mix") d.mix = val;
    else if (key == "no_sym") d.no_sym = val != 0;
    else if (key == "no_chain") d.no_chain = val != 0;
    else if (key == "no_append") d.no_append = val != 0;
    else if (key == "head_bf16") d.head_bf16 = val != 0;
    else if (key == "res_cap") d.res_cap = val;
    else if (key == "no_bucket") d.no_bucket = val != 0;
    else if (key == "bucket_range") d.bucket_range = val;
    else if (key == "merge") d.merge = (int)val;
    else if (key == "merge_u") d.merge_u = (int)val;
    else if (key == "no_merge_prune") d.no_merge_prune = val != 0;
    else if (key == "merge_single") d.merge_single = (int)val;
    else if (!key.empty()) fprintf(stderr, "[apss] unknown APSS_DEBUG token '%s' ignored\n", key.c_str());
  }
  return d;
}

}  // namespace

struct apss_handle {
  apss_config cfg{};
  DebugCfg dbgcfg;
  int dev = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::string err;
  bool sharded = false;
  bool nonneg = true;    // every stored weight so far is >= 0
  bool q_nonneg = true;  // every weight of the staged query batch is >= 0 (decided per call: a signed query batch does not change the handle)
  bool no_acc8 = false;  // the 8-bit filter proved unusable on this handle's data: 16-bit accumulators until apss_clear
  uint32_t downgrades = 0;  // APSS_DOWNGRADE_* (apss_stats): permanent fallbacks this handle has taken since the last clear
  int64_t store_max_nnz = 0, q_max_nnz = 0;  // longest row of the store / of the staged query batch
  float store_max_norm2 = 0.f, q_max_norm2 = 0.f;  // largest squared row norm (bounds every partial score)
  int64_t store_nonempty = 0;  // stored rows with at least one indexed entry (each touches itself in a self-join)
  int64_t last_batch_nonempty = 0;
  int32_t cb = 16384;

  // store (CSR) -- vectorsStore, IWA:22
  int64_t n_rows = 0, nnz = 0;
  int64_t idx_rows = 0;  // rows [0, idx_rows) are in the tile index (and in W); rows [idx_rows, n_rows) wait in the tail (probe: k_tail_score)
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
    DevBuf<uint32_t> scan_part;  // block sums of the segment-length scan (k_tile_scan_part)
    DevBuf<uint32_t> maxlen;   // [2] longest (tile, term) segment, number of long segments, over the builds since the rendering was last started from tile 0
    uint32_t max_seg = 0;      // host copies
    uint32_t long_segs = 0;
    DevBuf<unsigned long long> chunkw;  // [1] sum over (tile, term) of length x chunks (k_tile_scan)
    double round_chunks = 0.0;          // chunks an average stored row deals out per round (tile): chunkw / rows
    std::vector<int64_t> h_base;
    double build_ms = 0;
    int64_t built_rows = 0;  // the rendering covers store rows [0, built_rows) (an append needs to start exactly there)
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
  DevBuf<int64_t> scan_tmp;  // block sums of a multi-block scan and their scan
  DevBuf<int32_t> in_idx;
  DevBuf<float> s_inv, s_sub, in_val;
  DevBuf<double> in_val64;
  DevBuf<unsigned long long> bk_cnt;  // bucketed LDS build (large dims): entries per (tile, term range) + cursors,
  DevBuf<int64_t> bk_base;            //   their exclusive scan,
  DevBuf<int32_t> bk_idx;             //   and the entries partitioned by (tile, range): term, store row, value
  DevBuf<uint32_t> bk_erow;
  DevBuf<float> bk_val;
  DevBuf<uint2> app_seg;             // append build: the last tile's segment table and postings before the append
  DevBuf<char> app_post;
  DevBuf<int32_t> vq_first, vrow_q;  // virtual-row table of the last query batch (queries of > 512 terms)
  DevBuf<int64_t> vrow_ptr, vrow_np, vrow_first;
  // results of the last query-type call
  DevBuf<int32_t> res_q, res_c, fin_q, fin_c;
  DevBuf<float> res_s, fin_s;
  DevBuf<int32_t> res2_q, res2_c;    // merged rounds (k_expand_merged): the second candidate list, swapped with res_* after the expansion
  DevBuf<float> res2_s, q_prenorm;   // ... and the staged query weights divided by their rows' shard factors
  const int32_t *out_q = nullptr, *out_c = nullptr;  // where the last call's results live (res_* or fin_*)
  const float *out_s = nullptr;
  int64_t n_res = -1;
  const int64_t *res_q_ext = nullptr;      // ext ids of the last query batch (device)
  const int64_t *last_q_rowptr = nullptr;  // last query batch CSR (device), for apss_partial_scores_dev
  const int32_t *last_q_idx = nullptr;
  const float *last_q_val = nullptr;
  int64_t last_nq = 0;
  DevBuf<unsigned long long> counters, dbg, chain_ctr;
  DevBuf<unsigned int> flagword;
  // small messages (single vectors, batches of a few dozen: the reference's LoadGenerator sends ONE vector per message): the
  // batch crosses PCIe as one packed copy out of pinned memory and the answer comes back the same way
  char *pin = nullptr;             // pinned host staging, kPinBytes
  DevBuf<char> pack;               // its device twin
  const int64_t *up_rowptr = nullptr, *up_ext = nullptr;  // where upload() left the batch (the in_* arrays, or views into `pack`)
  const int32_t *up_idx = nullptr;
  // dense-head block (apss_head.hpp): the KH most frequent terms live in W instead of the inverted index
  int32_t head_k = 0;                 // 0: no block
  bool head_fixed = false;            // the block's terms were set through apss_set_head_terms: no policy, kept across apss_clear
  // geometry of a head of more than 256 terms: ONE block of 256 columns -- the head_exact most frequent terms with a column
  // each, the others FOLDED into the remaining head_fold_w = 256 - head_exact columns.  (APSS_DEBUG=fold_w=128|256 keeps
  // round 3's first form for comparison: 256 columns with a term each + a second block of fold_w folded columns.)
  int32_t head_fold_w = 128;
  int32_t head_exact = 128;
  int32_t head_fold_user = 0;         // folded columns named by the caller for the terms it sets (apss_set_head_fold; 0: 128)
  bool merge_off = false;             // term shard: merged rounds (apss_even.hpp) reported too many candidates on this data: not again (until apss_clear)
  bool head_longseg = false;          // term shard with a block: its tail still has segments too long for the thin-round kernel (a
                                      // hint kept across apss_clear: the next first build goes straight to the layout that serves them)
  int32_t head_part = 0, head_parts = 1;  // this handle multiplies the candidate tiles t % head_parts == head_part of the block
  int64_t head_eval_rows = 0;         // store size when the head policy last looked at the term distribution
  bool head_blocked = false;          // a call needed the plain path: no block until the next apss_clear
  std::vector<int32_t> head_terms;    // the block's terms, in block order
  DevBuf<int32_t> head_pos;           // [dim] term -> column of the block | -1
  // tail view (apss_head.hpp): the rows without the block's entries -- index build input and the sparse filter's query rows
  struct TailView {
    DevBuf<int64_t> rowptr;
    DevBuf<int32_t> idx;
    DevBuf<float> val;
    DevBuf<uint32_t> erow;
    int64_t rows = 0, nnz = 0, max_nnz = 0, nonempty = 0, last_nonempty = 0;
  };
  TailView tv, qtv;                   // of the store rows [0, idx_rows); of the staged (outside) query batch
  DevBuf<int64_t> tv_cnt, tv_off;
  DevBuf<unsigned int> tv_sum;
  DevBuf<uint16_t> W, q_W;            // [rows x head_k] rows of the store / of a staged query batch: bf16, or INT8 in the first half
                                      // of the same reservation (head_i8: one byte per column, rounded up; apss_head.hpp)
  bool head_i8 = false;               // the block's rows are the INT8 rendering (decided when the block's first row is packed)
  float head_s = 0.f;                 // the INT8 scale S (units per 1.0): 127 / the largest row norm so far (head_fit_scale)
  DevBuf<unsigned int> head_ovf;      // [1] set by k_head_pack if an element ever exceeded 127 (cannot happen: checked, an error)
  DevBuf<uint32_t> df;
  DevBuf<unsigned long long> dedup_tab, head_ctr;
  DevBuf<int32_t> uq_q, uq_c;         // candidate list after k_pair_dedup
  DevBuf<float> uq_s;
  int64_t head_nonempty = 0, last_batch_head_nonempty = 0;
  double head_sample_frac = 0.0;      // fraction of sampled pairs the dense filter passed when the policy last looked
  hipEvent_t ev2 = nullptr, ev3 = nullptr;
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
constexpr size_t kPinBytes = 256 << 10;   // pinned staging of a small batch / a small answer ...
constexpr size_t kPinScalars = 1024;      // ... + this much behind it for the small device-to-host reads (flags, counters)

// ---- ingest: validate (+ optional normalise / admission / value prune / term-range filter) and append ----
// Source arrays are device pointers (batch-relative rowptr).  Destination: the store (to_store) or the query
// staging buffers.  *n_out / *nnz_out: rows / entries that survived.
// exclusive scan of n int64 (n + 1 outputs, out[n] = total) on the handle's stream
int32_t scan_i64(apss_handle *h, const int64_t *in, int64_t *out, int64_t n) {
  if (n <= 4 * kScanBlock) {
    hipLaunchKernelGGL(k_scan_i64, dim3(1), dim3(1024), 0, h->stream, in, out, n);
    HIPCHK(h, hipGetLastError());
    return APSS_OK;
  }
  const int64_t nb = ceil_div(n, kScanBlock);
  APSS_TRY(ensure(h, h->scan_tmp, (size_t)(2 * nb + 1)));
  int64_t *sums = h->scan_tmp.p, *offs = h->scan_tmp.p + nb;
  hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)nb), dim3(1024), 0, h->stream, in, sums, n);
  hipLaunchKernelGGL(k_scan_i64, dim3(1), dim3(1024), 0, h->stream, (const int64_t *)sums, offs, nb);
  hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(1024), 0, h->stream, in, (const int64_t *)offs, out, n, nb);
  HIPCHK(h, hipGetLastError());
  return APSS_OK;
}

int32_t head_setup_rendering(apss_handle *h);
int32_t head_fit_scale(apss_handle *h, double norm2, unsigned char *W, int64_t packed_bytes);
inline void head_pack_rendering(const apss_handle *h, HeadPackArgs &p);

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
  const bool shard_head = h->sharded && h->head_k > 0;
  a.head_pos = shard_head ? h->head_pos.p : nullptr;
  const int threads = 256;
  const int64_t blocks = ceil_div(n * kGroup, threads);
  // (the workgroups loop over their rows: few workgroups = few same-address atomics on the batch summaries)
  hipLaunchKernelGGL(k_ingest_count<kGroup>, dim3((unsigned)std::min<int64_t>(blocks, 4096)), dim3(threads), 0, h->stream, a);
  HIPCHK(h, hipGetLastError());

  // destination
  int64_t dst_row0 = to_store ? h->n_rows : 0, dst_nnz0 = to_store ? h->nnz : 0;
  int64_t kept_rows = n, kept_nnz = nnz;
  unsigned int flags_stack[4] = {0, 0, 0, 0};
  // (read back into pinned memory when the handle has it: a copy into pageable memory is staged by the runtime)
  unsigned int *flags_host = h->pin ? reinterpret_cast<unsigned int *>(h->pin + kPinBytes) : flags_stack;
  if (transform) {
    APSS_TRY(ensure(h, h->s_rowdst, (size_t)n + 1));
    APSS_TRY(ensure(h, h->s_nnzdst, (size_t)n + 1));
    APSS_TRY(scan_i64(h, (const int64_t *)h->s_keep.p, h->s_rowdst.p, n));
    APSS_TRY(scan_i64(h, (const int64_t *)h->s_cnt.p, h->s_nnzdst.p, n));
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(&kept_rows, h->s_rowdst.p + n, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(&kept_nnz, h->s_nnzdst.p + n, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
  }
  HIPCHK(h, hipMemcpyAsync(flags_host, h->flagword.p, 4 * sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));  // (pinned or stack)
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (flags_host[0] & 1u)
    return fail(h, APSS_E_INVALID, "malformed vector: indices must be strictly increasing and in [0, dim) "
                                   "(SparseVector.scala:75; vectorDim mismatch is the require of CommonUtils.scala:99)");
  if (flags_host[0] & 2u) return fail(h, APSS_E_INVALID, "non-finite value in a vector");
  if (to_store) h->nonneg = h->nonneg && !(flags_host[0] & 4u);
  else h->q_nonneg = !(flags_host[0] & 4u);
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
  if (shard_head) {
    // the rows of the dense-head block come from the batch as the caller handed it in (whole rows: the store keeps this
    // shard's term range only); rows map 1:1 (apss_set_head_terms refuses the admission filter on a shard)
    const int64_t kh = h->head_k;
    DevBuf<uint16_t> &o_W = to_store ? h->W : h->q_W;
    const int64_t rows_end = dst_row0 + kept_rows;
    const int64_t w_pad = to_store ? ceil_div(rows_end, kHeadQBlock) * kHeadQBlock + kHeadCTile : ceil_div(rows_end, kHeadCTile) * kHeadCTile;
    APSS_TRY(ensure(h, o_W, (size_t)(w_pad * kh), (size_t)(ceil_div(dst_row0, kHeadCTile) * kHeadCTile * kh)));
    APSS_TRY(ensure(h, h->head_ctr, 4));
    HIPCHK(h, hipMemsetAsync(h->head_ctr.p, 0, 4 * sizeof(unsigned long long), h->stream));
    if (to_store && dst_row0 == 0) APSS_TRY(head_setup_rendering(h));
    APSS_TRY(head_fit_scale(h, to_store ? (double)h->store_max_norm2 : std::max((double)h->store_max_norm2, (double)h->q_max_norm2),
                            reinterpret_cast<unsigned char *>(h->W.p), ceil_div(to_store ? dst_row0 : h->n_rows, kHeadCTile) * kHeadCTile * kh));
    HeadPackArgs p{};
    head_pack_rendering(h, p);
    p.rowptr = d_rowptr;
    p.idx = d_idx;
    p.val = d_val;
    p.row0 = 0;
    p.row1 = n;
    p.head_pos = h->head_pos.p;
    p.kh = (int32_t)kh;
    p.W = o_W.p;
    p.w_row0 = dst_row0;
    p.w_pad = w_pad;
    p.ratio_t = nullptr;  // k_ingest_count wrote this shard's ratio
    p.head_nonempty = reinterpret_cast<unsigned int *>(h->head_ctr.p);
    p.row_inv = (h->cfg.flags & APSS_FLAG_NORMALIZE) ? h->s_inv.p : nullptr;
    p.prune_above = (h->cfg.flags & APSS_FLAG_VALUE_PRUNE) ? (float)h->cfg.index_threshold : -INFINITY;
    p.fold_from = h->head_exact;
    p.part = h->head_part;  // (a stored query's product with itself is counted by the shard that owns its tile)
    p.n_parts = h->head_parts;
    hipLaunchKernelGGL(k_head_pack, dim3((unsigned)(ceil_div(w_pad, 8) - dst_row0 / 8)), dim3(512), 0, h->stream, p);
    HIPCHK(h, hipGetLastError());
    if (to_store) {
      unsigned int nz = 0;
      HIPCHK(h, hipMemcpyAsync(&nz, h->head_ctr.p, sizeof(nz), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      if (dst_row0 == 0) h->head_nonempty = 0;
      h->head_nonempty += nz;
      h->last_batch_head_nonempty = nz;
    }
  }
  *n_out = kept_rows;
  *nnz_out = kept_nnz;
  return APSS_OK;
}

void fill_build_args(apss_handle *h, apss_handle::IndexSet &ix, BuildArgs &b, bool scaled) {
  // (a handle with a dense-head block builds from its tail view: the block's entries are not in the inverted index)
  b.rowptr = h->head_k ? h->tv.rowptr.p : h->rowptr.p;
  b.idx = h->head_k ? h->tv.idx.p : h->idx.p;
  b.val = h->head_k ? h->tv.val.p : h->val.p;
  b.cb = (int32_t)ix.cb;
  b.dim = h->cfg.dim;
  b.tile_seg = ix.seg.p;
  b.seg_stride = (int64_t)h->cfg.dim;
  b.coarse = ix.coarse ? 1 : 0;
  b.seg_align = ix.align;
  b.erow = h->head_k ? h->tv.erow.p : h->erow.p;
  b.row_scale = scaled && ix.coarse ? h->sub.p : nullptr;  // shard rule: postings normalised by |x_g| / |x| (k_probe_coarse)
  b.coarse_shift = ix.coarse && ix.cb <= 32768 ? 1 : 0;  // must agree with k_probe_coarse's SLOT2 (the 512-thread kernels)
  b.coarse_wide = ix.coarse && ix.cb > 65536 ? 1 : 0;     // ... and with its WIDE (131072-row tiles)
  b.tile_post_base = ix.base.p;
  b.post = ix.post.p;
  b.post_c = ix.post_c.p;
}

// segment lengths -> segment starts for tiles [tile0, tile0 + n): k_tile_scan_part + k_tile_scan_place
int32_t launch_tile_scan(apss_handle *h, apss_handle::IndexSet &ix, int64_t tile0, int64_t n, uint32_t keep_len, unsigned long long *chunk_w,
                         uint32_t count_long) {
  const int32_t n_blk = (int32_t)ceil_div(h->cfg.dim, kScanBlock);
  APSS_TRY(ensure(h, ix.scan_part, (size_t)(n * n_blk)));
  const dim3 grid((unsigned)(n * n_blk));
  hipLaunchKernelGGL(k_tile_scan_part, grid, dim3(1024), 0, h->stream, (const uint2 *)ix.seg.p, (int64_t)h->cfg.dim, h->cfg.dim, tile0, n_blk,
                     (uint32_t)ix.align, ix.scan_part.p, ix.maxlen.p, chunk_w, count_long);
  hipLaunchKernelGGL(k_tile_scan_place, grid, dim3(1024), 0, h->stream, ix.seg.p, (int64_t)h->cfg.dim, h->cfg.dim, tile0, n_blk, (uint32_t)ix.align,
                     keep_len, (const uint32_t *)ix.scan_part.p, ix.total.p);
  HIPCHK(h, hipGetLastError());
  return APSS_OK;
}

// ---- APPEND rows [row0, row1) to tile `tile` of `ix`, which holds rows [tile * cb, row0) (k_tile_shift, apss_kernels.hpp) ----
int32_t append_tile(apss_handle *h, apss_handle::IndexSet &ix, int64_t tile, int64_t row0, int64_t row1) {
  const int64_t stride = (int64_t)h->cfg.dim;
  const bool scaled = h->sharded || h->head_k > 0;
  const size_t elt = ix.coarse ? sizeof(uint32_t) : sizeof(Posting);
  const int64_t base = ix.h_base[(size_t)tile], old_total = ix.h_base[(size_t)tile + 1] - base;
  uint2 *sg = ix.seg.p + tile * stride;
  APSS_TRY(ensure(h, h->app_seg, (size_t)stride));
  APSS_TRY(ensure(h, h->app_post, (size_t)std::max<int64_t>(old_total, 1) * elt));
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->app_seg.p, sg, (size_t)stride * sizeof(uint2), hipMemcpyDeviceToDevice, h->stream));
  if (old_total > 0)
    HIPCHK(h, hipMemcpyAsync(h->app_post.p, ix.coarse ? (const void *)(ix.post_c.p + base) : (const void *)(ix.post.p + base),
                             (size_t)old_total * elt, hipMemcpyDeviceToDevice, h->stream));
  BuildArgs b{};
  fill_build_args(h, ix, b, scaled);
  b.row0 = row0;
  b.row1 = row1;
  const int threads = 256;
  const int64_t blocks = ceil_div((row1 - row0) * kWave, threads);
  hipLaunchKernelGGL(k_tile_hist, dim3((unsigned)blocks), dim3(threads), 0, h->stream, b);  // lengths: old + new
  APSS_TRY(launch_tile_scan(h, ix, tile, 1, 0u, nullptr, 0u));
  HIPCHK(h, hipGetLastError());
  int64_t new_total = 0;
  uint32_t seg_stats[2] = {0, 0};
  HIPCHK(h, hipMemcpyAsync(&new_total, ix.total.p + tile, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(seg_stats, ix.maxlen.p, sizeof(seg_stats), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  ix.max_seg = seg_stats[0];
  ix.h_base[(size_t)tile + 1] = base + new_total;
  ix.post_used = base + new_total;
  if (ix.coarse) APSS_TRY(ensure(h, ix.post_c, (size_t)ix.post_used + 64, (size_t)base));
  else APSS_TRY(ensure(h, ix.post, (size_t)ix.post_used + 64, (size_t)base));
  if (ix.coarse)  // the probe reads a zero word as "no posting": clear the tile's region (padding between segments)
    HIPCHK(h, hipMemsetAsync(ix.post_c.p + base, 0, (size_t)(new_total + 64) * sizeof(uint32_t), h->stream));
  HIPCHK(h, hipMemcpyAsync(ix.base.p + tile + 1, ix.h_base.data() + tile + 1, sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
  ShiftArgs sa{};
  sa.seg_old = h->app_seg.p;
  sa.seg_new = sg;
  sa.old_c = ix.coarse ? reinterpret_cast<const uint32_t *>(h->app_post.p) : nullptr;
  sa.old_x = ix.coarse ? nullptr : reinterpret_cast<const Posting *>(h->app_post.p);
  sa.new_c = ix.coarse ? ix.post_c.p + base : nullptr;
  sa.new_x = ix.coarse ? nullptr : ix.post.p + base;
  sa.dim = h->cfg.dim;
  hipLaunchKernelGGL(k_tile_shift, dim3((unsigned)ceil_div(stride * kGroup, 256)), dim3(256), 0, h->stream, sa);
  fill_build_args(h, ix, b, scaled);  // (the posting array may have moved)
  b.row0 = row0;
  b.row1 = row1;
  hipLaunchKernelGGL(k_tile_scatter, dim3((unsigned)blocks), dim3(threads), 0, h->stream, b);
  if (scaled) {
    DevBuf<float> &tmin = ix.coarse ? h->tile_min_c : h->tile_min;
    hipLaunchKernelGGL(k_tile_min_sub, dim3(1), dim3(1024), 0, h->stream, (const float *)h->sub.p, h->idx_rows, (int32_t)ix.cb, tmin.p, tile);
  }
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  HIPCHK(h, hipEventSynchronize(h->ev1));  // (the host vector the tile base was copied from lives on; the event gives the build its time)
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  ix.build_ms = ms;
  return APSS_OK;
}

// ---- index build for rows [row0, n_rows): rebuild every tile of `ix` that contains one of them ----
int32_t build_tiles(apss_handle *h, apss_handle::IndexSet &ix, int64_t row0) {
  const int64_t cb = ix.cb;
  int64_t tile0 = row0 / cb;
  const int64_t n_tiles = ceil_div(h->idx_rows, cb);
  const int64_t stride = (int64_t)h->cfg.dim;
  double appended_ms = 0.0;
  if (row0 % cb != 0 && ix.built_rows == row0 && ix.n_tiles == tile0 + 1 && (int64_t)ix.h_base.size() == tile0 + 2 &&
      !h->dbgcfg.no_append && !h->dbgcfg.bank_order) {
    // the batch lands in the last, partly filled tile of an up-to-date rendering: append to it, build the tiles beyond as ever
    DevBuf<float> &tmin_a = ix.coarse ? h->tile_min_c : h->tile_min;
    if (h->sharded || h->head_k > 0) APSS_TRY(ensure(h, tmin_a, (size_t)n_tiles, (size_t)tile0 + 1));
    APSS_TRY(ensure(h, ix.total, (size_t)n_tiles, (size_t)tile0 + 1));
    APSS_TRY(append_tile(h, ix, tile0, row0, std::min<int64_t>(h->idx_rows, (tile0 + 1) * cb)));
    appended_ms = ix.build_ms;
    ++tile0;
    row0 = tile0 * cb;
    if (h->idx_rows <= row0) {
      ix.built_rows = h->idx_rows;
      return APSS_OK;
    }
  }
  ix.n_tiles = n_tiles;
  ix.built_rows = h->idx_rows;
  if (n_tiles == tile0) return APSS_OK;
  APSS_TRY(ensure(h, ix.seg, (size_t)(n_tiles * stride), (size_t)(tile0 * stride)));
  APSS_TRY(ensure(h, ix.base, (size_t)n_tiles + 1, (size_t)tile0 + 1));
  APSS_TRY(ensure(h, ix.total, (size_t)n_tiles, 0));
  APSS_TRY(ensure(h, ix.maxlen, 2));
  APSS_TRY(ensure(h, ix.chunkw, 1));
  if (tile0 == 0) {
    HIPCHK(h, hipMemsetAsync(ix.maxlen.p, 0, 2 * sizeof(uint32_t), h->stream));
    HIPCHK(h, hipMemsetAsync(ix.chunkw.p, 0, sizeof(unsigned long long), h->stream));
  }
  DevBuf<float> &tmin = ix.coarse ? h->tile_min_c : h->tile_min;
  const bool scaled = h->sharded || h->head_k > 0;  // the probe scales its threshold per query and tile (shard rule)
  if (scaled) APSS_TRY(ensure(h, tmin, (size_t)n_tiles, (size_t)tile0));
  if (ix.h_base.empty()) ix.h_base.push_back(0);
  ix.h_base.resize((size_t)tile0 + 1);  // bases of the tiles that stay
  const int64_t r0 = tile0 * cb;
  // small dims: counters and cursors of a (tile, term range) live in one workgroup's LDS (no global atomics)
  int32_t n_ranges = (int32_t)ceil_div(h->cfg.dim, kBuildRange);
  // (one workgroup per (tile, range): with fewer than ~48 of them -- a small batch, a rebuilt tail tile, a 1/8 candidate
  // range -- the row-parallel global-atomic kernels are faster; APSS_BUILD_LDS / APSS_BUILD_ATOMIC force either)
  // large dims (more than kBuildMaxRanges ranges: vectorDim = 2^20): a build from the first tile partitions the entries by
  // term range first (k_bucket_pass), so that the LDS build's workgroups read their bucket instead of the whole tile
  const int64_t build_nnz = h->head_k ? h->tv.nnz : h->nnz;  // entries of the rows [0, idx_rows) the index is built from
  // (terms per range of a bucketed build, measured on C5's shape, build ms per step: 1024: 28.3, 2048: 23.9, 4096: 20.4, 8192: 15.6,
  // 16384: 17.4 -- against 37.5 with the global-atomic kernels; narrow ranges multiply the workgroups' fixed work)
  int32_t bucket_rt = h->dbgcfg.bucket_range > 0 ? h->dbgcfg.bucket_range : 8192;
  while (ceil_div(h->cfg.dim, bucket_rt) > kBucketMaxRanges && bucket_rt < kBuildRange) bucket_rt *= 2;
  const bool bucket_build = n_ranges > kBuildMaxRanges && ceil_div(h->cfg.dim, bucket_rt) <= kBucketMaxRanges && !h->dbgcfg.build_atomic &&
                            !h->dbgcfg.no_bucket && tile0 == 0 && n_tiles * n_ranges >= 48 && build_nnz > 0 && build_nnz <= (1LL << 32) &&
                            n_tiles * (int64_t)kBucketSlices < (1LL << 31);
  if (bucket_build) n_ranges = (int32_t)ceil_div(h->cfg.dim, bucket_rt);
  const bool lds_build = bucket_build || (n_ranges <= kBuildMaxRanges && !h->dbgcfg.build_atomic &&
                                          ((n_tiles - tile0) * n_ranges >= 48 || h->dbgcfg.build_lds));
  if (!lds_build)
    HIPCHK(h, hipMemsetAsync(ix.seg.p + tile0 * stride, 0, (size_t)((n_tiles - tile0) * stride) * sizeof(uint2), h->stream));
  BuildArgs b{};
  fill_build_args(h, ix, b, scaled);
  b.row0 = r0;
  b.row1 = h->idx_rows;
  const int threads = 256;
  const int64_t blocks = ceil_div((h->idx_rows - r0) * kWave, threads);
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  // bucketed build: tiles in GROUPS whose entries fit a bounded scratch (12 B per entry: a one-shot join of 2e9 entries would
  // otherwise spend 1.2 s allocating 24 GB to save 0.5 s of atomics).  A single group (C5's shape at N = 2M: 4.8 GB) is bucketed
  // once for both passes; several groups are bucketed again for the scatter pass (twice 43 ms at N = 10M, against 573 ms).
  int64_t group_tiles = n_tiles;
  BucketArgs bk{};
  if (bucket_build) {
    const int64_t per_tile = std::max<int64_t>(1, build_nnz / n_tiles);
    group_tiles = std::max<int64_t>(1, std::min<int64_t>(n_tiles, (int64_t)6e8 / per_tile));
    // entries of the largest group (row extents are on the device: one small read)
    std::vector<int64_t> grp_e;
    {
      std::vector<int64_t> at;
      for (int64_t g0 = 0; g0 < n_tiles; g0 += group_tiles) at.push_back(std::min<int64_t>(h->idx_rows, g0 * cb));
      at.push_back(h->idx_rows);
      grp_e.resize(at.size());
      for (size_t i = 0; i < at.size(); ++i)
        HIPCHK(h, hipMemcpyAsync(&grp_e[i], b.rowptr + at[i], sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    int64_t max_e = 1;
    for (size_t i = 0; i + 1 < grp_e.size(); ++i) max_e = std::max(max_e, grp_e[i + 1] - grp_e[i]);
    const int64_t nb = group_tiles * n_ranges;
    APSS_TRY(ensure(h, h->bk_cnt, (size_t)(2 * nb + 2)));
    APSS_TRY(ensure(h, h->bk_base, (size_t)nb + 2));
    APSS_TRY(ensure(h, h->bk_idx, (size_t)max_e));
    APSS_TRY(ensure(h, h->bk_erow, (size_t)max_e));
    APSS_TRY(ensure(h, h->bk_val, (size_t)max_e));
    bk.rowptr = b.rowptr;
    bk.idx = b.idx;
    bk.val = b.val;
    bk.erow = b.erow;
    bk.row1 = h->idx_rows;
    bk.cb = (int32_t)cb;
    bk.n_ranges = n_ranges;
    bk.range_terms = bucket_rt;
    bk.bucket_cnt = h->bk_cnt.p;
    bk.bucket_cur = h->bk_cnt.p + nb + 1;
    bk.bucket_base = h->bk_base.p;
    bk.o_idx = h->bk_idx.p;
    bk.o_erow = h->bk_erow.p;
    bk.o_val = h->bk_val.p;
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));  // (the build's clock starts behind the reservations)
  }
  // partition the entries of tiles [g0, g0 + gt) by term range into the scratch arrays (k_bucket_pass: count, scan, scatter)
  auto bucket_group = [&](int64_t g0, int64_t gt) -> int32_t {
    const int64_t nbg = gt * n_ranges;
    HIPCHK(h, hipMemsetAsync(h->bk_cnt.p, 0, (size_t)(2 * (group_tiles * n_ranges) + 2) * sizeof(unsigned long long), h->stream));
    bk.tile0 = g0;
    const dim3 bgrid((unsigned)(gt * kBucketSlices));
    hipLaunchKernelGGL(k_bucket_pass<false>, bgrid, dim3(1024), 0, h->stream, bk);
    APSS_TRY(scan_i64(h, reinterpret_cast<const int64_t *>(h->bk_cnt.p), h->bk_base.p, nbg));
    hipLaunchKernelGGL(k_bucket_pass<true>, bgrid, dim3(1024), 0, h->stream, bk);
    HIPCHK(h, hipGetLastError());
    return APSS_OK;
  };
  if (bucket_build) {
    b.idx = h->bk_idx.p;
    b.erow = h->bk_erow.p;
    b.val = h->bk_val.p;
    b.ent_base = h->bk_base.p;
    b.range_terms = bucket_rt;
    for (int64_t g0 = 0; g0 < n_tiles; g0 += group_tiles) {
      const int64_t gt = std::min<int64_t>(group_tiles, n_tiles - g0);
      APSS_TRY(bucket_group(g0, gt));
      hipLaunchKernelGGL(k_tile_hist_lds, dim3((unsigned)(gt * n_ranges)), dim3(1024), 0, h->stream, b, g0, n_ranges);
    }
    HIPCHK(h, hipGetLastError());
  }
  const dim3 lds_grid((unsigned)((n_tiles - tile0) * n_ranges));
  if (bucket_build) {}  // (counted above, group by group)
  else if (lds_build) hipLaunchKernelGGL(k_tile_hist_lds, lds_grid, dim3(1024), 0, h->stream, b, tile0, n_ranges);
  else hipLaunchKernelGGL(k_tile_hist, dim3((unsigned)blocks), dim3(threads), 0, h->stream, b);
  APSS_TRY(launch_tile_scan(h, ix, tile0, n_tiles - tile0, lds_build ? 1u : 0u, tile0 == 0 ? ix.chunkw.p : nullptr, 1u));
  HIPCHK(h, hipGetLastError());
  // padded posting counts -> tile bases (host prefix sum: a handful of values), then reserve the posting array
  std::vector<int64_t> tot((size_t)(n_tiles - tile0));
  HIPCHK(h, hipMemcpyAsync(tot.data(), ix.total.p + tile0, tot.size() * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
  uint32_t seg_stats[2] = {0, 0};
  unsigned long long chunk_w = 0;
  HIPCHK(h, hipMemcpyAsync(seg_stats, ix.maxlen.p, sizeof(seg_stats), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(&chunk_w, ix.chunkw.p, sizeof(chunk_w), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  ix.max_seg = seg_stats[0];
  ix.long_segs = seg_stats[1];
  // (measured on a build from the first tile; an appended batch keeps the figure of the build before it)
  if (tile0 == 0 && h->idx_rows > 0) ix.round_chunks = (double)chunk_w / (double)h->idx_rows;
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
  if (bucket_build) {
    for (int64_t g0 = 0; g0 < n_tiles; g0 += group_tiles) {
      const int64_t gt = std::min<int64_t>(group_tiles, n_tiles - g0);
      if (group_tiles < n_tiles) APSS_TRY(bucket_group(g0, gt));  // (a single group's buckets are still there)
      hipLaunchKernelGGL(k_tile_scatter_lds, dim3((unsigned)(gt * n_ranges)), dim3(1024), 0, h->stream, b, g0, n_ranges);
    }
  } else if (lds_build) hipLaunchKernelGGL(k_tile_scatter_lds, lds_grid, dim3(1024), 0, h->stream, b, tile0, n_ranges);
  else hipLaunchKernelGGL(k_tile_scatter, dim3((unsigned)blocks), dim3(threads), 0, h->stream, b);
  if (ix.coarse && ix.cb <= 32768 && h->dbgcfg.bank_order)
    hipLaunchKernelGGL(k_seg_bank_order, dim3((unsigned)ceil_div((n_tiles - tile0) * stride * kWave, 256)), dim3(256), 0, h->stream,
                       (const uint2 *)ix.seg.p, stride, (const int64_t *)ix.base.p, ix.post_c.p, tile0, n_tiles, h->cfg.dim);
  if (scaled)
    hipLaunchKernelGGL(k_tile_min_sub, dim3((unsigned)(n_tiles - tile0)), dim3(1024), 0, h->stream,
                       (const float *)h->sub.p, h->idx_rows, (int32_t)cb, tmin.p, tile0);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  HIPCHK(h, hipEventSynchronize(h->ev1));
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  ix.build_ms = ms + appended_ms;
  return APSS_OK;
}

// Which renderings of the index a handle keeps: the coarse one (two-pass speed path) is built at insert time when the
// handle may use it; the exact one is built at insert time otherwise, or lazily by the first probe that needs it.
int32_t ensure_exact_index(apss_handle *h) {
  if (h->ex_built_rows < h->idx_rows || h->ex.n_tiles == 0) {
    APSS_TRY(build_tiles(h, h->ex, std::min(h->ex_built_rows, h->idx_rows)));
    h->ex_built_rows = h->idx_rows;
    h->st.build_ms += h->ex.build_ms;
  }
  return APSS_OK;
}

// 8-bit accumulator scale of a term shard: the largest S = 2^k <= 2^7 with  S * max|q||c| * 1.0005 + shared terms < 2^8  (no
// carry into the neighbour's byte), taken only while the threshold in units stays well above what a chance pair collects
// (one unit of round-up per shared term).  0: not usable.
double acc8_scale(double bound, double shared, double theta) {
  for (int k = 7; k >= 5; --k) {
    const double S = std::ldexp(1.0, k);
    if (bound * 1.0005 * S < 255.0 - shared && std::floor(theta * S * (1.0 - 1.0 / 2048 - 1e-6)) - 2 >= 12.0) return S;
  }
  return 0.0;
}

// ---- dense-head block (apss_head.hpp) ----
constexpr int64_t kTailMaxRows = 1024;   // rows that may wait outside the tile index (scored pair by pair by k_tail_score; 4096 until the append build made folding them in cheap)
constexpr int64_t kTailMaxBatch = 64;    // a batch larger than this extends the index right away (256 until the append build)
constexpr int64_t kTailMaxPairs = 1 << 22;  // (queries x tail rows) a probe scores directly; beyond it the tail is folded in first
constexpr int32_t kHeadMaxTerms = kHeadBlock * (1 + kHeadMaxFold);   // 256 terms with a column each + 256 columns of kHeadMaxFold terms
// width of a W row holding n_terms head terms: one block of 64 | 128 | 256 columns, or 256 + a folded block of fold_w columns
inline int32_t head_width(int32_t n_terms, int32_t fold_w, int32_t exact = kHeadBlock, bool i8 = false) {
  // (INT8 rendering: a 64-column row would be 64 bytes, less than a 1-KiB DMA piece per wave and tile: 128 columns at least)
  return n_terms <= 64 ? (i8 ? 128 : 64) : (n_terms <= 128 ? 128 : (n_terms <= 256 ? 256 : exact + fold_w));
}
// column of the i-th head term (most frequent first): the first `exact` get a column each, the others fold into fold_w columns
inline int32_t head_column(int32_t i, int32_t fold_w, int32_t exact = kHeadBlock) { return i < exact ? i : exact + (i - exact) % fold_w; }
constexpr int64_t kHeadMinRows = 16384;  // below this a join is over before a GEMM pays for its set-up
constexpr double kHeadSparseRate = 8.0e11;  // posting visits / s of the sparse filter (measured, C3)
// seconds per (query, candidate) element of the head contraction, block width 64 / 128 / 256 (measured on random rows,
// profiles/r02_head_gemm.md: 1.65 PFLOP/s at 256; narrower blocks are bound by the epilogue's scan, not by the MFMAs)
constexpr double kHeadDenseCost[4] = {1.5e-13, 1.9e-13, 3.1e-13, 3.1e-13};  // (last: a head of more than 256 terms -- still ONE block of 256 columns)
// ... with the INT8 rendering (v_mfma_i32_32x32x32_i8, rows half as wide; round 4: C3-Zipf(1) at N = 1M 155 -> 88 ms, 1.75e-13 per
// element at 256 columns; a block of up to 64 terms takes 128 byte-wide columns)
constexpr double kHeadDenseCostI8[4] = {1.2e-13, 1.2e-13, 1.75e-13, 1.75e-13};
constexpr double kHeadFoldMaxRowTerms = 96.0;  // terms of the folded block a row may hold on average (chance pairs collide in m^2 / 256 columns)
constexpr double kHeadSurvivorCost = 2.5e-9;  // seconds per element the dense filter passes on (report + de-dup + exact re-score; measured in
                                              // round 3: 5.6e6 more survivors cost 5 ms, i.e. 0.9e-9 each; the sample's TRUE pairs count too)

// A plain handle decides for itself (choose_head) unless the block's terms were set through apss_set_head_terms; a term
// shard only ever takes the terms it was given -- the {H, T_1 .. T_T} partition of the shard rule must be the same on every
// shard of the join, so the caller that cut the term ranges names the block too.
inline bool head_allowed(const apss_handle *h) {
  return h->use_coarse && (!h->sharded || h->head_fixed) && (h->cfg.head_terms >= 0 || h->head_fixed) && h->cfg.theta > 0.0 &&
         !h->head_blocked && h->nonneg;
}

// Which rendering the block's rows take.  INT8 (rounded up: a sound filter for the non-negative weights a block needs anyway)
// for a single block of 128 | 256 columns; bf16 for the two-block experiment forms and on request (APSS_DEBUG=head_bf16).
inline bool head_wants_i8(const apss_handle *h) {
  return !h->dbgcfg.head_bf16 && !(h->dbgcfg.fold_w == 128 || h->dbgcfg.fold_w == 256);
}
// (re)decide the rendering for a block of width head_k whose FIRST row is about to be packed
int32_t head_setup_rendering(apss_handle *h) {
  h->head_i8 = h->head_k > 0 && h->head_k <= kHeadBlock && head_wants_i8(h);
  h->head_s = 0.f;
  APSS_TRY(ensure(h, h->head_ovf, 4));
  HIPCHK(h, hipMemsetAsync(h->head_ovf.p, 0, 4 * sizeof(unsigned int), h->stream));
  return APSS_OK;
}
// INT8 scale S (units per 1.0).  No element of a W row exceeds its row's norm (exact columns hold c_t |c| / |c_H| <= |c|, folded
// columns the L2 norm of their terms, scaled alike), so S = 127 / (the largest row norm) never overflows a byte.  A row with a
// larger norm than any before (un-normalised input, a stream) shrinks S for the WHOLE block -- one threshold serves every pair --
// and the `packed_bytes` of W packed so far are re-quantised in place (k_head_rescale: still upper bounds).
int32_t head_fit_scale(apss_handle *h, double norm2, unsigned char *W, int64_t packed_bytes) {
  if (!h->head_i8) return APSS_OK;
  const double need = 127.0 / (std::sqrt(std::max(norm2, 1e-30)) * 1.0005);
  if (h->head_s > 0.f && need >= (double)h->head_s) return APSS_OK;
  if (h->head_s > 0.f && packed_bytes > 0) {
    const double target = need / 1.25;  // (room for norms a quarter larger before the next pass over W)
    const uint32_t den = 65535u;
    const uint32_t num = (uint32_t)std::min<double>(den, std::ceil(target / (double)h->head_s * den));
    hipLaunchKernelGGL(k_head_rescale, dim3((unsigned)std::min<int64_t>(8192, ceil_div(packed_bytes, 16 * 256))), dim3(256), 0, h->stream, W,
                       packed_bytes / 16 * 16, num, den);
    HIPCHK(h, hipGetLastError());
    h->head_s = (float)target;
  } else {
    h->head_s = (float)need;
  }
  return APSS_OK;
}
inline void head_pack_rendering(const apss_handle *h, HeadPackArgs &p) {
  p.i8_scale = h->head_i8 ? h->head_s : 0.f;
  p.overflow = h->head_ovf.p;
}

// tail view of rows [row0, row1) of a CSR batch (absolute offsets), appended to `v` as its rows [dst_row0, ..)
int32_t build_tail_view(apss_handle *h, apss_handle::TailView &v, const int64_t *rowptr, const int32_t *idx, const float *val,
                        int64_t row0, int64_t row1, int64_t dst_row0, bool with_erow) {
  const int64_t n = row1 - row0;
  if (dst_row0 == 0) v.rows = v.nnz = v.max_nnz = v.nonempty = 0;
  v.last_nonempty = 0;
  APSS_TRY(ensure(h, v.rowptr, (size_t)(dst_row0 + n + 1), (size_t)(dst_row0 ? dst_row0 + 1 : 0)));
  if (dst_row0 == 0) HIPCHK(h, hipMemsetAsync(v.rowptr.p, 0, sizeof(int64_t), h->stream));
  if (n <= 0) return APSS_OK;
  APSS_TRY(ensure(h, h->tv_cnt, (size_t)n + 1));
  APSS_TRY(ensure(h, h->tv_off, (size_t)n + 1));
  APSS_TRY(ensure(h, h->tv_sum, 2));
  HIPCHK(h, hipMemsetAsync(h->tv_sum.p, 0, 2 * sizeof(unsigned int), h->stream));
  TailViewArgs a{};
  a.rowptr = rowptr;
  a.idx = idx;
  a.val = val;
  a.row0 = row0;
  a.row1 = row1;
  a.head_pos = h->head_pos.p;
  a.cnt = h->tv_cnt.p;
  a.summary = h->tv_sum.p;
  const unsigned blocks = (unsigned)ceil_div(n * kGroup, 256);
  hipLaunchKernelGGL(k_tailv_count, dim3(blocks), dim3(256), 0, h->stream, a);
  HIPCHK(h, hipGetLastError());
  APSS_TRY(scan_i64(h, (const int64_t *)h->tv_cnt.p, h->tv_off.p, n));
  int64_t total = 0;
  unsigned int sum[2] = {0, 0};
  HIPCHK(h, hipMemcpyAsync(&total, h->tv_off.p + n, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(sum, h->tv_sum.p, sizeof(sum), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const int64_t nnz0 = v.nnz;
  APSS_TRY(ensure(h, v.idx, (size_t)std::max<int64_t>(nnz0 + total, 1), (size_t)nnz0));
  APSS_TRY(ensure(h, v.val, (size_t)std::max<int64_t>(nnz0 + total, 1), (size_t)nnz0));
  if (with_erow) APSS_TRY(ensure(h, v.erow, (size_t)std::max<int64_t>(nnz0 + total, 1), (size_t)nnz0));
  a.off = h->tv_off.p;
  a.dst_row0 = dst_row0;
  a.dst_nnz0 = nnz0;
  a.o_rowptr = v.rowptr.p;
  a.o_idx = v.idx.p;
  a.o_val = v.val.p;
  a.o_erow = with_erow ? v.erow.p : nullptr;
  hipLaunchKernelGGL(k_tailv_write, dim3(blocks), dim3(256), 0, h->stream, a);
  HIPCHK(h, hipGetLastError());
  v.rows = dst_row0 + n;
  v.nnz = nnz0 + total;
  v.max_nnz = std::max<int64_t>(v.max_nnz, sum[0]);
  v.nonempty += sum[1];
  v.last_nonempty = sum[1];
  return APSS_OK;
}

// W rows, tail ratios and the masked term array for store rows [row0, n_rows)
int32_t head_pack_store(apss_handle *h, int64_t row0) {
  const int64_t kh = h->head_k;
  const int64_t rows_pad = ceil_div(h->idx_rows, kHeadQBlock) * kHeadQBlock + kHeadCTile;
  APSS_TRY(ensure(h, h->W, (size_t)(rows_pad * kh), (size_t)(ceil_div(row0, kHeadCTile) * kHeadCTile * kh)));  // (tiled: whole tiles)
  APSS_TRY(ensure(h, h->sub, (size_t)h->idx_rows, (size_t)row0));
  APSS_TRY(ensure(h, h->head_ctr, 4));
  HIPCHK(h, hipMemsetAsync(h->head_ctr.p, 0, 4 * sizeof(unsigned long long), h->stream));
  if (row0 == 0) APSS_TRY(head_setup_rendering(h));
  APSS_TRY(head_fit_scale(h, (double)h->store_max_norm2, reinterpret_cast<unsigned char *>(h->W.p), ceil_div(row0, kHeadCTile) * kHeadCTile * kh));
  {
    HeadPackArgs a{};
    head_pack_rendering(h, a);
    a.rowptr = h->rowptr.p;
    a.idx = h->idx.p;
    a.val = h->val.p;
    a.row0 = row0;
    a.row1 = h->idx_rows;
    a.head_pos = h->head_pos.p;
    a.kh = (int32_t)kh;
    a.W = h->W.p;
    a.w_row0 = row0;
    a.w_pad = rows_pad;  // the rows past the last one are zero rows: a GEMM tile or query block may read them
    a.ratio_t = h->sub.p;
    a.head_nonempty = reinterpret_cast<unsigned int *>(h->head_ctr.p);
    a.row_inv = nullptr;
    a.prune_above = -INFINITY;  // (the store holds the rows as they are scored)
    a.fold_from = h->head_exact;
    hipLaunchKernelGGL(k_head_pack, dim3((unsigned)(ceil_div(rows_pad, 8) - row0 / 8)), dim3(512), 0, h->stream, a);
  }
  HIPCHK(h, hipGetLastError());
  unsigned int nz = 0;
  HIPCHK(h, hipMemcpyAsync(&nz, h->head_ctr.p, sizeof(nz), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (row0 == 0) h->head_nonempty = 0;
  h->head_nonempty += nz;
  h->last_batch_head_nonempty = nz;
  return APSS_OK;
}

int32_t run_head(apss_handle *h, const ProbeArgs &a, int64_t nq, int64_t q_slot_base, const uint16_t *Wq, int64_t wq_rows,
                 float thr, double bound, int64_t n_cand);

// How selective is the dense filter on THIS data: the block's first rows (at most 8192) against its first 512 as queries,
// stored and re-scored.  Returns the fraction of (q, c != q) elements the filter passes that do NOT reach theta.  (Rows arrive in no
// particular order in the reference's stream -- ShardRegion routing is random, CommonUtils.scala:28-40 -- so a prefix is a
// sample.)
int32_t head_sample_selectivity(apss_handle *h, double *frac) {
  *frac = 0.0;
  const int64_t kh = h->head_k;
  const int64_t S = std::min<int64_t>(h->idx_rows, 8192), Q = std::min<int64_t>(S, 512);
  const int64_t rows_pad = ceil_div(h->idx_rows, kHeadQBlock) * kHeadQBlock + kHeadCTile;
  APSS_TRY(ensure(h, h->W, (size_t)(rows_pad * kh), 0));
  APSS_TRY(ensure(h, h->sub, (size_t)h->idx_rows, 0));
  APSS_TRY(ensure(h, h->head_ctr, 4));
  APSS_TRY(ensure(h, h->counters, kCtrCount));
  HIPCHK(h, hipMemsetAsync(h->head_ctr.p, 0, 4 * sizeof(unsigned long long), h->stream));
  APSS_TRY(head_setup_rendering(h));
  APSS_TRY(head_fit_scale(h, (double)h->store_max_norm2, nullptr, 0));
  HeadPackArgs p{};
  head_pack_rendering(h, p);
  p.rowptr = h->rowptr.p;
  p.idx = h->idx.p;
  p.val = h->val.p;
  p.row0 = 0;
  p.row1 = S;
  p.head_pos = h->head_pos.p;
  p.kh = (int32_t)kh;
  p.W = h->W.p;
  p.w_row0 = 0;
  p.w_pad = ceil_div(S, kHeadQBlock) * kHeadQBlock;
  p.ratio_t = h->sub.p;
  p.head_nonempty = nullptr;
  p.row_inv = nullptr;
  p.prune_above = -INFINITY;
  p.fold_from = h->head_exact;
  hipLaunchKernelGGL(k_head_pack, dim3((unsigned)ceil_div(p.w_pad, 8)), dim3(512), 0, h->stream, p);
  HIPCHK(h, hipGetLastError());
  // the sample's survivors are STORED and re-scored exactly: what counts against the block is what it passes in excess -- a
  // pair that reaches theta is a survivor of every filter (the sample of a batch with planted near-duplicates holds 3e-5 of
  // them: counted against the block they vetoed a head that halves the step of C3 with Zipf(0.5) terms)
  constexpr size_t kSampleCap = 1 << 20;
  APSS_TRY(ensure(h, h->res_q, kSampleCap));
  APSS_TRY(ensure(h, h->res_c, kSampleCap));
  APSS_TRY(ensure(h, h->res_s, kSampleCap));
  ProbeArgs a{};
  a.q_ext = h->ext.p;
  a.res_q = h->res_q.p;
  a.res_c = h->res_c.p;
  a.res_s = h->res_s.p;
  a.res_cap = kSampleCap;
  a.counters = h->head_ctr.p + 2;
  const double bound = (double)h->store_max_norm2 * 1.0001 + 1e-6;  // |q||c| <= the largest squared row norm
  APSS_TRY(run_head(h, a, Q, -1, h->W.p, p.w_pad, (float)(h->cfg.theta - 0.0080 * bound - 1e-5), bound, S));
  unsigned long long c[4];
  HIPCHK(h, hipMemcpyAsync(c, h->head_ctr.p, sizeof(c), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const double pairs = (double)Q * (double)S - (double)Q;
  unsigned long long n_true = 0;
  if (c[2] > 0 && c[2] <= kSampleCap) {
    APSS_TRY(ensure(h, h->fin_q, (size_t)c[2]));
    APSS_TRY(ensure(h, h->fin_c, (size_t)c[2]));
    APSS_TRY(ensure(h, h->fin_s, (size_t)c[2]));
    HIPCHK(h, hipMemsetAsync(h->head_ctr.p, 0, sizeof(unsigned long long), h->stream));
    RescoreArgs r{};
    r.n_pairs = (int64_t)c[2];
    r.q_row = h->res_q.p;
    r.c_slot = h->res_c.p;
    r.q_rowptr = h->rowptr.p;  // (the sample's queries are the store's first rows)
    r.q_idx = h->idx.p;
    r.q_val = h->val.p;
    r.c_rowptr = h->rowptr.p;
    r.c_idx = h->idx.p;
    r.c_val = h->val.p;
    r.theta = (float)h->cfg.theta;
    r.out_q = h->fin_q.p;
    r.out_c = h->fin_c.p;
    r.out_s = h->fin_s.p;
    r.out_count = h->head_ctr.p;
    hipLaunchKernelGGL(k_rescore, dim3((unsigned)std::max<int64_t>(1, ceil_div(r.n_pairs, (int64_t)kRescorePairs * kRescoreTrips))), dim3(kRescoreBlock), 0, h->stream, r);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(&n_true, h->head_ctr.p, sizeof(n_true), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  *frac = pairs > 0 ? (double)(c[2] - std::min(n_true, c[2])) / pairs : 0.0;
  return APSS_OK;
}

// Look at the store's term distribution (sampled document frequencies) and decide which terms, if any, go to the
// dense block: term t costs df_t^2 posting visits in the inverted index and 2 * N^2 flop as a column of the
// contraction (half of that for a stored batch: the product is symmetric).  *changed: the block's term set differs.
int32_t choose_head(apss_handle *h, bool *changed) {
  *changed = false;
  const int64_t n = h->n_rows;
  const int32_t dim = h->cfg.dim;
  h->head_eval_rows = n;
  const int64_t stride = std::max<int64_t>(1, n / 65536);
  const int64_t sampled = ceil_div(n, stride);
  APSS_TRY(ensure(h, h->df, (size_t)dim));
  HIPCHK(h, hipMemsetAsync(h->df.p, 0, (size_t)dim * sizeof(uint32_t), h->stream));
  hipLaunchKernelGGL(k_df_sample, dim3((unsigned)ceil_div(sampled * kWave, 256)), dim3(256), 0, h->stream,
                     (const int64_t *)h->rowptr.p, (const int32_t *)h->idx.p, n, stride, h->df.p);
  HIPCHK(h, hipGetLastError());
  std::vector<uint32_t> df((size_t)dim);
  HIPCHK(h, hipMemcpyAsync(df.data(), h->df.p, (size_t)dim * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  std::vector<int32_t> order((size_t)dim);
  for (int32_t t = 0; t < dim; ++t) order[(size_t)t] = t;
  const size_t top = (size_t)std::min<int32_t>(dim, kHeadMaxTerms);
  const auto more_frequent = [&](int32_t x, int32_t y) { return df[(size_t)x] != df[(size_t)y] ? df[(size_t)x] > df[(size_t)y] : x < y; };
  bool hopeless = false;
  if (h->cfg.head_terms == 0) {
    // before sorting anything: no head of kHeadMaxTerms terms can save more visits than min(all terms', that many of the most
    // frequent one's) -- on uniform data (C3: every benchmark step re-evaluates after its apss_clear) that is below the
    // cheapest block's cost and the 3 ms of sorting 100k terms are not spent
    double s2_all = 0.0, f_max = 0.0;
    for (int32_t t = 0; t < dim; ++t) {
      const double f = (double)df[(size_t)t] / (double)sampled;
      s2_all += f * f;
      f_max = std::max(f_max, f);
    }
    // (per block size: k terms save at most min(all terms', k x the most frequent one's) visits against THAT block's cost -- a
    // single comparison with the cheapest block's cost stopped holding on C3 when the INT8 rendering made that block cheaper)
    const double *cost = head_wants_i8(h) ? kHeadDenseCostI8 : kHeadDenseCost;
    const double terms_of[4] = {64.0, 128.0, 256.0, (double)top};
    hopeless = true;
    for (int ki = 0; ki < 4; ++ki)
      if (std::min(s2_all, terms_of[ki] * f_max * f_max) / kHeadSparseRate >= 1.5 * 0.5 * cost[ki]) hopeless = false;
  }
  if (!hopeless) {
    std::nth_element(order.begin(), order.begin() + (ptrdiff_t)top - 1, order.end(), more_frequent);
    std::sort(order.begin(), order.begin() + (ptrdiff_t)top, more_frequent);
  }
  int32_t k = 0;
  double best_gain = 0.0;  // seconds per N^2 pairs the chosen block is expected to save
  double gain256 = 0.0;    // ... and a plain 256-term block, the fall-back when the folded block proves unselective
  if (h->cfg.head_terms > 0) {
    k = std::min(h->cfg.head_terms <= 256 ? head_width(h->cfg.head_terms, 0) : h->cfg.head_terms, kHeadMaxTerms);  // terms wanted
  } else if (n >= kHeadMinRows && !hopeless) {
    double best = 0.0, s2 = 0.0;
    size_t i = 0;
    int ki = 0;
    double m_fold = 0.0;  // average number of folded-block terms per row
    for (int32_t kk : {64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768}) {
      if (kk > kHeadMaxTerms || (size_t)kk > top + 255) break;
      for (; i < std::min<size_t>(top, (size_t)kk); ++i) {
        const double f = (double)df[(size_t)order[i]] / (double)sampled;
        s2 += f * f;
        if (i >= (size_t)h->head_exact) m_fold += f;
      }
      if (m_fold > kHeadFoldMaxRowTerms) break;
      // per N^2 pairs of a stored batch: df_t^2 = f_t^2 N^2 visits saved; half of the product computed (symmetric).  A head
      // of more than 256 terms costs ONE more contraction however many terms fold into it
      const double save = s2 / kHeadSparseRate, cost = 0.5 * (head_wants_i8(h) ? kHeadDenseCostI8 : kHeadDenseCost)[std::min(ki++, 3)];
      if (save >= 1.5 * cost && kk == kHeadBlock) gain256 = save - cost;
      if (save >= 1.5 * cost && save - cost > best) {
        best = save - cost;
        k = kk;
      }
    }
    best_gain = best;
  }
  if (std::min(k, kHeadBlock) > dim) k = 0;
  std::vector<int32_t> terms;
  for (size_t i = 0; i < std::min<size_t>(top, (size_t)k); ++i)
    if (df[(size_t)order[i]] > 0) terms.push_back(order[i]);
  const int32_t old_k = h->head_k;
  const std::vector<int32_t> old_terms = h->head_terms;
  auto upload_columns = [&]() -> int32_t {
    std::vector<int32_t> pos((size_t)dim, -1);
    for (size_t i = 0; i < h->head_terms.size(); ++i) pos[(size_t)h->head_terms[i]] = head_column((int32_t)i, h->head_fold_w, h->head_exact);
    APSS_TRY(ensure(h, h->head_pos, (size_t)dim));
    HIPCHK(h, hipMemcpyAsync(h->head_pos.p, pos.data(), (size_t)dim * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));  // `pos` goes out of scope
    return APSS_OK;
  };
  for (;;) {
    k = terms.empty() ? 0 : head_width((int32_t)terms.size(), h->head_fold_w, h->head_exact, head_wants_i8(h));  // from here on: the width of a W row
    const bool differs = k != old_k || terms != old_terms;
    h->head_k = k;
    h->head_terms = terms;
    if (k && differs) APSS_TRY(upload_columns());
    if (!(k && differs && h->cfg.head_terms == 0)) break;  // (a block that is already in use has passed the test below)
    // the visit count says yes; now the other side of the ledger: every element the dense filter passes is reported,
    // de-duplicated and re-scored (measured: ~10 ns each).  At a low threshold on strongly skewed data that can be a
    // fraction of a percent of N^2 -- more than the posting visits saved (C2: theta = 0.5).  Measure it on a sample.
    double frac = 0.0;
    APSS_TRY(head_sample_selectivity(h, &frac));
    h->head_sample_frac = frac;
    if (frac * kHeadSurvivorCost <= 0.5 * best_gain) break;
    if (terms.size() > (size_t)kHeadBlock && gain256 > 0.0) {  // the folded block passes too much here: try the 256 terms alone
      terms.resize((size_t)kHeadBlock);
      best_gain = gain256;
      continue;
    }
    h->head_k = 0;
    h->head_terms.clear();
    break;
  }
  *changed = h->head_k != old_k || h->head_terms != old_terms;
  return APSS_OK;
}

int32_t build_index(apss_handle *h, int64_t row0) {
  h->st.build_ms = 0;
  row0 = std::min(row0, h->idx_rows);  // rows waiting in the tail are folded in with this batch
  h->idx_rows = h->n_rows;
  if (head_allowed(h)) {
    if (!h->head_fixed && h->n_rows >= std::max<int64_t>(h->cfg.head_terms > 0 ? 1 : kHeadMinRows, 2 * h->head_eval_rows)) {
      bool changed = false;
      APSS_TRY(choose_head(h, &changed));
      if (changed) {  // every tile's posting lists change with the term set: rebuild from the first row
        row0 = 0;
        h->cx.n_tiles = 0;
        h->ex_built_rows = 0;
      }
    }
    if (h->head_k) {
      h->cx.cb = std::min(h->cx.cb, 65536);  // the sparse half runs a 512-thread shard-rule kernel
      if (!h->sharded) APSS_TRY(head_pack_store(h, row0));  // (a term shard's W and ratios: ingest, from the caller's whole rows)
      const int64_t tv0 = row0 == h->tv.rows ? row0 : 0;  // (append, or start over when the view is not exactly the rows before)
      APSS_TRY(build_tail_view(h, h->tv, h->rowptr.p, h->idx.p, h->val.p, tv0, h->idx_rows, tv0, true));
    }
  } else if (h->head_k && h->sharded) {
    // a shard cannot leave the partition its peers were given: the block's terms are in no shard's index
    return fail(h, APSS_E_UNSUPPORTED, "a term shard with a dense-head block needs non-negative weights (apss_set_head_terms)");
  } else if (h->head_k) {  // e.g. a negative weight arrived: back to the plain index
    h->downgrades |= APSS_DOWNGRADE_HEAD;
    h->head_k = 0;
    h->head_terms.clear();
    row0 = 0;
    h->cx.n_tiles = 0;
    h->ex_built_rows = 0;
  }
  if (h->use_coarse) {
    if (h->cx.n_tiles == 0 && h->cfg.tile_rows == 0 && !h->dbgcfg.cx_tile && h->idx_rows > 0) {
      // a round's cost is mostly fixed, so what matters is how many postings a (tile, term) segment holds:
      // rows_per_tile * nnz_per_row / dim.  Below ~16 at 32768 rows (C5 shape: 6.5) the 65536-row tile with one
      // 1024-thread workgroup per CU wins (C5 shape at N=2M: 647 vs 790 ms); at C3 (33) two workgroups per CU win.
      const double seg32 = 32768.0 * ((double)h->nnz / (double)h->n_rows) / (double)h->cfg.dim;
      h->cx.cb = seg32 < 16.0 && !h->sharded && !h->head_k ? 65536 : 32768;  // (the 1024-thread kernel has no shard variant)
      // sparser still (C5: 6.5): 131072-row tiles with 8-bit accumulators (k_probe_coarse<1024, .., ACC8>) when the norms and
      // row lengths leave room for them: half the segment-descriptor look-ups and half the half-empty posting lines
      if (h->cx.cb == 65536 && seg32 < 8.0 && h->nonneg && !h->dbgcfg.no_acc8 && !h->no_acc8 && h->store_max_nnz <= 512 &&
          acc8_scale((double)h->store_max_norm2 * 1.0001 + 1e-6, (double)h->store_max_nnz, h->cfg.theta) > 0)
        h->cx.cb = 131072;
      // a term shard's rounds are thin (1/T of every query's terms): 8-bit accumulators hold 65536 candidates in the same
      // 64 KB, i.e. half the rounds at the same two workgroups per CU -- when the norms and row lengths leave room for them
      // (with a dense-head block: only while the tail has no long segments -- the prefetched long-segment sweeps exist for
      // 16-bit accumulators over 32768-row tiles only; known after the build, so an optimistic first build may be repeated)
      if ((h->sharded || h->head_k) && !(h->head_k && h->head_longseg) && !h->dbgcfg.no_acc8 && !h->no_acc8 &&
          acc8_scale((double)h->store_max_norm2 * 1.0001 + 1e-6, (double)h->store_max_nnz, h->cfg.theta) > 0)
        h->cx.cb = 65536;
    }
    APSS_TRY(build_tiles(h, h->cx, row0));
    h->st.build_ms += h->cx.build_ms;
    if (h->head_k && h->cfg.tile_rows == 0 && !h->dbgcfg.cx_tile) {
      if (h->cx.cb > 32768 && h->cx.max_seg > (uint32_t)kLongLenW) {
        h->head_longseg = true;
        h->cx.cb = 32768;
        h->cx.n_tiles = 0;
        APSS_TRY(build_tiles(h, h->cx, 0));
        h->st.build_ms += h->cx.build_ms;
      } else if (h->cx.cb <= 32768 && h->head_longseg && row0 == 0 && 2 * h->cx.max_seg <= (uint32_t)kLongLenW) {
        h->head_longseg = false;  // (this data would have fitted: the next first build tries the wide tiles again)
      }
    }
    h->ex_built_rows = std::min(h->ex_built_rows, row0 / h->ex.cb * h->ex.cb);  // exact tiles from here on are stale
  } else {
    h->ex_built_rows = std::min(h->ex_built_rows, row0);
    APSS_TRY(ensure_exact_index(h));
  }
  h->n_tiles = ceil_div(h->idx_rows, h->ex.cb);
  return APSS_OK;
}

template <int MODE, bool FX, int BLOCK = kProbeBlock>
int32_t launch_probe(apss_handle *h, const ProbeArgs &a, size_t lds) {
  auto kern = k_probe<MODE, BLOCK, FX>;
  HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)a.n_tiles * a.n_chunks)), dim3(BLOCK), lds, h->stream, a);
  HIPCHK(h, hipGetLastError());
  return APSS_OK;
}
// theta <= 0 (every touched candidate is an answer): the round is a chain of barriers and latencies (SQ counters, round 4:
// 73 % of the wave-cycles waiting, LDS issue 0.39, VALU 0.34), and the 1024-thread workgroup over a 16384-row tile holds 93 KB
// of LDS: ONE workgroup per CU.  512 threads over tiles of <= 8192 rows hold 58 KB: two per CU
inline int gen_block(int mode, int cb) { return mode == 2 && cb <= 8192 ? 512 : kProbeBlock; }

// ---- the filter kernel's instantiations: (threads, register-window steps, shard rule, postings per chunk, virtual
// rows, signed weights).  One table, one lookup: a combination that is not listed is an error, never another kernel.
#define APSS_CX_VARIANTS(X)                          \
  X(512, 5, false, 16, false, false, false, false)   \
  X(512, 4, false, 16, false, false, false, false)   \
  X(512, 3, false, 16, false, false, false, false)   \
  X(512, 2, false, 16, false, false, false, false)   \
  X(512, 5, true, 16, false, false, false, false)    \
  X(512, 4, true, 16, false, false, false, false)    \
  X(512, 3, true, 16, false, false, false, false)    \
  X(512, 2, true, 16, false, false, false, false)    \
  X(512, 5, true, 16, false, false, false, true)     \
  X(512, 4, true, 16, false, false, false, true)     \
  X(512, 3, true, 16, false, false, false, true)     \
  X(512, 2, true, 16, false, false, false, true)     \
  X(512, 5, false, 16, false, false, false, true)    \
  X(512, 4, false, 16, false, false, false, true)    \
  X(512, 3, false, 16, false, false, false, true)    \
  X(512, 2, false, 16, false, false, false, true)    \
  X(512, 5, true, 16, false, false, true, false)     \
  X(512, 5, false, 16, false, false, true, false)    \
  X(512, 5, true, 16, true, false, true, false)      \
  X(512, 4, false, 8, false, false, false, false)    \
  X(512, 5, false, 16, true, false, false, false)    \
  X(512, 5, false, 16, false, true, false, false)    \
  X(512, 5, true, 16, false, true, false, false)     \
  X(512, 5, false, 16, true, true, false, false)     \
  X(1024, 5, false, 16, false, false, false, false)  \
  X(1024, 3, false, 16, false, false, false, false)  \
  X(1024, 5, false, 16, true, false, false, false)   \
  X(1024, 3, false, 16, true, false, false, false)   \
  X(1024, 5, false, 16, false, true, false, false)   \
  X(1024, 3, false, 16, false, true, false, false)   \
  X(1024, 5, false, 16, false, false, false, true)   \
  X(1024, 3, false, 16, false, false, false, true)

struct CxVariant {
  int block, u;
  bool shard;
  int chunk;
  bool vrows, sgn;
  bool longpf;  // prefetched long-segment sweeps: the sparse half of a handle with a dense-head block
  bool acc8;    // 8-bit accumulators over 65536-row tiles: thin rounds of a term shard
  bool even;    // k_probe_even: a round is staged by ceil(longest query / 64) waves and its chunks dealt out evenly
  bool merge;   // k_probe_even<.., MERGE>: neighbouring query rows share a round (term shards)
};

// k_probe_even's instantiations: (threads, window steps, shard rule, signed weights, 8-bit accumulators)
#define APSS_EVEN_VARIANTS(X)          \
  X(512, 7, false, false, false)       \
  X(512, 6, false, false, false)       \
  X(512, 5, false, false, false)       \
  X(512, 4, false, false, false)       \
  X(512, 3, false, false, false)       \
  X(512, 2, false, false, false)       \
  X(512, 7, true, false, false)        \
  X(512, 6, true, false, false)        \
  X(512, 5, true, false, false)        \
  X(512, 4, true, false, false)        \
  X(512, 3, true, false, false)        \
  X(512, 2, true, false, false)        \
  X(512, 7, true, false, true)         \
  X(512, 6, true, false, true)         \
  X(512, 5, true, false, true)         \
  X(512, 4, true, false, true)         \
  X(512, 3, true, false, true)         \
  X(512, 2, true, false, true)         \
  X(512, 4, false, false, true)        \
  X(512, 3, false, false, true)        \
  X(512, 2, false, false, true)        \
  X(1024, 7, false, false, false)      \
  X(1024, 6, false, false, false)      \
  X(1024, 5, false, false, false)      \
  X(1024, 3, false, false, false)      \
  X(1024, 7, false, false, true)       \
  X(1024, 6, false, false, true)       \
  X(1024, 5, false, false, true)       \
  X(1024, 3, false, false, true)
// ... and with MERGE (shard rule, non-negative weights): (window steps, 8-bit accumulators)
#define APSS_EVEN_MERGE_VARIANTS(X) \
  X(7, false) X(6, false) X(5, false) X(4, false) X(3, false) X(2, false) \
  X(7, true) X(6, true) X(5, true) X(4, true) X(3, true) X(2, true)

bool cx_variant_exists(const CxVariant &v) {
  if (v.even && v.merge) {
#define X(U, A8) \
    if (v.block == 512 && v.u == U && v.shard && !v.sgn && v.acc8 == A8) return v.chunk == 16 && !v.vrows && !v.longpf;
    APSS_EVEN_MERGE_VARIANTS(X)
#undef X
    return false;
  }
  if (v.even) {
#define X(B, U, SH, SG, A8) \
    if (v.block == B && v.u == U && v.shard == SH && v.sgn == SG && v.acc8 == A8) return v.chunk == 16 && !v.vrows && !v.longpf;
    APSS_EVEN_VARIANTS(X)
#undef X
    return false;
  }
#define X(B, U, SH, CH, VR, SG, LP, A8)                                                                                 \
  if (v.block == B && v.u == U && v.shard == SH && v.chunk == CH && v.vrows == VR && v.sgn == SG && v.longpf == LP &&   \
      v.acc8 == A8)                                                                                                     \
    return true;
  APSS_CX_VARIANTS(X)
#undef X
  return false;
}

int32_t launch_cx(apss_handle *h, const CxVariant &v, const ProbeArgs &a) {
  const dim3 grid((unsigned)((int64_t)a.n_tiles * a.n_chunks));
  if (v.even && v.merge) {
#define X(U, A8)                                                                                                           \
    if (v.block == 512 && v.u == U && v.acc8 == A8) {                                                                      \
      hipLaunchKernelGGL((k_probe_even_merged<U, A8>), grid, dim3(512), (size_t)h->dbgcfg.pad_lds, h->stream, a);           \
      HIPCHK(h, hipGetLastError());                                                                                        \
      return APSS_OK;                                                                                                      \
    }
    APSS_EVEN_MERGE_VARIANTS(X)
#undef X
    return fail(h, APSS_E_UNSUPPORTED, "no filter kernel for this combination of options");
  }
  if (v.even) {
#define X(B, U, SH, SG, A8)                                                                                                \
    if (v.block == B && v.u == U && v.shard == SH && v.sgn == SG && v.acc8 == A8) {                                        \
      hipLaunchKernelGGL((k_probe_even<B, U, (B <= 512 ? 128 : 256), SH, SG, A8>), grid, dim3(B),                          \
                         (size_t)h->dbgcfg.pad_lds, h->stream, a);                                                                                 \
      HIPCHK(h, hipGetLastError());                                                                                        \
      return APSS_OK;                                                                                                      \
    }
    APSS_EVEN_VARIANTS(X)
#undef X
    return fail(h, APSS_E_UNSUPPORTED, "no filter kernel for this combination of options");
  }
#define X(B, U, SH, CH, VR, SG, LP, A8)                                                                                    \
  if (v.block == B && v.u == U && v.shard == SH && v.chunk == CH && v.vrows == VR && v.sgn == SG && v.longpf == LP &&      \
      v.acc8 == A8) {                                                                                                      \
    hipLaunchKernelGGL((k_probe_coarse<B, U, (B <= 512 ? 128 : 256), (B <= 512 ? 512 : 1024), SH, CH, VR, SG, LP, A8>), grid, dim3(B), \
                       0, h->stream, a);                                                                                   \
    HIPCHK(h, hipGetLastError());                                                                                          \
    return APSS_OK;                                                                                                        \
  }
  APSS_CX_VARIANTS(X)
#undef X
  return fail(h, APSS_E_UNSUPPORTED, "no filter kernel for this combination of options");
}

// ---- dense-head filter of one query batch (apss_head.hpp): appends its candidates to the sparse filter's list ----
int32_t run_head(apss_handle *h, const ProbeArgs &a, int64_t nq, int64_t q_slot_base, const uint16_t *Wq, int64_t wq_rows,
                 float thr, double bound, int64_t n_cand) {
  // INT8 rendering: integer products, int32 sums -- the only slack is the fp32 arithmetic that made the rows (k_head_pack)
  const bool i8 = h->head_i8;
  const double s2 = (double)h->head_s * (double)h->head_s;
  const int32_t thr_i = i8 ? (int32_t)std::max(1.0, std::floor(s2 * (h->cfg.theta - 2e-5 * (1.0 + bound)))) : 0;
  const float inv_s2 = i8 ? (float)(1.0 / s2) : 0.f;
  const int kt = h->head_k;                        // width of a W row
  const int kh = std::min(kt, kHeadBlock);        // width of the first block
  const int n_blocks = kt > kHeadBlock ? 2 : 1;
  const int fold_w = kt - kHeadBlock;             // width of the folded block (128 | 256) when there is one
  if (nq <= 2 * kGemvQ) {
    HeadGemvArgs g{};
    g.Wq = Wq;
    g.Wc = h->W.p;
    g.n_rows = n_cand;
    g.q_slot_base = q_slot_base;
    g.nq = (int32_t)nq;
    g.kh = kh;
    g.kt = kt;
    g.part = h->head_part;
    g.n_parts = h->head_parts;
    g.q_ext = a.q_ext;
    g.c_ext = h->ext.p;
    g.thr = i8 ? (float)thr_i : thr;
    g.i8 = i8 ? 1 : 0;
    g.inv_s2 = inv_s2;
    g.res_q = a.res_q;
    g.res_c = a.res_c;
    g.res_s = a.res_s;
    g.res_cap = a.res_cap;
    g.counters = a.counters;
    g.head_pairs = h->head_ctr.p + 1;
    const int64_t blocks = std::min<int64_t>(2048, std::max<int64_t>(1, ceil_div(n_cand, 256 * (int64_t)h->head_parts)));
    for (int b = 0; b < n_blocks; ++b) {
      g.chunk0 = b * (kHeadBlock / 8);
      g.kh = b ? fold_w : kh;
      g.head_pairs = h->head_ctr.p + (b == 0 ? 1 : 3);  // (pairs sharing a term of the FIRST block: the statistic; the others' counts are dropped)
      hipLaunchKernelGGL(k_head_gemv, dim3((unsigned)blocks), dim3(256), 0, h->stream, g);
    }
    HIPCHK(h, hipGetLastError());
    h->st.head_flops = 2.0 * kt * (double)nq * (double)n_cand / (double)h->head_parts;
    return APSS_OK;
  }
  HeadGemmArgs g{};
  g.Wq = Wq;
  g.Wc = h->W.p;
  g.wq_rows = wq_rows;
  g.n_rows = n_cand;
  g.q_slot_base = q_slot_base;
  g.nq = (int32_t)nq;
  const int64_t qs0 = q_slot_base >= 0 ? q_slot_base : 0;
  g.qblock0 = qs0 / kHeadQBlock * kHeadQBlock;
  g.n_qblocks = (int32_t)(ceil_div(qs0 + nq, kHeadQBlock) - qs0 / kHeadQBlock);
  const int64_t ct = head_tile_rows(kh);
  g.n_ctiles = (int32_t)ceil_div(n_cand, ct);
  // candidate panels: enough workgroups to fill the chip several times over, a multiple of 8 (= XCDs) so that every
  // workgroup of an XCD streams panels of one residue class, and long enough to amortise the A-fragment load
  g.part = h->head_part;
  g.n_parts = h->head_parts;
  const int64_t my_ctiles = std::max<int64_t>(1, g.n_ctiles / g.n_parts);  // (this handle's share of the candidate tiles)
  int64_t panels = 8;
  while (panels * g.n_qblocks < 2048 && my_ctiles / (2 * panels) >= 32) panels *= 2;
  panels = std::max<int64_t>(1, std::min<int64_t>(panels, my_ctiles));
  g.n_panels = (int32_t)panels;
  g.q_ext = a.q_ext;
  g.c_ext = h->ext.p;
  g.thr = thr;
  g.thr_i = thr_i;
  g.inv_s2 = inv_s2;
  g.res_q = a.res_q;
  g.res_c = a.res_c;
  g.res_s = a.res_s;
  g.res_cap = a.res_cap;
  g.counters = a.counters;
  g.head_pairs = h->head_ctr.p + 1;
  g.kt = kt;
  // one launch multiplies a group of query blocks sized for about a second of matrix-core time (4e12 elements): a join of ten
  // million rows becomes a sequence of launches of bounded duration instead of one kernel that runs for a quarter of a minute
  const int64_t all_qblocks = g.n_qblocks, first_qblock0 = g.qblock0;
  const int64_t per_launch = std::max<int64_t>(1, (int64_t)(4e12 / ((double)kHeadQBlock * (double)ct * (double)my_ctiles)));
  for (int b = 0; b < n_blocks; ++b) {  // (block 1: the folded terms -- the same contraction, no pair statistic)
    g.chunk0 = b * (kHeadBlock / 8);
    for (int64_t q0 = 0; q0 < all_qblocks; q0 += per_launch) {
      g.n_qblocks = (int32_t)std::min<int64_t>(per_launch, all_qblocks - q0);
      g.qblock0 = first_qblock0 + q0 * kHeadQBlock;
      const dim3 grid((unsigned)((int64_t)g.n_qblocks * g.n_panels));
      // (three tile buffers + the barrier in mid-tile pay at KH = 256 only: 0.69 vs 0.67 of the bf16 peak; narrow blocks are
      // bound by their epilogue and lose with it: profiles/r03_head_gemm.md)
      // (the folded block has no positive count in its epilogue: there the three-buffer pipeline pays at 128 columns too,
      // 0.60 -> 0.68 of the peak at N = 1M)
      // INT8 rendering (one block of 128 | 256 byte-wide columns): v_mfma_i32_32x32x32_i8
      if (i8 && kh == 128) hipLaunchKernelGGL((k_head_gemm<128, true, 3, 8, true>), grid, dim3(512), 0, h->stream, g);
      else if (i8) hipLaunchKernelGGL((k_head_gemm<256, true, 3, 8, true>), grid, dim3(512), 0, h->stream, g);
      else if (b > 0 && fold_w == 128) hipLaunchKernelGGL((k_head_gemm<128, false, 3>), grid, dim3(512), 0, h->stream, g);
      else if (b > 0) hipLaunchKernelGGL((k_head_gemm<256, false, 3>), grid, dim3(512), 0, h->stream, g);
      else if (kh == 64) hipLaunchKernelGGL((k_head_gemm<64, true, 2>), grid, dim3(512), 0, h->stream, g);
      else if (kh == 128) hipLaunchKernelGGL((k_head_gemm<128, true, 2>), grid, dim3(512), 0, h->stream, g);
      else hipLaunchKernelGGL((k_head_gemm<256, true, 3>), grid, dim3(512), 0, h->stream, g);
    }
  }
  g.n_qblocks = (int32_t)all_qblocks;
  g.qblock0 = first_qblock0;
  HIPCHK(h, hipGetLastError());
  // multiplied elements: every (query block, candidate tile) the grid does not skip, at MFMA granularity
  double tiles = 0;
  for (int64_t b = 0; b < g.n_qblocks; ++b) {
    const int64_t hi = q_slot_base >= 0 ? std::min<int64_t>(g.n_ctiles, (g.qblock0 + (b + 1) * kHeadQBlock) / ct) : g.n_ctiles;
    tiles += hi > g.part ? (double)ceil_div(hi - g.part, g.n_parts) : 0.0;  // the tiles t < hi with t % n_parts == part
  }
  h->st.head_flops = 2.0 * kt * tiles * (double)ct * (double)kHeadQBlock;
  return APSS_OK;
}

// ---- probe the whole index with a query batch resident on the device ----
// q_head: rows of the batch in the dense-head block (null when the handle has none): the store's W for a stored batch,
// the staged q_W otherwise.
// What a probe knows when it picks the filter kernel's instantiation (plan_filter)
struct FilterFacts {
  bool acc8;         // this call's norms and row lengths leave room for 8-bit sums over the handle's >= 65536-row tiles
  bool shard_rule;   // term shard, or the sparse half of a handle with a dense-head block
  bool sgn;          // weights of either sign: the filter sums the positive products
  bool hybrid;       // the handle has a dense-head block
  int64_t s_max_nnz, s_nnz_end;  // longest staged query row, elements staged
  int64_t nq, idx_nnz;           // query rows, postings in the inverted index
  int max_merge_log2;            // k_probe_even: up to 2^this query rows may share a round (0: none)
};

// ---- which instantiation of the filter kernel: the register window (U steps of 8 chunks per wave) is sized for the
// chunks a wave expects per round; a round's cost grows with U whether or not its slots hold postings, so thin rounds
// (term shards: 1/T of a query's terms; sparse regimes: short segments) take a short window
// Pure host arithmetic on what ingest and the index build measured; the choice is reported in apss_stats.probe_kernel.
int32_t plan_filter(apss_handle *h, const FilterFacts &f, CxVariant &cxv, ProbeArgs &a) {
  const DebugCfg &dbg = h->dbgcfg;
  cxv.acc8 = f.acc8;
  cxv.block = h->cx.cb > 65536 || (h->cx.cb > 32768 && !cxv.acc8) ? 1024 : 512;
  cxv.shard = f.shard_rule;
  cxv.chunk = dbg.chunk8 ? 8 : 16;
  cxv.vrows = f.s_max_nnz > 512;
  cxv.sgn = f.sgn;
  const double nw = cxv.block / kWave;
  const double seg = (double)h->cx.cb * ((double)f.idx_nnz / (double)h->n_rows) / (double)h->cfg.dim;  // postings per (tile, term)
  double q_terms = (double)f.s_nnz_end / (double)f.nq;
  // a shard holds a binomial share of each query's terms; a window that overflows costs a whole-tile clear, so the
  // shard-rule launches size it for the upper end (3 sigma) and for segments one chunk longer than their mean
  const double t_hi = f.shard_rule ? q_terms + 3.0 * std::sqrt(q_terms) : q_terms;
  const double wave_chunks = std::ceil(t_hi / nw - 1e-9) * std::max(1.0, seg / 16.0 + (f.shard_rule ? 1.0 : 0.5));
  int u = (int)std::ceil(wave_chunks * (f.shard_rule ? 1.0 : 1.05) / 8.0);
  if (cxv.block == 1024) u = (q_terms / nw) * std::max(1.0, seg / 16.0 + 0.5) <= 17.0 && !dbg.big_u5 ? 3 : 5;
  else u = std::max(2, std::min(5, u));
  if (cxv.vrows || cxv.sgn) u = cxv.block == 1024 ? u : 5;
  if (dbg.chunk8) u = 4;
  if (dbg.window && cxv.block == 512 && !cxv.vrows && !cxv.sgn) u = dbg.window;
  // the skewed tail of a handle with a dense-head block: full window, prefetched long sweeps.  (A term shard's tail range
  // may hold no long segment at all once the block has taken the frequent terms: it then runs like any other shard)
  const bool long_tail = f.hybrid && h->cx.max_seg > (uint32_t)kLongLenW && !cxv.acc8;
  if (long_tail && !dbg.window) {
    u = 5;
    cxv.longpf = true;
  }
  if (cxv.vrows && cxv.shard && !cxv.sgn) {  // long rows (real TF-IDF) on a term shard: the shard-rule instantiation that takes virtual rows
    u = 5;
    cxv.longpf = true;
  }
  if (dbg.longpf && cxv.block == 512 && !cxv.vrows && !cxv.sgn && !cxv.acc8) {
    u = 5;
    cxv.longpf = true;
  }
  cxv.u = u;
  // k_probe_even (apss_even.hpp): F waves stage the round, the others add an even share of its chunks -- a wave's window
  // then holds a 1/A share of the ROUND's chunks, not the chunks of the wave's own terms.
  // Staging lanes per term: as many as keep the staging waves at <= a quarter of the workgroup.
  // Where it is taken (measured on C3 and its shards, ms of the filter kernel, k_probe_even vs k_probe_coarse): term shards
  // T = 8: 23.4 vs 40.5, T = 4: 35.6 vs 62.4, T = 2 (one staging lane per term, 7-step windows): 58.7 vs 77.9; the sparse
  // regime's 1024-thread kernel (C5's shape at a fifth of N): 218.2 vs 291.5; the plain handle of C3 itself (100-term rows,
  // LDS-throughput-bound): 100.6 vs 111.3 with six adding waves x 6 steps, 115.4 with 7 steps -- a plain 512-thread handle
  // takes it when its adding waves issue no more window steps per round than k_probe_coarse's eight would.
  const bool wide_ok = f.shard_rule || cxv.block == 1024 || dbg.even_wide;
  // (not where long segments abound -- more than 16 long terms per tile: the thin-round kernel sweeps them on its rare
  // path, meant for one round in a few.  C3 with Zipf(0.5) terms, ~400 long terms per tile and half a dozen in every round,
  // measured 1395 ms there against 518 ms on k_probe_coarse)
  const bool few_longs = (double)h->cx.long_segs <= 16.0 * (double)std::max<int64_t>(1, h->cx.n_tiles) || f.shard_rule || cxv.block == 1024;
  // MERGED rounds (apss_even.hpp): M = 2^m neighbouring query rows staged as one row, when a single query's round is thin (a
  // window of <= 4 steps) and the merged round still fits a window; the caller says how many the accumulators have room for
  a.merge_log2 = 0;
  int ue_single = 0;
  for (int m = 0; m <= f.max_merge_log2; ++m) {
    const int64_t M = 1LL << m;
    const int64_t row_max = std::min<int64_t>(f.s_max_nnz * M, f.s_nnz_end);  // longest staged row: at most M of the longest
    int flat_group_log2 = 2;
    while (flat_group_log2 > 0 && ceil_div(row_max, kWave >> flat_group_log2) > (int64_t)nw / 4) --flat_group_log2;
    if (dbg.flat_group >= 0 && dbg.flat_group < flat_group_log2) flat_group_log2 = dbg.flat_group;
    const int64_t flat_waves = std::max<int64_t>(1, ceil_div(row_max, kWave >> flat_group_log2));
    if (!(!cxv.vrows && !cxv.longpf && cxv.chunk == 16 && !dbg.no_even && !dbg.window && flat_waves <= (int64_t)nw / 4 && few_longs)) break;
    CxVariant ev = cxv;
    ev.even = true;
    ev.merge = m > 0;
    const double add_waves = nw - (double)flat_waves;               // a wave that stages a round adds nothing in it
    // postings per (tile, term) over the terms this handle indexes (a term shard: its range, not the whole dimension)
    const double seg_here = seg * (double)h->cfg.dim / (double)std::max<int64_t>(1, (int64_t)h->cfg.term_hi - h->cfg.term_lo);
    const double cpt = std::max(1.0, seg_here / 16.0 + 0.5);      // chunks per term
    const double mq_terms = q_terms * (double)M;                    // terms of a round
    // (mean + 2 sigma of a binomial share of the terms: the rounds beyond read their last chunks from the strip and end
    // with a whole-tile clear, ~2 % of them; at 3 sigma the T = 8 shard took a 3-step window: 24.8 vs 23.2 ms)
    double round_chunks = mq_terms * cpt + 2.0 * std::sqrt(mq_terms * cpt * cpt + mq_terms * 0.3);
    // ... that is the uniform-terms estimate; the index build MEASURED what an average stored row deals out per tile
    // (sum over terms of P(term in the row) x chunks of its segment): under a skewed distribution the terms a query holds
    // are the ones with the long segments, and the estimate above falls short by a factor (power-law C5's tail: 85
    // estimated, 200 dealt out: most rounds overflowed their window, a whole-tile clear each)
    if (h->cx.round_chunks > 0.0) {
      const double rc = h->cx.round_chunks * (double)M;
      const double per_term = std::max(1.0, rc / std::max(1.0, mq_terms));
      round_chunks = std::max(round_chunks, rc + 2.0 * std::sqrt(rc * per_term));
    }
    // (a plain handle's rows are whole rows: no binomial share of the terms; the longest row bounds the round)
    if (!f.shard_rule) round_chunks = std::min(round_chunks, 1.05 * (double)f.s_max_nnz * cpt);
    int ue = (int)std::ceil(round_chunks / (8.0 * add_waves));
    // (a window that overflows most rounds pays a whole-tile clear each time)
    const bool fits = ue <= 7 && !cxv.sgn && (wide_ok || add_waves * std::max(2, ue) <= nw * (double)cxv.u);
    ue = cxv.block == 1024 ? (ue <= 3 ? 3 : std::max(5, ue)) : std::max(2, ue);
    ev.u = ue;
    if (dbg.diag)
      fprintf(stderr, "[apss diag] even? merge %lld block %d u %d | F %lld G %d A %.0f chunks %.1f ue %d fits %d exists %d q_max %lld q_terms %.1f seg %.1f\n",
              (long long)M, cxv.block, cxv.u, (long long)flat_waves, 1 << flat_group_log2, add_waves, round_chunks, ue, (int)fits,
              (int)cx_variant_exists(ev), (long long)f.s_max_nnz, q_terms, seg);
    if (m == 0) ue_single = ue;
    if (!(fits && cx_variant_exists(ev))) break;
    // (measured on C3's shards, filter kernel: T = 8, window 2 -> 4 steps: 13.1 -> 11.1 ms; T = 4, 4 -> 7 steps: 19.9 -> 17.3 ms; T = 2
    // needs its 7 steps for one row.  FOUR rows per round at T = 8, 7 steps and 2^5 units per 1.0: 10.6 ms but 6.5 x the
    // candidates for the exchange -- not by default, APSS_DEBUG=merge=2)
    if (m > 0 && !(ue_single <= (dbg.merge_single > 0 ? dbg.merge_single : 4) && ue <= (dbg.merge_u > 0 ? dbg.merge_u : 7))) break;
    cxv = ev;
    a.flat_waves = (int32_t)flat_waves;
    a.flat_group_log2 = flat_group_log2;
    a.merge_log2 = m;
  }
  if (!cx_variant_exists(cxv)) return fail(h, APSS_E_UNSUPPORTED, "no filter kernel for this combination of options");
  return APSS_OK;
}

// Queries of more terms than a round of the filter takes (512) are cut into parts that share the accumulators: part v covers
// elements vrow_ptr[v] .. vrow_ptr[v + 1]) of query vrow_q[v]; vq_first[q] is query q's first part.
int32_t cut_long_queries(apss_handle *h, const int64_t *s_rowptr, int64_t nq, int vrow_part, ProbeArgs &a) {
  // queries of more terms than a round takes: cut them into parts that share the accumulators
  APSS_TRY(ensure(h, h->vrow_np, (size_t)nq + 1));
  APSS_TRY(ensure(h, h->vrow_first, (size_t)nq + 2));
  APSS_TRY(ensure(h, h->vq_first, (size_t)nq + 1));
  hipLaunchKernelGGL(k_vrow_count, dim3((unsigned)ceil_div(nq, 256)), dim3(256), 0, h->stream, s_rowptr, nq, vrow_part, h->vrow_np.p);
  APSS_TRY(scan_i64(h, (const int64_t *)h->vrow_np.p, h->vrow_first.p, nq));
  HIPCHK(h, hipGetLastError());
  int64_t nv = 0;
  HIPCHK(h, hipMemcpyAsync(&nv, h->vrow_first.p + nq, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  APSS_TRY(ensure(h, h->vrow_ptr, (size_t)nv + 1));
  APSS_TRY(ensure(h, h->vrow_q, (size_t)nv + 1));
  hipLaunchKernelGGL(k_vrow_fill, dim3((unsigned)ceil_div(nq + 1, 256)), dim3(256), 0, h->stream, s_rowptr, nq, vrow_part,
                     (const int64_t *)h->vrow_first.p, h->vq_first.p, h->vrow_ptr.p, h->vrow_q.p);
  HIPCHK(h, hipGetLastError());
  a.vq_first = h->vq_first.p;
  a.vrow_ptr = h->vrow_ptr.p;
  a.vrow_q = h->vrow_q.p;
  return APSS_OK;
}

// The exact pass of the two-pass join: the filters' survivors (h->res_*, h->n_res of them; with a dense-head block a pair may
// be there twice) are re-scored from the fp32 store and pruned at theta (k_rescore); then the rows that still wait outside the
// index are scored against every query, pair by pair (k_tail_score).  Leaves the final list in h->fin_* / h->n_res.
int32_t exact_pass(apss_handle *h, bool hybrid, double theta, int64_t nq, const int64_t *q_rowptr, const int32_t *q_idx,
                   const float *q_val, const int64_t *q_ext, int64_t tail_n) {
  float ms = 0.f;
  // exact pass: re-score what the filter(s) let through from the fp32 store and prune at theta
  int64_t n_cand = h->n_res;
  h->st.filter_survivors = n_cand;
  h->n_res = 0;
  const int32_t *cand_q = h->res_q.p, *cand_c = h->res_c.p;
  if (hybrid && n_cand > 0) {
    // a pair that passed both filters is in the list twice
    uint64_t tab = 1024;
    while (tab < (uint64_t)n_cand * 2) tab <<= 1;
    APSS_TRY(ensure(h, h->dedup_tab, (size_t)tab, 0, true));
    APSS_TRY(ensure(h, h->uq_q, (size_t)n_cand, 0, true));
    APSS_TRY(ensure(h, h->uq_c, (size_t)n_cand, 0, true));
    APSS_TRY(ensure(h, h->uq_s, (size_t)n_cand, 0, true));
    HIPCHK(h, hipMemsetAsync(h->dedup_tab.p, 0xff, (size_t)tab * sizeof(unsigned long long), h->stream));
    HIPCHK(h, hipMemsetAsync(h->counters.p, 0, sizeof(unsigned long long), h->stream));
    hipLaunchKernelGGL(k_pair_dedup, dim3((unsigned)ceil_div(n_cand, 256)), dim3(256), 0, h->stream,
                       (const int32_t *)h->res_q.p, (const int32_t *)h->res_c.p, (const float *)h->res_s.p, n_cand,
                       h->dedup_tab.p, tab - 1, h->uq_q.p, h->uq_c.p, h->uq_s.p, h->counters.p);
    HIPCHK(h, hipGetLastError());
    unsigned long long nu = 0;
    HIPCHK(h, hipMemcpyAsync(&nu, h->counters.p, sizeof(nu), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    n_cand = (int64_t)nu;
    cand_q = h->uq_q.p;
    cand_c = h->uq_c.p;
  }
  if (n_cand > 0) {
    APSS_TRY(ensure(h, h->fin_q, (size_t)n_cand, 0, true));
    APSS_TRY(ensure(h, h->fin_c, (size_t)n_cand, 0, true));
    APSS_TRY(ensure(h, h->fin_s, (size_t)n_cand, 0, true));
    HIPCHK(h, hipMemsetAsync(h->counters.p, 0, sizeof(unsigned long long), h->stream));
    RescoreArgs r{};
    r.n_pairs = n_cand;
    r.q_row = cand_q;
    r.c_slot = cand_c;
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
    // (a launch may not have 2^32 threads or more: 16 lanes per pair -> at most 2^27 pairs per launch)
    for (int64_t p0 = 0; p0 < n_cand; p0 += (1LL << 27)) {
      RescoreArgs rr = r;
      rr.n_pairs = std::min<int64_t>(1LL << 27, n_cand - p0);
      rr.q_row = cand_q + p0;
      rr.c_slot = cand_c + p0;
      hipLaunchKernelGGL(k_rescore, dim3((unsigned)std::max<int64_t>(1, ceil_div(rr.n_pairs, (int64_t)kRescorePairs * kRescoreTrips))), dim3(kRescoreBlock), 0, h->stream, rr);
      HIPCHK(h, hipGetLastError());
    }
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    unsigned long long nfin = 0;
    HIPCHK(h, hipMemcpyAsync(&nfin, h->counters.p, sizeof(nfin), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->st.rescore_ms = ms;
    h->n_res = (int64_t)nfin;
  }
  if (tail_n > 0) {
    // the rows that wait outside the index: every (query, tail row) pair, exactly (k_tail_score)
    const int64_t pairs = nq * tail_n;
    APSS_TRY(ensure(h, h->fin_q, (size_t)(h->n_res + pairs), (size_t)h->n_res));
    APSS_TRY(ensure(h, h->fin_c, (size_t)(h->n_res + pairs), (size_t)h->n_res));
    APSS_TRY(ensure(h, h->fin_s, (size_t)(h->n_res + pairs), (size_t)h->n_res));
    HIPCHK(h, hipMemsetAsync(h->counters.p, 0, kCtrCount * sizeof(unsigned long long), h->stream));
    TailArgs t{};
    t.nq = nq;
    t.n_tail = tail_n;
    t.q_rowptr = q_rowptr;
    t.q_idx = q_idx;
    t.q_val = q_val;
    t.q_ext = q_ext;
    t.c_rowptr = h->rowptr.p;
    t.c_idx = h->idx.p;
    t.c_val = h->val.p;
    t.c_ext = h->ext.p;
    t.tail0 = h->idx_rows;
    t.theta = (float)theta;
    t.out_q = h->fin_q.p;
    t.out_c = h->fin_c.p;
    t.out_s = h->fin_s.p;
    t.out_base = h->n_res;
    t.counters = h->counters.p;
    hipLaunchKernelGGL(k_tail_score, dim3((unsigned)ceil_div(pairs * kGroup, 256)), dim3(256), 0, h->stream, t);
    HIPCHK(h, hipGetLastError());
    unsigned long long tc[kCtrCount];
    HIPCHK(h, hipMemcpyAsync(tc, h->counters.p, sizeof(tc), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->n_res += (int64_t)tc[kCtrResults];
    h->st.posting_visits += (int64_t)tc[kCtrVisits];
    h->st.device_posting_visits += (int64_t)tc[kCtrVisits];
    h->st.candidate_pairs += (int64_t)tc[kCtrCands];
  }
  h->out_q = h->fin_q.p;
  h->out_c = h->fin_c.p;
  h->out_s = h->fin_s.p;
  h->st.result_pairs = h->n_res;
  return APSS_OK;
}

int32_t pack_query_head(apss_handle *h, const int64_t *rowptr, const int32_t *idx, const float *val, int64_t nq);

// q_slot_first: slot of query row 0 when the batch is rows of the store (self-join, insert-and-query), else -1.
int32_t probe(apss_handle *h, int64_t nq, const int64_t *q_rowptr, const int32_t *q_idx, const float *q_val,
              const int64_t *q_ext, int64_t q_slot_first, int64_t q_max_nnz, float q_max_norm2,
              int64_t q_nnz_end, int64_t *n_results) {
  const DebugCfg &dbg = h->dbgcfg;
  h->res_q_ext = q_ext;
  h->last_q_rowptr = q_rowptr;
  h->last_q_idx = q_idx;
  h->last_q_val = q_val;
  h->last_nq = nq;
  h->n_res = 0;
  h->st.posting_visits = h->st.candidate_pairs = h->st.result_pairs = 0;
  h->st.device_posting_visits = 0;
  h->st.symmetric = 0;
  h->st.symmetric_declined = APSS_SYM_NOT_WHOLE;
  h->st.query_chunk = 0;
  h->st.filter_tile_rows = 0;
  h->st.probe_ms = 0;
  h->st.probe_launches = 0;
  h->st.thin_launches = 0;
  h->st.queries_per_round = 1;
  h->st.probe_kernel[0] = 0;
  h->st.head_pairs = h->st.head_survivors = 0;
  h->st.head_ms = h->st.head_flops = 0;
  h->st.head_terms = h->head_k ? (int64_t)h->head_terms.size() : 0;
  h->st.head_columns = (uint32_t)h->head_k;
  h->st.head_int8 = h->head_k && h->head_i8 ? 1 : 0;
  if (n_results) *n_results = 0;
  APSS_TRY(ensure(h, h->counters, kCtrCount));
  h->st.filter_survivors = 0;
  h->st.rescore_ms = 0;
  if (nq == 0 || h->n_rows == 0 || q_nnz_end <= 0 || h->nnz == 0) return APSS_OK;  // nothing can share a term
  if (nq > 0x7fffffffLL) return fail(h, APSS_E_INVALID, "query batch too large");

  // What the SPARSE filter stages: the query rows themselves -- or, on a handle with a dense-head block, their tail view
  // (the rows without the block's entries; a stored batch's was made when it was indexed, an outside batch's is made here).
  // Exact rescoring, the waiting rows' direct scoring and apss_partial_scores_dev keep reading the whole rows.
  const int64_t *s_rowptr = q_rowptr;
  const int32_t *s_idx = q_idx;
  const float *s_val = q_val;
  int64_t s_max_nnz = q_max_nnz, s_nnz_end = q_nnz_end;
  if (h->head_k > 0) {
    if (q_slot_first >= 0 && q_slot_first < h->idx_rows) {
      s_rowptr = h->tv.rowptr.p + q_slot_first;
      s_idx = h->tv.idx.p;
      s_val = h->tv.val.p;
      s_max_nnz = h->tv.max_nnz;
      s_nnz_end = h->tv.nnz;
    } else {
      APSS_TRY(build_tail_view(h, h->qtv, q_rowptr, q_idx, q_val, 0, nq, 0, false));
      s_rowptr = h->qtv.rowptr.p;
      s_idx = h->qtv.idx.p;
      s_val = h->qtv.val.p;
      s_max_nnz = h->qtv.max_nnz;
      s_nnz_end = h->qtv.nnz;
    }
  }
  const int64_t c_max_nnz = h->head_k > 0 ? h->tv.max_nnz : h->store_max_nnz;  // longest indexed row
  const int64_t idx_nnz = h->head_k > 0 ? std::max<int64_t>(h->tv.nnz, 1) : h->nnz;  // postings in the inverted index

  const double theta = h->cfg.theta;
  int mode;
  if (!(theta > 0.0)) mode = 2;
  else if (!h->nonneg || (q_slot_first < 0 && !h->q_nonneg) || (h->cfg.flags & APSS_FLAG_FORCE_SCAN)) mode = 1;
  else mode = 0;

  // every partial score must fit the fixed-point accumulators: by Cauchy-Schwarz a partial sum of non-negative
  // products is at most |q| * |c| <= the product of the largest row norms
  const double bound = std::sqrt((double)q_max_norm2) * std::sqrt((double)h->store_max_norm2) * 1.0001 + 1e-6;
  const double fx_scale = bound < 3.9 ? 1073741824.0 : (bound < 15.6 ? 268435456.0 : 0.0);
  const bool forced_general = (h->cfg.flags & APSS_FLAG_FORCE_GENERAL) || dbg.force_general;

  // ---- path 1: two-pass join (coarse filter + exact rescoring) ----
  // accumulator units per 1.0: a 16-bit sum holds S * |q||c| (fp16 weights: + 2^-11) plus one unit per shared term
  // S = the largest power of two that fits (2^15 for unit-norm input); un-normalised input gets a smaller one as long as
  // the threshold in units stays well above the round-up bias (one unit per shared term), else the filter passes too much
  const double cx_shared = (double)std::min<int64_t>(s_max_nnz, c_max_nnz);
  const double cx_room = 65535.0 - cx_shared;
  double cx_scale = 0.0;
  for (int k = 15; k >= 4 && cx_scale == 0.0; --k)
    if (bound * 1.0005 * std::ldexp(1.0, k) < cx_room) cx_scale = std::ldexp(1.0, k);
  const double cx_theta = std::floor(theta * cx_scale * (1.0 - 1.0 / 2048 - 1e-6));
  const bool cx_selective = cx_scale >= 16384.0 || cx_theta >= 4.0 * cx_shared;
  const bool cx_fp16_ok = std::sqrt((double)h->store_max_norm2) < 60000.0;  // a weight never exceeds its row's norm
  // the shard rule (term-range shards, and the sparse half of a handle with a dense-head block) scales the threshold
  // down per query and tile: the kernel clamps it at 1, which only admits more
  const bool hybrid_wanted = h->head_k > 0;
  const float head_thr = (float)(theta - 0.0080 * bound - 1e-5);
  // weights of either sign with theta > 0 go through the filter too (it then sums positive products only: an upper bound of
  // the partial score, under the shard rule as much as without it -- the rule's ratios are norms); a handle with a
  // dense-head block (and long queries over 65536-row tiles) has no such instantiation and keeps the general kernel
  const bool shard_rule = h->sharded || hybrid_wanted;
  const bool cx_signed = mode == 1 && !(h->cfg.flags & APSS_FLAG_FORCE_SCAN) && !hybrid_wanted &&
                         (h->cx.cb <= 32768 || s_max_nnz <= 512) && !dbg.chunk8 && !dbg.window;
  // 65536-row tiles (term shards; the sparse regime of a plain handle): the 8-bit filter -- 65536 candidates in 64 KB, two
  // 512-thread workgroups per CU -- if this call's norms and row lengths leave room for its sums
  const bool big_shard_tiles = shard_rule && h->cx.cb > 32768;
  const double a8_scale = h->cx.cb >= 65536 && mode == 0 && s_max_nnz <= 512 && !dbg.no_acc8 && !h->no_acc8 && !dbg.chunk8
                              ? acc8_scale(bound, cx_shared, theta) : 0.0;
  if ((big_shard_tiles || h->cx.cb > 65536) && !(a8_scale > 0)) {
    // not this time (a long row, a large norm, signed weights): back to 16-bit accumulators over smaller tiles, for good
    h->cx.cb = shard_rule ? 32768 : 65536;
    h->cx.n_tiles = 0;
    h->no_acc8 = true;
    h->downgrades |= APSS_DOWNGRADE_ACC8;
    APSS_TRY(build_index(h, 0));
    return probe(h, nq, q_rowptr, q_idx, q_val, q_ext, q_slot_first, q_max_nnz, q_max_norm2, q_nnz_end, n_results);
  }
  if (a8_scale > 0) {
    cx_scale = a8_scale;
  }
  const double cx_theta_used = std::floor(theta * cx_scale * (1.0 - 1.0 / 2048 - 1e-6));
  const bool coarse_path = h->use_coarse && (mode == 0 || cx_signed) && (cx_selective || a8_scale > 0) && cx_fp16_ok && !forced_general && nq < (1LL << 30) &&
                           !(shard_rule && h->cx.cb > 32768 && !(a8_scale > 0)) &&
                           !dbg.exact_accum && cx_scale > 0 && cx_theta < 65000.0 && (shard_rule || cx_theta - 2 >= 1.0) &&
                           std::min(c_max_nnz * (int64_t)h->cx.cb, idx_nnz) + (int64_t)kSegAlignC * h->cfg.dim < (1LL << 27);
  // rows waiting in the tail are scored pair by pair after the join over the index; that needs the two-pass path's final
  // list and a bounded number of pairs -- otherwise they are folded into the index first
  const int64_t tail_n = h->n_rows - h->idx_rows;
  if (tail_n > 0 && (!(coarse_path && !h->sharded) || nq * tail_n > kTailMaxPairs)) {
    APSS_TRY(build_index(h, h->idx_rows));
    return probe(h, nq, q_rowptr, q_idx, q_val, q_ext, q_slot_first, q_max_nnz, q_max_norm2, q_nnz_end, n_results);
  }
  // a stored batch that still waits in the tail is not in the index: for the join over the index it is an outside batch
  const int64_t q_slot_base = q_slot_first >= 0 && q_slot_first < h->idx_rows ? q_slot_first : -1;
  if (hybrid_wanted && !(coarse_path && mode == 0 && (h->head_i8 || head_thr >= 0.5 * theta))) {  // (the INT8 rows carry no rounding bound)
    // this call cannot take the hybrid path (signed or very long queries, norms out of range ...): the dense block's
    // terms go back into the inverted index, for good, and the call runs as on a handle without a block
    if (h->sharded)  // (a shard cannot leave the partition its peers were given)
      return fail(h, APSS_E_UNSUPPORTED, "this call needs the plain index (signed weights, norms out of the filter's range, theta <= 0): "
                                         "not available on a term shard with a dense-head block");
    h->head_blocked = true;
    h->downgrades |= APSS_DOWNGRADE_HEAD;
    APSS_TRY(build_index(h, 0));
    return probe(h, nq, q_rowptr, q_idx, q_val, q_ext, q_slot_first, q_max_nnz, q_max_norm2, q_nnz_end, n_results);
  }
  const bool hybrid = hybrid_wanted;
  // the batch's rows of the dense-head block and its tail ratios: the store's were packed when it was indexed; an outside
  // batch (or one waiting in the tail) is packed here
  if (hybrid && q_slot_base < 0 && !h->sharded) APSS_TRY(pack_query_head(h, q_rowptr, q_idx, q_val, nq));  // (a shard's ingest packed them)
  const float *q_sub = !shard_rule ? nullptr : (q_slot_base >= 0 ? h->sub.p + q_slot_base : h->q_sub.p);

  ProbeArgs a{};
  a.seg_stride = (int64_t)h->cfg.dim;
  a.ext_id = h->ext.p;
  a.c_scale = shard_rule ? h->sub.p : nullptr;
  a.n_rows = h->idx_rows;
  a.q_rowptr = s_rowptr;
  a.q_idx = s_idx;
  a.q_val = s_val;
  a.q_ext = q_ext;
  a.q_scale = shard_rule ? q_sub : nullptr;
  a.nq = (int32_t)nq;
  a.nq_rows = (int32_t)nq;
  a.q_nnz_end = s_nnz_end;
  // ~2 workgroups per CU per tile in flight at once, tiles swept one after another (tile-major grid) so the
  // chip works on one tile's postings at a time and they stay in L2 / Infinity Cache
  // ... that is the whole-store join (a million queries: 1024 chunks of ~1000 rounds each).  A streamed batch has few queries:
  // 1024 chunks would give a workgroup ONE round (B = 1024) for the price of its set-up (64 KB of accumulators cleared, the
  // tile's descriptors fetched) -- measured on C3 streamed in batches of 1024: probe kernels 655 ms at 1024 chunks, 143 ms at
  // 128.  So: as few chunks as still fill the chip eight times over with the tiles there are, never fewer than nq / 256.
  int64_t want_chunks = 1024;
  if (h->idx_rows > 0) {
    const int64_t tiles_now = std::max<int64_t>(1, ceil_div(h->idx_rows, h->use_coarse ? h->cx.cb : h->ex.cb));
    // (eight workgroups per resident slot: with four, the last, partly filled wave of workgroups cost a fifth of the launch)
    want_chunks = std::min<int64_t>(1024, std::max<int64_t>(ceil_div(4096, tiles_now), ceil_div(nq, 256)));
  }
  if (dbg.chunks > 0) want_chunks = dbg.chunks;
  a.q_chunk = (int32_t)std::max<int64_t>(1, ceil_div(nq, want_chunks));
  a.n_chunks = (int32_t)ceil_div(nq, a.q_chunk);
  a.q_slot_base = q_slot_base;
  a.theta = (float)theta;
  a.counters = h->counters.p;
  a.cx_scale = (float)cx_scale;
  a.cx_theta = (float)cx_theta_used;

  apss_handle::IndexSet &ix = coarse_path ? h->cx : h->ex;
  if (!coarse_path) APSS_TRY(ensure_exact_index(h));
  a.tile_seg = ix.seg.p;
  a.post = h->ex.post.p;
  a.post_c = h->cx.post_c.p;
  a.tile_post_base = ix.base.p;
  a.tile_scale = shard_rule ? (coarse_path ? h->tile_min_c.p : h->tile_min.p) : nullptr;
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

  // MERGED rounds (a term shard's thin rounds, apss_even.hpp): M neighbouring query rows share a round, their M filter sums one
  // accumulator -- which then has to hold M sums: the scale for M rows is the scale of one row with M times the norm bound and
  // M times the shared terms.  How many rows the accumulators have room for (plan_filter decides how many a window has):
  auto merged_scale = [&](int m) -> double {
    const double Mx = (double)(1 << m);
    if (a8_scale > 0) return acc8_scale(bound * Mx, cx_shared * Mx, theta);
    for (int k = 15; k >= 4; --k) {
      const double S = std::ldexp(1.0, k);
      if (bound * Mx * 1.0005 * S < 65535.0 - cx_shared * Mx)
        return S >= 16384.0 || std::floor(theta * S * (1.0 - 1.0 / 2048 - 1e-6)) >= 4.0 * cx_shared * Mx ? S : 0.0;
    }
    return 0.0;
  };
  // ... and how many the filter stays SELECTIVE with: a chance pair collects about 1 / t of the threshold's scale per shared term
  // (t = terms of a row inside this shard), and a round of M rows hits a given candidate M t^2 / R times on average (R = terms
  // of the range).  Rows of a dozen terms over thousands of terms (C3's shards: t = 12.5, R = 12500): a chance candidate is
  // hit once, by one row.  Rows of three or four terms over a few hundred: three chance hits from different rows of a round
  // cross the threshold (measured: 8 x more candidates than unmerged) -- not there.  A fuse behind the estimate: a merged
  // launch that reports more than two rounds per query row turns merging off for the handle (merge_off).
  // (Term shards only, with or without a dense-head block -- power-law C5's 8 x 1 shard at N = 2M: tail filter 39.5 -> 24.0 ms.  The
  // sparse half of a PLAIN handle with a block was tried: its rounds take 4-5 window steps, two rows do not fit one window.)
  const double t_shard = (double)s_nnz_end / (double)std::max<int64_t>(nq, 1);
  const double r_shard = (double)std::max<int64_t>(1, (int64_t)h->cfg.term_hi - h->cfg.term_lo);
  int max_merge_log2 = 0;
  if (coarse_path && h->sharded && mode == 0 && !cx_signed && dbg.merge != 0 && !h->merge_off && t_shard >= 8.0)
    while (max_merge_log2 < (dbg.merge > 0 ? std::min(dbg.merge, 2) : 1) && (nq >> (max_merge_log2 + 1)) >= 1 &&
           (double)(2 << max_merge_log2) * t_shard * t_shard <= 0.25 * r_shard && merged_scale(max_merge_log2 + 1) > 0)
      ++max_merge_log2;
  CxVariant cxv{};
  if (coarse_path)
    APSS_TRY(plan_filter(h, FilterFacts{a8_scale > 0, shard_rule, cx_signed, hybrid_wanted, s_max_nnz, s_nnz_end, nq, idx_nnz, max_merge_log2}, cxv, a));
  if (a.merge_log2 > 0) {
    const double S = merged_scale(a.merge_log2);
    a.cx_scale = (float)S;
    a.cx_theta = (float)std::floor(theta * S * (1.0 - 1.0 / 2048 - 1e-6));
  }
  // ---- SYMMETRIC whole-store join: the batch IS the indexed store (query row v = stored row v: apss_self_join, or
  // insert-and-query into an empty handle), so the pair (q, c) and the pair (c, q) share their terms and their score.  The
  // filter runs a (query tile S, candidate tile T) workgroup only for T <= S and k_mirror_survivors adds the other direction
  // of every survivor with T < S; both directions are then re-scored exactly, as without the symmetry: same result list,
  // half the rounds.  Needs query chunks that do not straddle a tile (q_chunk | cb).  The reference finds both directions by
  // probing twice (IndexingWorkerActor.scala:123-134); posting_visits / candidate_pairs keep counting what IT visits (the
  // kernels count a tile pair below the diagonal twice: both statistics are symmetric in the two tiles), and
  // device_posting_visits says what the kernels visited.
  bool tri = false;
  uint32_t sym_declined = APSS_SYM_RAN;  // why not (apss_stats.symmetric_declined)
  if (dbg.no_sym || (h->cfg.flags & APSS_FLAG_NO_SYMMETRY)) sym_declined = APSS_SYM_FLAG;
  else if (!(q_slot_base == 0 && nq == h->idx_rows && nq == h->n_rows && tail_n == 0)) sym_declined = APSS_SYM_NOT_WHOLE;
  else if (!coarse_path) sym_declined = APSS_SYM_PATH;
  else if (cxv.vrows) sym_declined = APSS_SYM_LONG_ROWS;
  else if (ix.n_tiles <= 1) sym_declined = APSS_SYM_ONE_TILE;
  else {
    int64_t qc = 1;
    while (qc < a.q_chunk) qc <<= 1;
    if (qc <= ix.cb && ix.cb % qc == 0) {
      tri = true;
      a.tri = 1;
      a.q_chunk = (int32_t)qc;
      a.n_chunks = (int32_t)ceil_div(nq, qc);
    } else {
      sym_declined = APSS_SYM_CHUNK;
    }
  }
  h->st.symmetric_declined = sym_declined;
  h->st.query_chunk = a.q_chunk;
  h->st.filter_tile_rows = ix.cb;
  h->st.queries_per_round = 1 << a.merge_log2;
  if (a.merge_log2 > 0) {
    // from here on the launch counts ROUNDS: a chunk is a whole number of them (a symmetric join's power of two stays one)
    const int64_t M = 1LL << a.merge_log2;
    const int64_t qc = ceil_div(a.q_chunk, M) * M;
    a.q_chunk = (int32_t)(qc / M);
    a.n_chunks = (int32_t)ceil_div(nq, qc);
    a.nq = (int32_t)ceil_div(nq, M);
    APSS_TRY(ensure(h, h->q_prenorm, (size_t)s_nnz_end));
    hipLaunchKernelGGL(k_prenorm_rows, dim3((unsigned)ceil_div(nq * kGroup, 256)), dim3(256), 0, h->stream, s_rowptr, s_val, q_sub, nq, h->q_prenorm.p);
    HIPCHK(h, hipGetLastError());
    a.q_val = h->q_prenorm.p;
  }
  const int vrow_part = 512;
  // (the filter kernels keep their LDS in static arrays: no dynamic allocation)
  const size_t lds = coarse_path ? 0
                     : wave_path ? probe_wave_lds_bytes(h->ex.cb, wave_block, wave_u, wave_longcap, wave_survcap)
                                 : probe_lds_bytes(h->ex.cb, gen_block(mode, h->ex.cb), mode);
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

  if (coarse_path && s_max_nnz > vrow_part) APSS_TRY(cut_long_queries(h, s_rowptr, nq, vrow_part, a));
  // one launch sweeps a group of tiles sized for roughly 2e11 posting visits (a few hundred ms): a very large join
  // becomes a sequence of launches of bounded duration instead of one kernel that runs for tens of seconds
  const int64_t total_tiles = ix.n_tiles;
  int64_t tiles_per_launch = std::max<int64_t>(1, total_tiles);
  if (total_tiles > 0) {
    const double per_tile = (double)s_nnz_end * ((double)idx_nnz / (double)total_tiles / (double)h->cfg.dim) + 1.0;
    tiles_per_launch = (int64_t)std::min<double>((double)total_tiles, std::max(1.0, std::floor(2e11 / per_tile)));
    if (dbg.tiles_per_launch > 0) tiles_per_launch = dbg.tiles_per_launch;
    tiles_per_launch = std::min<int64_t>(tiles_per_launch, std::max<int64_t>(1, 2000000000LL / std::max(1, a.n_chunks)));
  }
  // room for two hits per query up front -- and, with a dense-head block, for what its filters passed on the policy's sample
  // (a narrow folded block passes a few pairs per million: 1e8 of them at N = 1e7): growing means running the whole probe again
  int64_t want_cap = std::min<int64_t>(2 * nq, 1LL << 28);
  if (hybrid_wanted && h->head_sample_frac > 0.0)
    want_cap = std::min<int64_t>(want_cap + (int64_t)(1.5 * h->head_sample_frac * (double)nq * (double)h->idx_rows), 1LL << 30);
  else if (hybrid_wanted && h->sharded)  // (a shard was given its head: no sample of its own; power-law C5 measured ~1 per query and shard)
    want_cap = std::min<int64_t>(want_cap + 6 * nq, 1LL << 30);
  if (h->res_q.cap < (size_t)want_cap) {
    const size_t cap0 = (size_t)(dbg.res_cap > 0 ? dbg.res_cap : std::max<int64_t>(1 << 20, want_cap));  // (res_cap: test hook, forces the regrowth path)
    APSS_TRY(ensure(h, h->res_q, cap0, 0, true));
    APSS_TRY(ensure(h, h->res_c, cap0, 0, true));
    APSS_TRY(ensure(h, h->res_s, cap0, 0, true));
  }
  if (hybrid) APSS_TRY(ensure(h, h->head_ctr, 4));
  // ---- streamed batches (single-vector messages up to a few tens of thousands of rows): the exact pass is CHAINED behind the filter on the stream --
  // k_rescore takes its pair count from the filter's counter on the device, k_tail_score appends to the same list -- and the
  // call synchronises ONCE, reading all counters together (a host round trip costs as much as these kernels do)
  constexpr int64_t kChainMaxQueries = 65536;
  const bool chain = coarse_path && !h->sharded && !hybrid && !tri && nq <= kChainMaxQueries && !dbg.no_chain;
  const int64_t chain_tail_pairs = chain ? nq * tail_n : 0;
  if (chain) APSS_TRY(ensure(h, h->chain_ctr, kCtrCount));
  for (int attempt = 0; attempt < 3; ++attempt) {
    if (chain) {  // (room for everything the candidate list can hold + every (query, waiting row) pair)
      // (sized for the longest tail there can be, once: the tail grows by a row per message)
      const size_t fin_cap = h->res_q.cap + (size_t)(nq * std::max<int64_t>(tail_n, kTailMaxRows));
      APSS_TRY(ensure(h, h->fin_q, fin_cap));
      APSS_TRY(ensure(h, h->fin_c, fin_cap));
      APSS_TRY(ensure(h, h->fin_s, fin_cap));
    }
    if (a.merge_log2 > 0) {  // (k_expand_merged writes the pairs out of place)
      APSS_TRY(ensure(h, h->res2_q, h->res_q.cap, 0, true));
      APSS_TRY(ensure(h, h->res2_c, h->res_q.cap, 0, true));
      APSS_TRY(ensure(h, h->res2_s, h->res_q.cap, 0, true));
    }
    a.dbg = nullptr;
    a.res_q = h->res_q.p;
    a.res_c = h->res_c.p;
    a.res_s = h->res_s.p;
    a.res_cap = h->res_q.cap;
    HIPCHK(h, hipMemsetAsync(h->counters.p, 0, kCtrCount * sizeof(unsigned long long), h->stream));
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    int64_t n_launches = 0, thin_launches = 0;
    {  // the instantiation this call launches, spelled as rocprofv3 prints it (apss_stats.probe_kernel)
      auto tf = [](bool b) { return b ? "true" : "false"; };
      char *nm = h->st.probe_kernel;
      const size_t cap = sizeof(h->st.probe_kernel);
      if (coarse_path && cxv.even)
        if (cxv.merge) snprintf(nm, cap, "k_probe_even_merged<%d, %s>", cxv.u, tf(cxv.acc8));
        else snprintf(nm, cap, "k_probe_even<%d, %d, %d, %s, %s, %s>", cxv.block, cxv.u, cxv.block <= 512 ? 128 : 256, tf(cxv.shard), tf(cxv.sgn), tf(cxv.acc8));
      else if (coarse_path)
        snprintf(nm, cap, "k_probe_coarse<%d, %d, %d, %d, %s, %d, %s, %s, %s, %s>", cxv.block, cxv.u, cxv.block <= 512 ? 128 : 256,
                 cxv.block <= 512 ? 512 : 1024, tf(cxv.shard), cxv.chunk, tf(cxv.vrows), tf(cxv.sgn), tf(cxv.longpf), tf(cxv.acc8));
      else if (wave_path)
        snprintf(nm, cap, "k_probe_wave<%d, %d, %d, %d, %s, %s>", wave_block, wave_u, wave_longcap, wave_survcap, tf(h->sharded), tf(dbg.diag));
      else
        snprintf(nm, cap, "k_probe<%d, %d, %s>", mode, gen_block(mode, h->ex.cb), tf(gen_fx));
    }
    for (int64_t t0 = 0; t0 < total_tiles; t0 += tiles_per_launch, ++n_launches) {
      a.tile0 = (int32_t)t0;
      a.n_tiles = (int32_t)std::min<int64_t>(tiles_per_launch, total_tiles - t0);
      if (coarse_path) {
        APSS_TRY(launch_cx(h, cxv, a));
        thin_launches += cxv.even ? 1 : 0;
      } else if (wave_path && dbg.diag) {
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
        else if (gen_block(2, h->ex.cb) == 512) APSS_TRY((launch_probe<2, true, 512>(h, a, lds)));
        else APSS_TRY((launch_probe<2, true>(h, a, lds)));
      } else {
        if (mode == 0) APSS_TRY((launch_probe<0, false>(h, a, lds)));
        else if (mode == 1) APSS_TRY((launch_probe<1, false>(h, a, lds)));
        else if (gen_block(2, h->ex.cb) == 512) APSS_TRY((launch_probe<2, false, 512>(h, a, lds)));
        else APSS_TRY((launch_probe<2, false>(h, a, lds)));
      }
    }  // tile groups
    if (a.merge_log2 > 0) {  // (round, candidate) -> the round's (query row, candidate) pairs
      hipLaunchKernelGGL(k_counter_move, dim3(1), dim3(1), 0, h->stream, h->counters.p, (int)kCtrResults, (int)kCtrPre);
      hipLaunchKernelGGL(k_expand_merged, dim3((unsigned)std::min<int64_t>(1024, ceil_div(nq / 4 + 4096, 256))), dim3(256), 0, h->stream, (const int32_t *)a.res_q, (const int32_t *)a.res_c, (const float *)a.res_s,
                         (const unsigned long long *)(h->counters.p + kCtrPre), (uint64_t)a.res_cap, a.merge_log2, nq, q_ext, (const int64_t *)h->ext.p,
                         h->res2_q.p, h->res2_c.p, h->res2_s.p, h->counters.p + kCtrResults, h->counters.p + kCtrOver);
      HIPCHK(h, hipGetLastError());
      std::swap(h->res_q, h->res2_q);
      std::swap(h->res_c, h->res2_c);
      std::swap(h->res_s, h->res2_s);
      a.res_q = h->res_q.p;
      a.res_c = h->res_c.p;
      a.res_s = h->res_s.p;
      a.res_cap = std::min(h->res_q.cap, h->res2_q.cap);
      if (h->sharded && !dbg.no_merge_prune) {
        // ... and only those of them that pass the shard rule on their EXACT partial score stay (k_shard_prune): the exchange
        // gets no more candidates than without merging
        hipLaunchKernelGGL(k_counter_move, dim3(1), dim3(1), 0, h->stream, h->counters.p, (int)kCtrResults, (int)kCtrPre);
        ShardPruneArgs sp{};
        sp.in_q = h->res_q.p;
        sp.in_c = h->res_c.p;
        sp.in_s = h->res_s.p;
        sp.n_in = h->counters.p + kCtrPre;
        sp.cap = a.res_cap;
        sp.q_rowptr = s_rowptr;
        sp.q_idx = s_idx;
        sp.q_val = s_val;
        sp.q_sub = q_sub;
        sp.c_rowptr = h->head_k ? h->tv.rowptr.p : h->rowptr.p;
        sp.c_idx = h->head_k ? h->tv.idx.p : h->idx.p;
        sp.c_val = h->head_k ? h->tv.val.p : h->val.p;
        sp.c_sub = h->sub.p;
        sp.theta = (float)theta;
        sp.out_q = h->res2_q.p;
        sp.out_c = h->res2_c.p;
        sp.out_s = h->res2_s.p;
        sp.counter = h->counters.p + kCtrResults;
        sp.over = h->counters.p + kCtrOver;
        // (a grid for what a shard usually reports -- a fraction of a pair per query row; the kernel strides over whatever the counter holds)
        const int64_t grid_pairs = std::min<int64_t>((int64_t)a.res_cap, nq / 4 + 4096);
        hipLaunchKernelGGL(k_shard_prune, dim3((unsigned)std::min<int64_t>(2048, ceil_div(grid_pairs, kPrunePairs * 4))), dim3(kPruneBlock), 0, h->stream, sp);
        HIPCHK(h, hipGetLastError());
        std::swap(h->res_q, h->res2_q);
        std::swap(h->res_c, h->res2_c);
        std::swap(h->res_s, h->res2_s);
        a.res_q = h->res_q.p;
        a.res_c = h->res_c.p;
        a.res_s = h->res_s.p;
      }
    }
    if (tri) {
      HIPCHK(h, hipMemcpyAsync(h->counters.p + kCtrSnap, h->counters.p + kCtrResults, sizeof(unsigned long long), hipMemcpyDeviceToDevice, h->stream));
      hipLaunchKernelGGL(k_mirror_survivors, dim3(1024), dim3(256), 0, h->stream, a.res_q, a.res_c, a.res_s,
                         (const unsigned long long *)(h->counters.p + kCtrSnap), (uint64_t)a.res_cap, (int32_t)a.cb, h->counters.p + kCtrResults);
      HIPCHK(h, hipGetLastError());
    }
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    unsigned long long cc_stack[kCtrCount] = {0, 0, 0, 0, 0, 0, 0, 0}, c_stack[kCtrCount];
    unsigned long long *cc = h->pin ? reinterpret_cast<unsigned long long *>(h->pin + kPinBytes + 64) : cc_stack;
    unsigned long long *c = h->pin ? reinterpret_cast<unsigned long long *>(h->pin + kPinBytes + 128) : c_stack;
    for (int k = 0; k < kCtrCount; ++k) cc[k] = 0;
    if (chain) {
      HIPCHK(h, hipMemsetAsync(h->chain_ctr.p, 0, kCtrCount * sizeof(unsigned long long), h->stream));
      RescoreArgs r{};
      r.n_pairs = (int64_t)a.res_cap;
      r.n_pairs_dev = h->counters.p + kCtrResults;
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
      r.out_count = h->chain_ctr.p + kCtrResults;
      // (grid for what such a batch usually passes; the kernel strides over whatever the counter holds)
      const int64_t grid_pairs = std::min<int64_t>((int64_t)a.res_cap, 64 * nq + 4096);
      hipLaunchKernelGGL(k_rescore, dim3((unsigned)std::max<int64_t>(1, ceil_div(grid_pairs, (int64_t)kRescorePairs * kRescoreTrips))), dim3(kRescoreBlock), 0, h->stream, r);
      if (tail_n > 0) {
        TailArgs t{};
        t.nq = nq;
        t.n_tail = tail_n;
        t.q_rowptr = q_rowptr;
        t.q_idx = q_idx;
        t.q_val = q_val;
        t.q_ext = q_ext;
        t.c_rowptr = h->rowptr.p;
        t.c_idx = h->idx.p;
        t.c_val = h->val.p;
        t.c_ext = h->ext.p;
        t.tail0 = h->idx_rows;
        t.theta = (float)theta;
        t.out_q = h->fin_q.p;
        t.out_c = h->fin_c.p;
        t.out_s = h->fin_s.p;
        t.out_base = 0;  // (appends through the same counter as k_rescore: one list)
        t.counters = h->chain_ctr.p;
        hipLaunchKernelGGL(k_tail_score, dim3((unsigned)ceil_div(chain_tail_pairs * kGroup, 256)), dim3(256), 0, h->stream, t);
      }
      HIPCHK(h, hipGetLastError());
      HIPCHK(h, hipMemcpyAsync(cc, h->chain_ctr.p, kCtrCount * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    }
    unsigned long long sparse_results = 0, head_c[4] = {0, 0, 0, 0};
    if (hybrid) {
      // the dense half: same candidate list, same counter
      HIPCHK(h, hipMemcpyAsync(&sparse_results, h->counters.p + kCtrResults, sizeof(sparse_results), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipMemsetAsync(h->head_ctr.p, 0, 4 * sizeof(unsigned long long), h->stream));
      HIPCHK(h, hipEventRecord(h->ev2, h->stream));
      const bool stored = q_slot_base >= 0;
      APSS_TRY(run_head(h, a, nq, q_slot_base, stored ? h->W.p : h->q_W.p, stored ? (int64_t)(h->W.cap / h->head_k) : ceil_div(nq, kHeadCTile) * kHeadCTile, head_thr, bound, h->idx_rows));
      HIPCHK(h, hipEventRecord(h->ev3, h->stream));
      HIPCHK(h, hipMemcpyAsync(head_c, h->head_ctr.p, sizeof(head_c), hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipMemcpyAsync(c, h->counters.p, kCtrCount * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->st.probe_ms += ms;
    h->st.probe_launches += n_launches;
    h->st.thin_launches += thin_launches;
    h->st.posting_visits = (int64_t)c[kCtrVisits];
    h->st.device_posting_visits = (int64_t)(tri ? c[kCtrDevVisits] : c[kCtrVisits]);
    h->st.symmetric = tri ? 1u : 0u;
    h->st.candidate_pairs = (int64_t)c[kCtrCands];
    // the speed paths count a stored query's touch of its own slot; it is not a (q, c != q) pair
    if ((wave_path || coarse_path) && q_slot_base >= 0)
      h->st.candidate_pairs -= h->head_k > 0 ? ((q_slot_base == 0 && nq == h->n_rows) ? h->tv.nonempty : h->tv.last_nonempty)
                                             : ((q_slot_base == 0 && nq == h->n_rows) ? h->store_nonempty : h->last_batch_nonempty);
    if (hybrid) {
      HIPCHK(h, hipEventElapsedTime(&ms, h->ev2, h->ev3));
      h->st.head_ms += ms;
      // positive elements of the contraction, minus a stored query's product with itself
      int64_t self = 0;
      if (q_slot_base >= 0) self = (q_slot_base == 0 && nq == h->n_rows) ? h->head_nonempty : h->last_batch_head_nonempty;
      h->st.head_pairs = (int64_t)head_c[1] - self;
      h->st.head_survivors = (int64_t)(c[kCtrResults] - sparse_results);
      // a pair sharing head AND tail terms is scored by both filters; the distinct count lies between max and sum
      h->st.candidate_pairs = std::max(h->st.candidate_pairs, h->st.head_pairs);
    }
    h->st.result_pairs = (int64_t)c[kCtrResults];
    // (kCtrPre: the rounds the filter reported, or -- with the shard-rule test behind it -- the pairs they expanded to)
    if (a.merge_log2 > 0 && (int64_t)c[kCtrPre] > (2 * nq + 100000) * (h->sharded && !dbg.no_merge_prune ? (1LL << a.merge_log2) : 1LL))
      h->merge_off = true;  // (this call's list is right, the filter just passed too much)
    if (cxv.acc8 && (int64_t)c[kCtrResults] > std::max<int64_t>(64 * nq, 4000000)) {
      // the 8-bit filter turned out unselective on this data (skewed terms: chance pairs share dozens of them, and every
      // shared term adds its unit of round-up): correct, but the survivors would swamp the exact pass.  Back to 16-bit
      // accumulators, for good, and run the call again.
      h->cx.cb = shard_rule ? 32768 : 65536;
      h->cx.n_tiles = 0;
      h->no_acc8 = true;
      h->downgrades |= APSS_DOWNGRADE_ACC8;
      APSS_TRY(build_index(h, 0));
      return probe(h, nq, q_rowptr, q_idx, q_val, q_ext, q_slot_first, q_max_nnz, q_max_norm2, q_nnz_end, n_results);
    }
    if (std::max(c[kCtrResults], c[kCtrOver]) > a.res_cap) {
      // the result list overflowed (or a list on the way to it did: kCtrOver): grow to what the run asked for and repeat the (idempotent) probe
      const size_t asked = (size_t)std::max(c[kCtrResults], c[kCtrOver]);
      size_t need = asked + asked / 8 + 1024;
      if (tri) {
        // a symmetric run whose FILTER overflowed mirrored only what the list held: the counter under-reports.  What the run
        // needs is known all the same: the filter's own count (kCtrSnap, taken before the mirror) twice over, plus what the
        // dense half appended -- asked for at once, instead of one more overflow per stage (each retry is a whole probe)
        const size_t snap = (size_t)c[kCtrSnap];
        const size_t head_part = hybrid && c[kCtrResults] > sparse_results ? (size_t)(c[kCtrResults] - sparse_results) : 0;
        need = std::max(need, 2 * snap + head_part + (2 * snap + head_part) / 8 + 1024);
      }
      APSS_TRY(ensure(h, h->res_q, need, 0, true));
      APSS_TRY(ensure(h, h->res_c, need, 0, true));
      APSS_TRY(ensure(h, h->res_s, need, 0, true));
      continue;
    }
    if (chain) {  // the exact pass has run already: its list is final
      h->st.filter_survivors = (int64_t)c[kCtrResults];
      h->n_res = (int64_t)cc[kCtrResults];
      h->st.posting_visits += (int64_t)cc[kCtrVisits];
      h->st.device_posting_visits += (int64_t)cc[kCtrVisits];
      h->st.candidate_pairs += (int64_t)cc[kCtrCands];
      h->st.result_pairs = h->n_res;
      h->out_q = h->fin_q.p;
      h->out_c = h->fin_c.p;
      h->out_s = h->fin_s.p;
      if (n_results) *n_results = h->n_res;
      return APSS_OK;
    }
    h->n_res = (int64_t)c[kCtrResults];
    h->out_q = h->res_q.p;
    h->out_c = h->res_c.p;
    h->out_s = h->res_s.p;
    if (coarse_path && h->sharded) h->st.filter_survivors = h->n_res;  // a shard's survivors are its candidates (phase 2 scores them)
    if (coarse_path && !h->sharded) APSS_TRY(exact_pass(h, hybrid, theta, nq, q_rowptr, q_idx, q_val, q_ext, tail_n));
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
  APSS_TRY(ensure(h, h->in_val, (size_t)std::max<int64_t>(nnz, 1)));
  const size_t off_ext = (size_t)(n + 1) * sizeof(int64_t), off_val = off_ext + (size_t)n * sizeof(int64_t);
  const size_t off_idx = off_val + (size_t)nnz * sizeof(double), bytes = off_idx + (size_t)nnz * sizeof(int32_t);
  if (n > 0 && h->pin && bytes <= kPinBytes) {
    // a small message: packed into pinned memory [rowptr | ext ids | values | indices] -> ONE truly asynchronous copy, no
    // synchronisation here (the staging is the handle's own: the caller's arrays are consumed by the memcpy below, and every
    // entry point synchronises before it returns)
    APSS_TRY(ensure(h, h->pack, kPinBytes));
    std::memcpy(h->pin, rowptr, off_ext);
    std::memcpy(h->pin + off_ext, ext_ids, (size_t)n * sizeof(int64_t));
    if (nnz) {
      std::memcpy(h->pin + off_val, values, (size_t)nnz * sizeof(double));
      std::memcpy(h->pin + off_idx, indices, (size_t)nnz * sizeof(int32_t));
    }
    HIPCHK(h, hipMemcpyAsync(h->pack.p, h->pin, bytes, hipMemcpyHostToDevice, h->stream));
    h->up_rowptr = reinterpret_cast<const int64_t *>(h->pack.p);
    h->up_ext = reinterpret_cast<const int64_t *>(h->pack.p + off_ext);
    h->up_idx = reinterpret_cast<const int32_t *>(h->pack.p + off_idx);
    if (nnz) {
      hipLaunchKernelGGL(k_narrow_f64, dim3((unsigned)std::min<int64_t>(2048, ceil_div(nnz, 256))), dim3(256), 0, h->stream,
                         reinterpret_cast<const double *>(h->pack.p + off_val), h->in_val.p, nnz);
      HIPCHK(h, hipGetLastError());
    }
    return APSS_OK;
  }
  APSS_TRY(ensure(h, h->in_rowptr, (size_t)n + 1));
  APSS_TRY(ensure(h, h->in_ext, (size_t)std::max<int64_t>(n, 1)));
  APSS_TRY(ensure(h, h->in_idx, (size_t)std::max<int64_t>(nnz, 1)));
  APSS_TRY(ensure(h, h->in_val64, (size_t)std::max<int64_t>(nnz, 1)));
  h->up_rowptr = h->in_rowptr.p;
  h->up_ext = h->in_ext.p;
  h->up_idx = h->in_idx.p;
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
  // the results and the query batch of the last call may point into store arrays that this insert reallocates:
  // they are gone (apss_fetch_results / apss_partial_scores_dev answer APSS_E_STATE until the next query-type call)
  h->n_res = -1;
  h->res_q_ext = nullptr;
  h->last_q_rowptr = nullptr;
  h->last_q_idx = nullptr;
  h->last_q_val = nullptr;
  h->last_nq = 0;
  if (h->n_rows + n > 0x7fffffffLL) return fail(h, APSS_E_INVALID, "more than 2^31 - 1 vectors in one handle");
  int64_t kept_rows = 0, kept_nnz = 0;
  const bool was_nonneg = h->nonneg;
  APSS_TRY(ingest(h, n, nnz, d_rowptr, d_idx, d_val, d_ext, true, &kept_rows, &kept_nnz));
  if (h->sharded && h->head_k && !h->nonneg) {  // refused before anything is committed: the index stays as it was
    h->nonneg = was_nonneg;
    return fail(h, APSS_E_UNSUPPORTED, "a term shard with a dense-head block needs non-negative weights (apss_set_head_terms)");
  }
  const int64_t row0 = h->n_rows;
  h->n_rows += kept_rows;
  h->nnz += kept_nnz;
  // a small batch onto an existing index waits in the TAIL (k_tail_score scores it directly) until the tail is worth a
  // rebuild of the last tile: a single-vector message then costs no pass over the dim-wide segment table
  const bool tail_ok = h->use_coarse && !h->sharded && h->idx_rows > 0 && !h->dbgcfg.no_tail;
  if (tail_ok && kept_rows <= kTailMaxBatch && h->n_rows - h->idx_rows <= kTailMaxRows) {
    h->st.build_ms = 0;
    return APSS_OK;
  }
  APSS_TRY(build_index(h, row0));
  return APSS_OK;
}

// rows of the dense-head block and tail ratios of a query batch given as CSR with absolute row offsets
int32_t pack_query_head(apss_handle *h, const int64_t *rowptr, const int32_t *idx, const float *val, int64_t nq) {
  const int64_t q_pad = ceil_div(nq, kHeadCTile) * kHeadCTile;
  APSS_TRY(ensure(h, h->q_W, (size_t)(q_pad * h->head_k)));
  APSS_TRY(ensure(h, h->q_sub, (size_t)nq));
  // (a query batch with larger norms than the store's shrinks the block's scale: the store's rows are re-quantised)
  APSS_TRY(head_fit_scale(h, std::max((double)h->store_max_norm2, (double)h->q_max_norm2), reinterpret_cast<unsigned char *>(h->W.p),
                          ceil_div(h->idx_rows, kHeadCTile) * kHeadCTile * (int64_t)h->head_k));
  HeadPackArgs a{};
  head_pack_rendering(h, a);
  a.rowptr = rowptr;
  a.idx = idx;
  a.val = val;
  a.row0 = 0;
  a.row1 = nq;
  a.head_pos = h->head_pos.p;
  a.kh = h->head_k;
  a.W = h->q_W.p;
  a.w_row0 = 0;
  a.w_pad = q_pad;
  a.ratio_t = h->q_sub.p;
  a.head_nonempty = nullptr;
  a.row_inv = nullptr;
  a.prune_above = -INFINITY;
  a.fold_from = h->head_exact;
  hipLaunchKernelGGL(k_head_pack, dim3((unsigned)ceil_div(q_pad, 8)), dim3(512), 0, h->stream, a);
  HIPCHK(h, hipGetLastError());
  return APSS_OK;
}

int32_t query_dev_impl(apss_handle *h, int64_t n, int64_t nnz, const int64_t *d_rowptr, const int32_t *d_idx,
                       const float *d_val, const int64_t *d_ext, int64_t *n_results) {
  int64_t kept_rows = 0, kept_nnz = 0;
  APSS_TRY(ingest(h, n, nnz, d_rowptr, d_idx, d_val, d_ext, false, &kept_rows, &kept_nnz));
  // the batch's rows of the dense-head block and its tail ratios (the store's were packed when it was indexed)
  if (h->head_k && !h->sharded && kept_rows > 0) APSS_TRY(pack_query_head(h, h->q_rowptr.p, h->q_idx.p, h->q_val.p, kept_rows));
  return probe(h, kept_rows, h->q_rowptr.p, h->q_idx.p, h->q_val.p, h->q_ext.p, -1, h->q_max_nnz, h->q_max_norm2, kept_nnz,
               n_results);
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
  h->dbgcfg = parse_debug_env();
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
                  (!h->sharded || !h->dbgcfg.shard_exact);
  h->cb = cfg->tile_rows ? cfg->tile_rows : (cfg->theta > 0.0 ? 16384 : 8192);  // (theta <= 0: gen_block, two workgroups per CU)
  h->ex.cb = h->cb;
  h->ex.align = kSegAlign;
  h->cx.cb = std::min(2 * h->cb, 32768);
  if (h->dbgcfg.cx_tile) h->cx.cb = h->dbgcfg.cx_tile;  // experiment hook (multiple of 64, <= 65536)
  h->no_acc8 = h->dbgcfg.no_acc8;
  if (h->dbgcfg.fold_w == 128 || h->dbgcfg.fold_w == 256) {  // experiment: two blocks
    h->head_exact = 256;
    h->head_fold_w = h->dbgcfg.fold_w;
  } else if (h->dbgcfg.mix >= 32 && h->dbgcfg.mix <= 224 && h->dbgcfg.mix % 32 == 0) {  // experiment: another split of the one block
    h->head_exact = h->dbgcfg.mix;
    h->head_fold_w = 256 - h->dbgcfg.mix;
  }
  h->cx.align = h->dbgcfg.seg_align == 16 ? 16 : kSegAlignC;
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
      (e = hipEventCreate(&h->ev0)) != hipSuccess || (e = hipEventCreate(&h->ev1)) != hipSuccess ||
      (e = hipEventCreate(&h->ev2)) != hipSuccess || (e = hipEventCreate(&h->ev3)) != hipSuccess) {
    g_create_error = std::string("HIP init failed: ") + hipGetErrorString(e);
    delete h;
    return APSS_E_DEVICE;
  }
  h->stream = h->own_stream;
  if (hipHostMalloc((void **)&h->pin, kPinBytes + kPinScalars, hipHostMallocDefault) != hipSuccess) h->pin = nullptr;  // (without it: the unpacked copies)
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
  for (apss_handle::IndexSet *s : {&h->ex, &h->cx}) { release(s->seg); release(s->post); release(s->post_c); release(s->base); release(s->total); release(s->maxlen); release(s->chunkw); release(s->scan_part); }
  release(h->tile_min); release(h->tile_min_c); release(h->fin_q); release(h->fin_c); release(h->fin_s);
  release(h->q_rowptr); release(h->q_ext); release(h->q_idx); release(h->q_val); release(h->q_sub);
  release(h->s_keep); release(h->s_cnt); release(h->s_rowdst); release(h->s_nnzdst); release(h->scan_tmp);
  release(h->in_rowptr); release(h->in_ext); release(h->in_idx); release(h->s_inv); release(h->s_sub); release(h->in_val); release(h->in_val64); release(h->vq_first); release(h->vrow_q); release(h->vrow_ptr); release(h->vrow_np); release(h->vrow_first);
  release(h->res_q); release(h->res_c); release(h->res_s); release(h->res2_q); release(h->res2_c); release(h->res2_s); release(h->q_prenorm); release(h->counters); release(h->flagword); release(h->dbg);
  release(h->head_pos); release(h->W);
  for (apss_handle::TailView *v : {&h->tv, &h->qtv}) { release(v->rowptr); release(v->idx); release(v->val); release(v->erow); }
  release(h->tv_cnt); release(h->tv_off); release(h->tv_sum); release(h->q_W); release(h->df); release(h->dedup_tab);
  release(h->head_ctr); release(h->uq_q); release(h->uq_c); release(h->uq_s); release(h->pack); release(h->chain_ctr); release(h->app_seg); release(h->app_post); release(h->bk_cnt); release(h->bk_base); release(h->bk_idx); release(h->bk_erow); release(h->bk_val);
  if (h->pin) (void)hipHostFree(h->pin);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->ev2) (void)hipEventDestroy(h->ev2);
  if (h->ev3) (void)hipEventDestroy(h->ev3);
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
  return insert_dev_impl(h, n, rowptr[n], h->up_rowptr, h->up_idx, h->in_val.p, h->up_ext, &first);
}

int32_t apss_query(apss_handle *h, int64_t n, const int64_t *rowptr, const int32_t *indices, const double *values,
                   const int64_t *ext_ids, int64_t *n_results) {
  APSS_TRY(enter(h));
  APSS_TRY(validate_host_csr(h, n, rowptr, indices, values, ext_ids));
  APSS_TRY(upload(h, n, rowptr, indices, values, ext_ids));
  return query_dev_impl(h, n, n ? rowptr[n] : 0, h->up_rowptr, h->up_idx, h->in_val.p, h->up_ext, n_results);
}

int32_t apss_insert_and_query(apss_handle *h, int64_t n, const int64_t *rowptr, const int32_t *indices,
                              const double *values, const int64_t *ext_ids, int64_t *n_results) {
  APSS_TRY(enter(h));
  APSS_TRY(validate_host_csr(h, n, rowptr, indices, values, ext_ids));
  APSS_TRY(upload(h, n, rowptr, indices, values, ext_ids));
  return apss_insert_and_query_dev(h, n, n ? rowptr[n] : 0, h->up_rowptr, h->up_idx, h->in_val.p, h->up_ext, n_results);
}

int32_t apss_self_join(apss_handle *h, int64_t *n_results) {
  APSS_TRY(enter(h));
  if (h->idx_rows < h->n_rows) APSS_TRY(build_index(h, h->idx_rows));  // a self-join runs over the index: fold the tail in
  return probe(h, h->n_rows, h->rowptr.p, h->idx.p, h->val.p, h->ext.p, 0, h->store_max_nnz, h->store_max_norm2, h->nnz, n_results);
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
  if (h->pin && (size_t)count * 20 <= kPinBytes) {  // a small answer: one packed copy into pinned memory
    APSS_TRY(ensure(h, h->pack, kPinBytes));
    int64_t *pq = reinterpret_cast<int64_t *>(h->pack.p), *pc = pq + count;
    float *ps = reinterpret_cast<float *>(pc + count);
    hipLaunchKernelGGL(k_gather_ids, dim3((unsigned)ceil_div(count, 256)), dim3(256), 0, h->stream, h->out_q + offset, h->out_c + offset,
                       h->res_q_ext, (const int64_t *)h->ext.p, count, pq, pc);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(ps, h->out_s + offset, (size_t)count * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->pin, h->pack.p, (size_t)count * 20, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::memcpy(out_q, h->pin, (size_t)count * sizeof(int64_t));
    std::memcpy(out_c, h->pin + (size_t)count * sizeof(int64_t), (size_t)count * sizeof(int64_t));
    std::memcpy(out_score, h->pin + (size_t)count * 2 * sizeof(int64_t), (size_t)count * sizeof(float));
    return APSS_OK;
  }
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
  h->st.tiles = h->use_coarse && h->ex_built_rows < h->idx_rows ? h->cx.n_tiles : ceil_div(h->idx_rows, h->ex.cb);
  h->st.hbm_bytes = (int64_t)h->bytes_reserved;
  h->st.downgrades = h->downgrades;
  h->st.head_terms = h->head_k ? (int64_t)h->head_terms.size() : 0;  // (as they are now: a policy handle is asked after its insert)
  h->st.head_columns = (uint32_t)h->head_k;
  // the caller says how large ITS apss_stats is: an older caller's shorter struct gets the fields it knows, nothing beyond
  const int32_t caller = out->struct_size;
  if (caller < (int32_t)(2 * sizeof(int32_t)) || caller > (1 << 16))
    return fail(h, APSS_E_INVALID, "apss_stats.struct_size must be set to sizeof(apss_stats) before the call");
  const int32_t n = std::min<int32_t>(caller, (int32_t)sizeof(apss_stats));
  h->st.struct_size = n;
  std::memcpy(out, &h->st, (size_t)n);
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
  return probe(h, nq, h->rowptr.p + first, h->idx.p, h->val.p, h->ext.p + first, first, h->store_max_nnz, h->store_max_norm2, h->nnz,
               n_results);
}

int32_t apss_clear(apss_handle *h) {
  APSS_TRY(enter(h));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->n_rows = 0;
  h->idx_rows = 0;
  h->nnz = 0;
  h->n_tiles = 0;
  for (apss_handle::IndexSet *s : {&h->ex, &h->cx}) { s->n_tiles = 0; s->post_used = 0; s->h_base.clear(); s->built_rows = 0; }
  h->ex_built_rows = 0;
  h->n_res = -1;
  h->res_q_ext = nullptr;
  h->last_q_rowptr = nullptr;
  h->last_q_idx = nullptr;
  h->last_q_val = nullptr;
  h->last_nq = 0;
  h->nonneg = true;
  h->q_nonneg = true;
  h->no_acc8 = h->dbgcfg.no_acc8;
  h->merge_off = false;
  h->downgrades = 0;
  h->store_max_nnz = 0;
  h->store_max_norm2 = 0.f;
  h->store_nonempty = 0;
  if (!h->head_fixed) {  // (terms set through apss_set_head_terms are configuration: they outlive the data)
    h->head_k = 0;
    h->head_terms.clear();
  }
  h->head_eval_rows = 0;
  h->head_blocked = false;
  h->head_nonempty = 0;
  return APSS_OK;
}

int32_t apss_set_head_terms(apss_handle *h, int32_t n_terms, const int32_t *terms, int32_t part, int32_t n_parts) {
  APSS_TRY(enter(h));
  if (h->n_rows != 0) return fail(h, APSS_E_STATE, "apss_set_head_terms: the handle holds vectors (set the block before the first insert or after apss_clear)");
  if (n_terms < 0 || n_terms > kHeadMaxTerms || (n_terms > 0 && !terms)) return fail(h, APSS_E_INVALID, "apss_set_head_terms: 0 .. 32768 terms");
  if (n_parts < 1 || part < 0 || part >= n_parts) return fail(h, APSS_E_INVALID, "apss_set_head_terms: part must be in [0, n_parts)");
  if (n_parts > 1 && !h->sharded)
    return fail(h, APSS_E_INVALID, "apss_set_head_terms: only a term shard multiplies a share of the block (n_parts > 1)");
  if (n_terms == 0) {
    h->head_fixed = false;
    h->head_k = 0;
    h->head_terms.clear();
    h->head_part = 0;
    h->head_parts = 1;
    return APSS_OK;
  }
  if (!h->use_coarse || !(h->cfg.theta > 0.0))
    return fail(h, APSS_E_UNSUPPORTED, "a dense-head block needs the two-pass join (theta > 0, no EXACT_ACCUM / FORCE_* flag)");
  if (h->sharded && (h->cfg.flags & APSS_FLAG_ADMISSION))
    return fail(h, APSS_E_UNSUPPORTED, "a term shard packs the block's rows from the batch row by row: not with APSS_FLAG_ADMISSION");
  if (!(h->dbgcfg.fold_w == 128 || h->dbgcfg.fold_w == 256) && !h->dbgcfg.mix) {  // (the experiments' geometry stays)
    h->head_fold_w = h->head_fold_user ? h->head_fold_user : 128;
    h->head_exact = 256 - h->head_fold_w;
  }
  std::vector<int32_t> pos((size_t)h->cfg.dim, -1);
  for (int32_t i = 0; i < n_terms; ++i) {
    if (terms[i] < 0 || terms[i] >= h->cfg.dim || pos[(size_t)terms[i]] >= 0)
      return fail(h, APSS_E_INVALID, "apss_set_head_terms: terms must be distinct and in [0, dim)");
    pos[(size_t)terms[i]] = head_column(i, h->head_fold_w, h->head_exact);
  }
  APSS_TRY(ensure(h, h->head_pos, (size_t)h->cfg.dim));
  HIPCHK(h, hipMemcpyAsync(h->head_pos.p, pos.data(), (size_t)h->cfg.dim * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->head_terms.assign(terms, terms + n_terms);
  h->head_k = head_width(n_terms, h->head_fold_w, h->head_exact, head_wants_i8(h));
  h->head_fixed = true;
  h->head_blocked = false;
  h->head_part = part;
  h->head_parts = n_parts;
  h->cx.n_tiles = 0;
  h->ex_built_rows = 0;
  return APSS_OK;
}

int32_t apss_set_head_fold(apss_handle *h, int32_t columns) {
  APSS_TRY(enter(h));
  if (columns != 0 && columns != 64 && columns != 128 && columns != 192) return fail(h, APSS_E_INVALID, "apss_set_head_fold: 0 (default: 128), 64, 128 or 192 columns");
  if (h->n_rows != 0) return fail(h, APSS_E_STATE, "apss_set_head_fold: on an empty handle (it takes effect at the next apss_set_head_terms)");
  h->head_fold_user = columns;
  return APSS_OK;
}

int32_t apss_get_head_terms(apss_handle *h, int32_t capacity, int32_t *out_terms, int32_t *n_terms) {
  if (!h || !n_terms) return APSS_E_INVALID;
  const int32_t n = h->head_k ? (int32_t)h->head_terms.size() : 0;
  *n_terms = n;
  if (out_terms)
    for (int32_t i = 0; i < std::min(n, capacity); ++i) out_terms[i] = h->head_terms[(size_t)i];
  return APSS_OK;
}

int32_t apss_ext_ids_dev(apss_handle *h, const int64_t **d_store_ext, const int64_t **d_query_ext) {
  if (!h) return APSS_E_INVALID;
  if (d_store_ext) *d_store_ext = h->n_rows > 0 ? h->ext.p : nullptr;
  if (d_query_ext) *d_query_ext = h->res_q_ext;
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

int32_t apss_results_copy_dev(apss_handle *h, int64_t offset, int64_t count, int32_t *d_q_row, int32_t *d_c_slot,
                              float *d_score) {
  APSS_TRY(enter(h));
  if (h->n_res < 0) return fail(h, APSS_E_STATE, "no query has run on this handle since the last insert");
  if (offset < 0 || count < 0 || offset + count > h->n_res) return fail(h, APSS_E_INVALID, "copy range out of bounds");
  if (count == 0) return APSS_OK;
  if (d_q_row) HIPCHK(h, hipMemcpyAsync(d_q_row, h->out_q + offset, (size_t)count * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
  if (d_c_slot) HIPCHK(h, hipMemcpyAsync(d_c_slot, h->out_c + offset, (size_t)count * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
  if (d_score) HIPCHK(h, hipMemcpyAsync(d_score, h->out_s + offset, (size_t)count * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  return APSS_OK;
}

int32_t apss_partial_scores_dev(apss_handle *h, int64_t n_pairs, const int32_t *d_q_row, const int32_t *d_c_slot,
                                float *d_out_partial) {
  APSS_TRY(enter(h));
  if (n_pairs < 0) return fail(h, APSS_E_INVALID, "negative pair count");
  if (n_pairs == 0) return APSS_OK;
  if (!h->last_q_rowptr) return fail(h, APSS_E_STATE, "no query batch on this handle since the last insert");
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
  for (int64_t p0 = 0; p0 < n_pairs; p0 += (1LL << 27)) {  // (a launch may not have 2^32 threads or more)
    PartialArgs aa = a;
    aa.n_pairs = std::min<int64_t>(1LL << 27, n_pairs - p0);
    aa.q_row = d_q_row + p0;
    aa.c_slot = d_c_slot + p0;
    aa.out = d_out_partial + p0;
    hipLaunchKernelGGL(k_partial_scores, dim3((unsigned)ceil_div(aa.n_pairs * kGroup, 256)), dim3(256), 0, h->stream, aa);
    HIPCHK(h, hipGetLastError());
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return APSS_OK;
}

}  // extern "C"
