// apss_group.hip -- the term-sharded index of one node behind ONE object of the C ABI (include/apss.h, apss_group_*).
//
// What the reference does with actors -- WriteWorkerActor buckets a vector by dim % maxShardNum and flushes a DataPacket per
// shard (WriteWorkerActor.scala:164-183), EntryProxyActor fans it out to the IndexingWorkerActors by
// dim % maxIndexEntryActorNum (EntryProxyActor.scala:37-49), every worker handles IndexData alone
// (IndexingWorkerActor.scala:122-137) -- with the workers resident on the GPUs of one node: member g = one shard handle
// (apss_hip.hip) on one device, one host thread per member for the member-local phase, and the exchange of the members'
// answers (all-gather of candidate lists, all-reduce(SUM) of per-candidate partial scores) over RCCL on the members' streams.
// This file only uses the public handle ABI; it holds no index state of its own.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <rccl/rccl.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

#include "../../include/apss.h"

namespace {

thread_local std::string g_group_create_error;

using Clock = std::chrono::steady_clock;
inline double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }
__host__ __device__ inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- RCCL, loaded on first use: libapss_hip.so itself does not depend on librccl (a single-GPU deployment, or a JVM without
// RCCL on its library path, loads and runs without it).  A process that already maps an image with soname librccl.so.1
// (PyTorch-ROCm bundles one) gets THAT image: there is never a second copy.  RTLD_LOCAL, always: promoting librccl and its
// dependencies (librocm_smi64 ...) to the global scope lets a later loader of the same libraries bind to their statics and
// destroy them twice at exit (seen with `import torch` after a group's first RCCL call: double free in rocm_smi's teardown).
struct Rccl {
  void *lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  std::string err;
};

Rccl *load_rccl() {
  static std::mutex mu;
  static Rccl r;
  std::lock_guard<std::mutex> lk(mu);
  if (r.lib) return &r;
  std::vector<std::string> names = {"librccl.so.1", "librccl.so"};
  void *lib = nullptr;
  for (const std::string &n : names)
    if ((lib = dlopen(n.c_str(), RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL))) break;
  if (!lib) {
    // beside the HIP runtime this process runs on (PyTorch's bundle, or /opt/rocm/lib), then the loader's own search path
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
      std::string dir(info.dli_fname);
      const size_t slash = dir.rfind('/');
      if (slash != std::string::npos) {
        dir.resize(slash + 1);
        names.insert(names.begin(), {dir + "librccl.so.1", dir + "librccl.so"});
      }
    }
    names.push_back("/opt/rocm/lib/librccl.so.1");
    for (const std::string &n : names)
      if ((lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL))) break;
  }
  if (!lib) {
    r.err = std::string("librccl.so.1 not found: ") + (dlerror() ? dlerror() : "");
    return &r;
  }
#define APSS_RCCL_SYM(field, name)                                                      \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(lib, name));                     \
  if (!r.field) {                                                                       \
    r.err = std::string("librccl: missing symbol ") + name;                             \
    return &r;                                                                          \
  }
  APSS_RCCL_SYM(CommInitAll, "ncclCommInitAll")
  APSS_RCCL_SYM(CommDestroy, "ncclCommDestroy")
  APSS_RCCL_SYM(GetErrorString, "ncclGetErrorString")
  APSS_RCCL_SYM(Broadcast, "ncclBroadcast")
  APSS_RCCL_SYM(AllReduce, "ncclAllReduce")
  APSS_RCCL_SYM(GroupStart, "ncclGroupStart")
  APSS_RCCL_SYM(GroupEnd, "ncclGroupEnd")
#undef APSS_RCCL_SYM
  r.lib = lib;
  return &r;
}

// ---- a reusable barrier for the member threads of one call (C++17: no std::barrier)
class Barrier {
 public:
  explicit Barrier(int n) : n_(n) {}
  void wait() {
    std::unique_lock<std::mutex> lk(mu_);
    const uint64_t gen = gen_;
    if (++count_ == n_) {
      count_ = 0;
      ++gen_;
      cv_.notify_all();
    } else {
      cv_.wait(lk, [&] { return gen_ != gen; });
    }
  }

 private:
  std::mutex mu_;
  std::condition_variable cv_;
  int n_, count_ = 0;
  uint64_t gen_ = 0;
};

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;
};

// ---- kernels of the exchange (everything else is the shard handles' work)
// candidate (query row of the batch, candidate slot) -> one sortable 8-B key
__global__ void k_pack_keys(const int32_t *q_row, const int32_t *c_slot, int64_t n, unsigned long long *keys) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    keys[i] = ((unsigned long long)(uint32_t)q_row[i] << 32) | (unsigned long long)(uint32_t)c_slot[i];
}

__global__ void k_unpack_keys(const unsigned long long *keys, int64_t n, int32_t *q_row, int32_t *c_slot) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    q_row[i] = (int32_t)(keys[i] >> 32);
    c_slot[i] = (int32_t)(keys[i] & 0xffffffffULL);
  }
}

// the copies exchange's reduction: sum += part (one launch per member after the first)
__global__ void k_accumulate(float *sum, const float *part, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) sum[i] += part[i];
}

// the `>= theta` prune (IndexingWorkerActor.scala:93) over the reduced scores; survivors compacted per wavefront: ballot,
// one atomic per wave, prefix popcount for the lane's place
__global__ void k_threshold_compact(const float *score, const int32_t *q_row, const int32_t *c_slot, int64_t n, float theta,
                                    int32_t *out_q, int32_t *out_c, float *out_s, unsigned long long *out_count) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n_pad = ceil_div(n, 64) * 64;  // whole waves take every trip of the loop together
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += stride) {
    const bool keep = i < n && score[i] >= theta;
    const unsigned long long mask = __ballot(keep);
    if (mask == 0) continue;
    const int lane = threadIdx.x & 63;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(out_count, (unsigned long long)__popcll(mask));
    base = __shfl(base, 0);
    if (keep) {
      const unsigned long long at = base + (unsigned long long)__popcll(mask & ((1ULL << lane) - 1ULL));
      out_q[at] = q_row[i];
      out_c[at] = c_slot[i];
      out_s[at] = score[i];
    }
  }
}

__global__ void k_gather_ext(const int32_t *q_row, const int32_t *c_slot, const int64_t *q_ext, const int64_t *c_ext, int64_t n,
                             int64_t *out_q, int64_t *out_c) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    out_q[i] = q_ext[q_row[i]];
    out_c[i] = c_ext[c_slot[i]];
  }
}

// document frequencies of a device-resident batch (the layout decision of a group fed through the device entry point)
__global__ void k_df_hist(const int32_t *idx, int64_t nnz, int32_t dim, unsigned int *df) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t t = idx[i];
    if (t >= 0 && t < dim) atomicAdd(&df[t], 1u);  // (a malformed index is the shard handle's to refuse)
  }
}

inline unsigned grid_for(int64_t n, int threads = 256, int64_t cap = 4096) {
  return (unsigned)std::max<int64_t>(1, std::min<int64_t>(cap, ceil_div(n, threads)));
}

}  // namespace

struct apss_group {
  struct Member {
    int dev = 0;
    apss_handle *h = nullptr;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    // exchange buffers, on this member's device
    DevBuf<unsigned long long> keys, sorted, uniq, count;
    DevBuf<char> tmp;
    DevBuf<int32_t> uq, uc;
    DevBuf<float> partial, sum, stage;
    int64_t n_cand = 0, n_union = 0;
    double member_ms = 0, partial_ms = 0;
    int32_t rc = APSS_OK;
    std::string err;
  };
  apss_config cfg{};
  uint32_t flags = 0;
  int T = 0;
  std::vector<Member> m;
  std::string err;
  bool created = false;         // the members' handles exist (the layout is decided)
  bool cuts_named = false;      // apss_group_set_term_cuts
  std::vector<int32_t> cuts;    // T + 1
  std::vector<int32_t> head;    // shared dense-head terms
  int exchange = APSS_EXCHANGE_NONE;
  bool distinct_devices = true;
  int64_t n_rows = 0;
  // results of the last query-type call (exchange modes): on member 0's device
  DevBuf<int32_t> res_q, res_c;
  DevBuf<float> res_s;
  DevBuf<int64_t> ext_q, ext_c;
  int64_t n_res = -1;
  bool results_in_handle = false;  // one member, no exchange: the handle's own result list
  apss_group_stats st{};
};

namespace {

int32_t gfail(apss_group *g, int32_t rc, const std::string &msg) {
  g->err = msg;
  return rc;
}

#define GHIP(g, M, expr)                                                                            \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess) {                                                                          \
      (M).err = std::string(#expr) + ": " + hipGetErrorString(e_);                                   \
      return e_ == hipErrorOutOfMemory ? APSS_E_NOMEM : APSS_E_DEVICE;                               \
    }                                                                                                \
  } while (0)

template <class T>
int32_t ensure(apss_group::Member &M, DevBuf<T> &b, size_t n) {
  if (n <= b.cap && b.p) return APSS_OK;
  const size_t ncap = std::max<size_t>(std::max(n, b.cap + b.cap / 2), 256);
  if (b.p) {
    GHIP(nullptr, M, hipStreamSynchronize(M.stream));
    GHIP(nullptr, M, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  GHIP(nullptr, M, hipMalloc((void **)&b.p, ncap * sizeof(T)));
  b.cap = ncap;
  return APSS_OK;
}

template <class T>
void release(DevBuf<T> &b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

// one batch as the caller handed it in
struct Batch {
  int64_t n = 0, nnz = 0;
  // host form
  const int64_t *rowptr = nullptr;
  const int32_t *indices = nullptr;
  const double *values = nullptr;
  const int64_t *ext = nullptr;
  // device form (per member)
  const int64_t *const *d_rowptr = nullptr;
  const int32_t *const *d_indices = nullptr;
  const float *const *d_values = nullptr;
  const int64_t *const *d_ext = nullptr;
  bool on_device = false;
};

// contiguous term ranges with (nearly) equal sum of df^2 (= the posting visits of a self-join), each non-empty
std::vector<int32_t> balanced_cuts(const std::vector<uint32_t> &df, int T) {
  const int32_t dim = (int32_t)df.size();
  std::vector<double> cum((size_t)dim + 1, 0.0);
  for (int32_t t = 0; t < dim; ++t) cum[(size_t)t + 1] = cum[(size_t)t] + (double)df[(size_t)t] * (double)df[(size_t)t];
  std::vector<int32_t> cuts((size_t)T + 1, 0);
  for (int g = 1; g < T; ++g) {
    const double want = cum.back() * (double)g / (double)T;
    cuts[(size_t)g] = (int32_t)(std::lower_bound(cum.begin(), cum.end(), want) - cum.begin());
  }
  cuts[(size_t)T] = dim;
  for (int g = 1; g <= T; ++g) cuts[(size_t)g] = std::max(cuts[(size_t)g], cuts[(size_t)g - 1] + 1);
  cuts[(size_t)T] = dim;
  for (int g = T - 1; g >= 1; --g) cuts[(size_t)g] = std::min(cuts[(size_t)g], cuts[(size_t)g + 1] - 1);
  return cuts;
}

std::vector<int32_t> equal_cuts(int32_t dim, int T) {
  std::vector<int32_t> cuts((size_t)T + 1, 0);
  for (int g = 0; g <= T; ++g) cuts[(size_t)g] = (int32_t)((int64_t)dim * g / T);
  return cuts;
}

// Decide the layout from the first batch: the shared dense-head block (member 0's device runs the library's own policy on a
// sample of the rows: a plain handle indexes them and is asked which terms it took) and the term cuts, then create the
// members' shard handles, their streams and -- when every member has its own GPU -- the RCCL communicators.
int32_t create_members(apss_group *g, const Batch &b) {
  const int T = g->T;
  const int32_t dim = g->cfg.dim;
  apss_group::Member &M0 = g->m[0];
  std::vector<uint32_t> df;
  const bool need_df = T > 1 && (!g->cuts_named || g->cfg.head_terms == 0);
  if (need_df) {
    df.assign((size_t)dim, 0u);
    if (!b.on_device) {
      for (int64_t i = 0; i < b.nnz; ++i) {
        const int32_t t = b.indices[i];
        if (t >= 0 && t < dim) ++df[(size_t)t];
      }
    } else if (b.nnz > 0) {
      if (hipSetDevice(M0.dev) != hipSuccess) return gfail(g, APSS_E_DEVICE, "hipSetDevice failed");
      unsigned int *d_df = nullptr;
      hipError_t e = hipMalloc((void **)&d_df, (size_t)dim * sizeof(unsigned int));
      if (e == hipSuccess) e = hipMemsetAsync(d_df, 0, (size_t)dim * sizeof(unsigned int), M0.stream);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(k_df_hist, dim3(grid_for(b.nnz)), dim3(256), 0, M0.stream, b.d_indices[0], b.nnz, dim, d_df);
        e = hipGetLastError();
      }
      if (e == hipSuccess) e = hipMemcpyAsync(df.data(), d_df, (size_t)dim * sizeof(unsigned int), hipMemcpyDeviceToHost, M0.stream);
      if (e == hipSuccess) e = hipStreamSynchronize(M0.stream);
      if (d_df) (void)hipFree(d_df);
      if (e != hipSuccess) return gfail(g, APSS_E_DEVICE, std::string("document frequencies: ") + hipGetErrorString(e));
    }
  }
  g->head.clear();
  if (T > 1 && g->cfg.head_terms >= 0 && g->cfg.theta > 0.0 && b.n > 0 &&
      !(g->cfg.flags & (APSS_FLAG_EXACT_ACCUM | APSS_FLAG_FORCE_GENERAL | APSS_FLAG_FORCE_SCAN | APSS_FLAG_ADMISSION))) {
    apss_config pc = g->cfg;
    pc.struct_size = (int32_t)sizeof(apss_config);
    pc.device_id = M0.dev;
    pc.term_lo = pc.term_hi = 0;
    pc.capacity_rows = pc.capacity_nnz = 0;
    apss_handle *ph = nullptr;
    int32_t rc = apss_create(&pc, &ph);
    if (rc != APSS_OK) return gfail(g, rc, std::string("head policy handle: ") + apss_last_error(nullptr));
    const int64_t sample = std::min<int64_t>(b.n, 131072);
    if (!b.on_device) {
      rc = apss_insert(ph, sample, b.rowptr, b.indices, b.values, b.ext);
    } else {
      int64_t e_end = 0;
      (void)hipSetDevice(M0.dev);
      if (hipMemcpy(&e_end, b.d_rowptr[0] + sample, sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess) rc = APSS_E_DEVICE;
      else rc = apss_insert_dev(ph, sample, e_end, b.d_rowptr[0], b.d_indices[0], b.d_values[0], b.d_ext[0]);
    }
    int32_t nt = 0;
    if (rc == APSS_OK) rc = apss_get_head_terms(ph, 0, nullptr, &nt);
    if (rc == APSS_OK && nt > 0) {
      g->head.resize((size_t)nt);
      rc = apss_get_head_terms(ph, nt, g->head.data(), &nt);
    }
    const std::string perr = rc == APSS_OK ? "" : apss_last_error(ph);
    apss_destroy(ph);
    if (rc != APSS_OK) return gfail(g, rc, "head policy handle: " + perr);
    if (g->cfg.head_terms == 0 && g->head.size() > 256) {
      // A deep head leaves few tail terms per row, and T term ranges cut them T ways: a row with a single term in a range
      // pairs up at within-range cosine 1 with every row sharing it and the candidate rule stops being selective.  Keep the
      // head only as deep as leaves about 8 tail terms per row and member, never shallower than the 256 terms that hold the
      // long posting lists (measured on C3 with Zipf(1) terms).
      double total = 0;
      for (uint32_t f : df) total += (double)f;
      size_t k = g->head.size();
      for (;;) {
        double in_head = 0;
        for (size_t i = 0; i < k; ++i) in_head += (double)df[(size_t)g->head[i]];
        if (k <= 256 || (total - in_head) / (double)std::max<int64_t>(1, b.n) / (double)T >= 8.0) break;
        k /= 2;
      }
      g->head.resize(std::max<size_t>(k, 256));
    }
  }
  if (!g->cuts_named) {
    if (T == 1) {
      g->cuts = {0, dim};
    } else if (b.n < 1024) {
      g->cuts = equal_cuts(dim, T);  // (a first batch this small says nothing about the term distribution)
    } else {
      for (int32_t t : g->head) df[(size_t)t] = 0;  // the block's terms are in no member's index: balance the tail's visits
      g->cuts = balanced_cuts(df, T);
    }
  }
  if (dim < T) return gfail(g, APSS_E_INVALID, "more members than terms");

  for (int i = 0; i < T; ++i) {
    apss_group::Member &M = g->m[(size_t)i];
    apss_config c = g->cfg;
    c.struct_size = (int32_t)sizeof(apss_config);
    c.device_id = M.dev;
    c.term_lo = T == 1 ? 0 : g->cuts[(size_t)i];
    c.term_hi = T == 1 ? 0 : g->cuts[(size_t)i + 1];
    int32_t rc = apss_create(&c, &M.h);
    if (rc != APSS_OK) return gfail(g, rc, std::string("member handle: ") + apss_last_error(nullptr));
    if ((rc = apss_set_stream(M.h, (void *)M.stream, 0)) != APSS_OK) return gfail(g, rc, apss_last_error(M.h));
    if (!g->head.empty() && (rc = apss_set_head_terms(M.h, (int32_t)g->head.size(), g->head.data(), i, T)) != APSS_OK)
      return gfail(g, rc, std::string("member head block: ") + apss_last_error(M.h));
  }
  const bool exchange_needed = T > 1 || (g->flags & APSS_GROUP_FORCE_EXCHANGE);
  g->exchange = !exchange_needed ? APSS_EXCHANGE_NONE : APSS_EXCHANGE_COPIES;
  if (exchange_needed && g->distinct_devices && !(g->flags & APSS_GROUP_NO_RCCL)) {
    Rccl *r = load_rccl();
    if (!r->lib) return gfail(g, APSS_E_UNSUPPORTED, "RCCL exchange: " + r->err + " (APSS_GROUP_NO_RCCL combines the members by copies)");
    std::vector<ncclComm_t> comms((size_t)T, nullptr);
    std::vector<int> devs;
    for (const apss_group::Member &M : g->m) devs.push_back(M.dev);
    const ncclResult_t nr = r->CommInitAll(comms.data(), T, devs.data());
    if (nr != ncclSuccess) return gfail(g, APSS_E_DEVICE, std::string("ncclCommInitAll: ") + r->GetErrorString(nr));
    for (int i = 0; i < T; ++i) g->m[(size_t)i].comm = comms[(size_t)i];
    g->exchange = APSS_EXCHANGE_RCCL;
  }
  g->created = true;
  return APSS_OK;
}

// ---- one member's share of a call; every member thread runs this, the barriers keep the phases in step.  A failure is
// published BEFORE the next barrier, so after it every thread sees the same flag and they leave (or skip a collective) together.
struct CallCtx {
  apss_group *g;
  int mode;  // 0 insert, 1 query (frozen index), 2 insert-and-query
  Batch b;
  Barrier *bar;
  std::atomic<int> failed{0};
  double exchange_ms = 0;
};

int32_t member_phase1(apss_group::Member &M, int i, CallCtx &cx) {
  const Batch &b = cx.b;
  int64_t n_res = 0;
  int32_t rc;
  if (!b.on_device) {
    if (cx.mode == 0) rc = apss_insert(M.h, b.n, b.rowptr, b.indices, b.values, b.ext);
    else if (cx.mode == 1) rc = apss_query(M.h, b.n, b.rowptr, b.indices, b.values, b.ext, &n_res);
    else rc = apss_insert_and_query(M.h, b.n, b.rowptr, b.indices, b.values, b.ext, &n_res);
  } else {
    rc = apss_insert_and_query_dev(M.h, b.n, b.nnz, b.d_rowptr[i], b.d_indices[i], b.d_values[i], b.d_ext[i], &n_res);
  }
  if (rc != APSS_OK) M.err = apss_last_error(M.h);
  M.n_cand = n_res;
  return rc;
}

int32_t member_pack(apss_group::Member &M, int64_t total, int64_t my_off) {
  int32_t rc;
  if ((rc = ensure(M, M.keys, (size_t)total)) != APSS_OK) return rc;
  if ((rc = ensure(M, M.sorted, (size_t)total)) != APSS_OK) return rc;
  if ((rc = ensure(M, M.uniq, (size_t)total)) != APSS_OK) return rc;
  if ((rc = ensure(M, M.count, 4)) != APSS_OK) return rc;
  if (M.n_cand > 0) {
    const int32_t *dq = nullptr, *dc = nullptr;
    const float *ds = nullptr;
    int64_t n = 0;
    if ((rc = apss_results_dev(M.h, &dq, &dc, &ds, &n)) != APSS_OK) {
      M.err = apss_last_error(M.h);
      return rc;
    }
    hipLaunchKernelGGL(k_pack_keys, dim3(grid_for(M.n_cand)), dim3(256), 0, M.stream, dq, dc, M.n_cand, M.keys.p + my_off);
    GHIP(nullptr, M, hipGetLastError());
  }
  GHIP(nullptr, M, hipStreamSynchronize(M.stream));  // (the copies exchange lets the peers read this list)
  return APSS_OK;
}

int32_t member_union(apss_group::Member &M, int64_t total, int key_bits) {
  size_t a = 0, bsz = 0;
  GHIP(nullptr, M, rocprim::radix_sort_keys(nullptr, a, M.keys.p, M.sorted.p, (size_t)total, 0, (unsigned)key_bits, M.stream));
  GHIP(nullptr, M, rocprim::unique(nullptr, bsz, M.sorted.p, M.uniq.p, M.count.p, (size_t)total, rocprim::equal_to<unsigned long long>(), M.stream));
  int32_t rc;
  if ((rc = ensure(M, M.tmp, std::max(a, bsz) + 256)) != APSS_OK) return rc;
  GHIP(nullptr, M, rocprim::radix_sort_keys((void *)M.tmp.p, a, M.keys.p, M.sorted.p, (size_t)total, 0, (unsigned)key_bits, M.stream));
  GHIP(nullptr, M, rocprim::unique((void *)M.tmp.p, bsz, M.sorted.p, M.uniq.p, M.count.p, (size_t)total, rocprim::equal_to<unsigned long long>(), M.stream));
  unsigned long long nu = 0;
  GHIP(nullptr, M, hipMemcpyAsync(&nu, M.count.p, sizeof(nu), hipMemcpyDeviceToHost, M.stream));
  GHIP(nullptr, M, hipStreamSynchronize(M.stream));
  M.n_union = (int64_t)nu;
  if ((rc = ensure(M, M.uq, (size_t)std::max<int64_t>(1, M.n_union))) != APSS_OK) return rc;
  if ((rc = ensure(M, M.uc, (size_t)std::max<int64_t>(1, M.n_union))) != APSS_OK) return rc;
  if ((rc = ensure(M, M.partial, (size_t)std::max<int64_t>(1, M.n_union))) != APSS_OK) return rc;
  if ((rc = ensure(M, M.sum, (size_t)std::max<int64_t>(1, M.n_union))) != APSS_OK) return rc;
  if (M.n_union > 0) {
    hipLaunchKernelGGL(k_unpack_keys, dim3(grid_for(M.n_union)), dim3(256), 0, M.stream, (const unsigned long long *)M.uniq.p, M.n_union, M.uq.p, M.uc.p);
    GHIP(nullptr, M, hipGetLastError());
  }
  return APSS_OK;
}

void member_main(int i, CallCtx *pcx) {
  CallCtx &cx = *pcx;
  apss_group *g = cx.g;
  apss_group::Member &M = g->m[(size_t)i];
  const int T = g->T;
  Rccl *r = g->exchange == APSS_EXCHANGE_RCCL ? load_rccl() : nullptr;
  M.rc = APSS_OK;
  M.err.clear();
  M.n_cand = M.n_union = 0;
  M.member_ms = M.partial_ms = 0;
  auto fail_here = [&](int32_t rc) {
    M.rc = rc;
    cx.failed.store(1);
  };
  if (hipSetDevice(M.dev) != hipSuccess) {
    M.err = "hipSetDevice failed";
    fail_here(APSS_E_DEVICE);
  }
  const auto t0 = Clock::now();
  if (!cx.failed.load()) {
    const int32_t rc = member_phase1(M, i, cx);
    if (rc != APSS_OK) fail_here(rc);
  }
  M.member_ms = ms_since(t0);
  cx.bar->wait();  // ---- B1: every member's candidate count (or failure) is known
  if (cx.failed.load() || cx.mode == 0 || g->exchange == APSS_EXCHANGE_NONE) return;
  const auto tx = Clock::now();
  int64_t total = 0, my_off = 0;
  std::vector<int64_t> off((size_t)T + 1, 0);
  for (int k = 0; k < T; ++k) {
    if (k == i) my_off = total;
    off[(size_t)k] = total;
    total += g->m[(size_t)k].n_cand;
  }
  off[(size_t)T] = total;
  if (total == 0) {  // nothing to exchange: every member leaves together
    if (i == 0) g->n_res = 0;
    return;
  }
  {
    const int32_t rc = member_pack(M, total, my_off);
    if (rc != APSS_OK) fail_here(rc);
  }
  cx.bar->wait();  // ---- B2: every list is packed
  if (cx.failed.load()) return;
  // ---- step 2: all-gather of the candidate lists (in place: member k's list sits at off[k] in every member's buffer)
  int32_t rc = APSS_OK;
  if (g->exchange == APSS_EXCHANGE_RCCL) {
    ncclResult_t nr = r->GroupStart();
    for (int k = 0; k < T && nr == ncclSuccess; ++k) {
      const int64_t nk = g->m[(size_t)k].n_cand;
      if (nk > 0) nr = r->Broadcast(M.keys.p + off[(size_t)k], M.keys.p + off[(size_t)k], (size_t)nk, ncclUint64, k, M.comm, M.stream);
    }
    const ncclResult_t ne = r->GroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) {
      M.err = std::string("RCCL all-gather of candidate lists: ") + r->GetErrorString(nr);
      rc = APSS_E_DEVICE;
    }
  } else {
    for (int k = 0; k < T && rc == APSS_OK; ++k) {
      const int64_t nk = g->m[(size_t)k].n_cand;
      if (k == i || nk == 0) continue;
      const hipError_t e = hipMemcpyAsync(M.keys.p + off[(size_t)k], g->m[(size_t)k].keys.p + off[(size_t)k], (size_t)nk * sizeof(unsigned long long),
                                          hipMemcpyDefault, M.stream);
      if (e != hipSuccess) {
        M.err = std::string("candidate list copy: ") + hipGetErrorString(e);
        rc = APSS_E_DEVICE;
      }
    }
  }
  // sorted union (the same list in the same order on every member), then this member's exact partial score of every pair
  if (rc == APSS_OK) {
    int q_bits = 1;
    while (q_bits < 31 && (1LL << q_bits) < std::max<int64_t>(2, cx.b.n)) ++q_bits;
    rc = member_union(M, total, 32 + q_bits);
  }
  if (rc == APSS_OK && M.n_union > 0) {
    const auto tp = Clock::now();
    rc = apss_partial_scores_dev(M.h, M.n_union, M.uq.p, M.uc.p, M.partial.p);  // (synchronises the member's stream)
    if (rc != APSS_OK) M.err = apss_last_error(M.h);
    M.partial_ms = ms_since(tp);
  }
  if (rc != APSS_OK) fail_here(rc);
  cx.bar->wait();  // ---- B3: every member's partial scores are complete
  if (cx.failed.load()) return;
  const int64_t nu = M.n_union;
  if (g->m[0].n_union != nu) {  // (cannot happen: the same sort of the same keys; checked because a collective of unequal counts hangs)
    M.err = "members disagree on the candidate union";
    fail_here(APSS_E_STATE);
  }
  cx.bar->wait();  // ---- B3b
  if (cx.failed.load()) return;
  // ---- step 4: all-reduce(SUM) of the partial scores
  const float *total_score = M.partial.p;
  if (!cx.failed.load() && nu > 0) {
    if (g->exchange == APSS_EXCHANGE_RCCL) {
      const ncclResult_t nr = r->AllReduce(M.partial.p, M.sum.p, (size_t)nu, ncclFloat, ncclSum, M.comm, M.stream);
      if (nr != ncclSuccess) {
        M.err = std::string("RCCL all-reduce of partial scores: ") + r->GetErrorString(nr);
        fail_here(APSS_E_DEVICE);
      }
      total_score = M.sum.p;
    } else if (i == 0) {
      hipError_t e = hipMemcpyAsync(M.sum.p, M.partial.p, (size_t)nu * sizeof(float), hipMemcpyDeviceToDevice, M.stream);
      for (int k = 1; k < T && e == hipSuccess; ++k) {
        const float *src = g->m[(size_t)k].partial.p;
        if (g->m[(size_t)k].dev != M.dev) {  // (another device: bring the vector over first)
          if (ensure(M, M.stage, (size_t)nu) != APSS_OK) {
            e = hipErrorOutOfMemory;
            break;
          }
          e = hipMemcpyAsync(M.stage.p, src, (size_t)nu * sizeof(float), hipMemcpyDefault, M.stream);
          src = M.stage.p;
        }
        if (e == hipSuccess) {
          hipLaunchKernelGGL(k_accumulate, dim3(grid_for(nu)), dim3(256), 0, M.stream, M.sum.p, src, nu);
          e = hipGetLastError();
        }
      }
      if (e != hipSuccess) {
        M.err = std::string("partial score reduction: ") + hipGetErrorString(e);
        fail_here(e == hipErrorOutOfMemory ? APSS_E_NOMEM : APSS_E_DEVICE);
      }
      total_score = M.sum.p;
    }
  }
  // ---- the `>= theta` prune, on member 0 (every member of the RCCL exchange holds the same sums)
  if (i == 0 && !cx.failed.load()) {
    auto finish = [&]() -> int32_t {
      int32_t rc2;
      if ((rc2 = ensure(M, g->res_q, (size_t)std::max<int64_t>(1, nu))) != APSS_OK) return rc2;
      if ((rc2 = ensure(M, g->res_c, (size_t)std::max<int64_t>(1, nu))) != APSS_OK) return rc2;
      if ((rc2 = ensure(M, g->res_s, (size_t)std::max<int64_t>(1, nu))) != APSS_OK) return rc2;
      GHIP(nullptr, M, hipMemsetAsync(M.count.p, 0, sizeof(unsigned long long), M.stream));
      if (nu > 0) {
        hipLaunchKernelGGL(k_threshold_compact, dim3(grid_for(nu)), dim3(256), 0, M.stream, total_score, (const int32_t *)M.uq.p,
                           (const int32_t *)M.uc.p, nu, (float)g->cfg.theta, g->res_q.p, g->res_c.p, g->res_s.p, M.count.p);
        GHIP(nullptr, M, hipGetLastError());
      }
      unsigned long long nres = 0;
      GHIP(nullptr, M, hipMemcpyAsync(&nres, M.count.p, sizeof(nres), hipMemcpyDeviceToHost, M.stream));
      GHIP(nullptr, M, hipStreamSynchronize(M.stream));
      g->n_res = (int64_t)nres;
      return APSS_OK;
    };
    const int32_t rc2 = finish();
    if (rc2 != APSS_OK) fail_here(rc2);
    cx.exchange_ms = ms_since(tx);
  } else if (!cx.failed.load()) {
    if (hipStreamSynchronize(M.stream) != hipSuccess) {
      M.err = "stream synchronisation failed after the exchange";
      fail_here(APSS_E_DEVICE);
    }
  }
  cx.bar->wait();  // ---- B4: member 0 has read every peer's partial scores (copies exchange) before anyone returns
}

int32_t run_call(apss_group *g, int mode, const Batch &b, int64_t *n_results) {
  if (n_results) *n_results = 0;
  g->err.clear();
  if (b.n < 0 || b.nnz < 0) return gfail(g, APSS_E_INVALID, "negative size");
  g->n_res = -1;
  g->results_in_handle = false;
  if (!g->created) {
    if (mode == 1 || b.n == 0) {  // nothing is indexed yet: an empty answer, no layout decided
      g->n_res = mode == 0 ? -1 : 0;
      return APSS_OK;
    }
    const int32_t rc = create_members(g, b);
    if (rc != APSS_OK) {
      for (apss_group::Member &M : g->m) {
        if (M.h) apss_destroy(M.h);
        M.h = nullptr;
      }
      return rc;
    }
  }
  const auto t0 = Clock::now();
  CallCtx cx;
  cx.g = g;
  cx.mode = mode;
  cx.b = b;
  Barrier bar(g->T);
  cx.bar = &bar;
  std::vector<std::thread> th;
  for (int i = 1; i < g->T; ++i) th.emplace_back(member_main, i, &cx);
  member_main(0, &cx);  // (member 0 runs on the caller's thread)
  for (std::thread &t : th) t.join();
  if (cx.failed.load()) {
    g->n_res = -1;
    for (int i = 0; i < g->T; ++i)
      if (g->m[(size_t)i].rc != APSS_OK)
        return gfail(g, g->m[(size_t)i].rc, "member " + std::to_string(i) + ": " + g->m[(size_t)i].err);
    return gfail(g, APSS_E_STATE, "a member failed");
  }
  if (mode != 1) {
    int64_t rows = 0;
    (void)apss_size(g->m[0].h, &rows, nullptr);
    g->n_rows = rows;
  }
  // statistics of the call
  apss_group_stats &st = g->st;
  const int32_t keep_size = st.struct_size;
  st = apss_group_stats{};
  st.struct_size = keep_size;
  st.n_members = g->T;
  st.exchange = g->exchange;
  st.head_terms = (int32_t)g->head.size();
  for (int i = 0; i <= g->T && i <= APSS_GROUP_MAX_MEMBERS; ++i) st.term_cuts[i] = g->cuts[(size_t)i];
  for (int i = 0; i < g->T; ++i) {
    apss_group::Member &M = g->m[(size_t)i];
    apss_stats ms{};
    ms.struct_size = (int32_t)sizeof(apss_stats);
    (void)apss_stats_get(M.h, &ms);
    st.nnz += ms.nnz;
    st.rows = ms.rows;
    if (mode != 0) {
      st.posting_visits += ms.posting_visits;
      st.device_posting_visits += ms.device_posting_visits;
      st.member_touched_pairs += ms.candidate_pairs;
      st.candidates_sum += M.n_cand;
      st.candidates_max = std::max(st.candidates_max, M.n_cand);
      st.probe_ms_max = std::max(st.probe_ms_max, ms.probe_ms);
      st.head_ms_max = std::max(st.head_ms_max, ms.head_ms);
    }
    st.build_ms_max = std::max(st.build_ms_max, ms.build_ms);
    st.member_ms_max = std::max(st.member_ms_max, M.member_ms);
    st.partial_ms_max = std::max(st.partial_ms_max, M.partial_ms);
  }
  if (mode != 0) {
    if (g->exchange == APSS_EXCHANGE_NONE) {
      g->results_in_handle = true;
      g->n_res = g->m[0].n_cand;
      st.union_pairs = g->n_res;
    } else {
      st.union_pairs = g->m[0].n_union;
      int64_t least = st.candidates_max;
      for (const apss_group::Member &M : g->m) least = std::min(least, M.n_cand);
      st.all_gather_bytes = 8 * (st.candidates_sum - least);
      st.all_reduce_bytes = 4 * st.union_pairs;
      st.exchange_ms = cx.exchange_ms;
    }
    st.result_pairs = g->n_res;
    if (n_results) *n_results = g->n_res;
  }
  st.total_ms = ms_since(t0);
  return APSS_OK;
}

int32_t validate_host(apss_group *g, int64_t n, const int64_t *rowptr, const int32_t *indices, const double *values, const int64_t *ext) {
  if (n < 0) return gfail(g, APSS_E_INVALID, "negative row count");
  if (n == 0) return APSS_OK;
  if (!rowptr || !ext) return gfail(g, APSS_E_INVALID, "null rowptr / ext_ids");
  if (rowptr[0] != 0) return gfail(g, APSS_E_INVALID, "rowptr[0] must be 0");
  for (int64_t i = 0; i < n; ++i)
    if (rowptr[i + 1] < rowptr[i]) return gfail(g, APSS_E_INVALID, "rowptr must be non-decreasing");
  if (rowptr[n] > 0 && (!indices || !values)) return gfail(g, APSS_E_INVALID, "null indices / values");
  return APSS_OK;
}

Batch host_batch(int64_t n, const int64_t *rowptr, const int32_t *indices, const double *values, const int64_t *ext) {
  Batch b;
  b.n = n;
  b.nnz = n > 0 ? rowptr[n] : 0;
  b.rowptr = rowptr;
  b.indices = indices;
  b.values = values;
  b.ext = ext;
  return b;
}

}  // namespace

extern "C" {

int32_t apss_group_create(const apss_config *cfg, int32_t n_members, const int32_t *device_ids, uint32_t group_flags, apss_group **out) {
  if (!cfg || !out || !device_ids) {
    g_group_create_error = "null argument";
    return APSS_E_INVALID;
  }
  *out = nullptr;
  if (cfg->struct_size != (int32_t)sizeof(apss_config)) {
    g_group_create_error = "apss_config.struct_size mismatch";
    return APSS_E_INVALID;
  }
  if (n_members < 1 || n_members > APSS_GROUP_MAX_MEMBERS) {
    g_group_create_error = "n_members must be in [1, 64]";
    return APSS_E_INVALID;
  }
  if (cfg->dim <= 0 || !std::isfinite(cfg->theta)) {
    g_group_create_error = "dim must be > 0 and theta finite";
    return APSS_E_INVALID;
  }
  if (n_members > 1 && !(cfg->theta > 0.0)) {
    g_group_create_error = "term-range shards need theta > 0 (candidate test p_g >= theta*|q_g|*|c_g|)";
    return APSS_E_UNSUPPORTED;
  }
  int ndev = 0;
  const hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_group_create_error = std::string("no usable HIP device (there is no CPU fallback): ") + (e != hipSuccess ? hipGetErrorString(e) : "none found");
    return APSS_E_DEVICE;
  }
  apss_group *g = new (std::nothrow) apss_group();
  if (!g) return APSS_E_NOMEM;
  g->cfg = *cfg;
  g->flags = group_flags;
  g->T = n_members;
  g->m.resize((size_t)n_members);
  g->st.struct_size = (int32_t)sizeof(apss_group_stats);
  for (int i = 0; i < n_members; ++i) {
    const int d = device_ids[i];
    if (d < 0 || d >= ndev) {
      g_group_create_error = "device ordinal out of range";
      apss_group_destroy(g);
      return APSS_E_DEVICE;
    }
    for (int k = 0; k < i; ++k)
      if (g->m[(size_t)k].dev == d) g->distinct_devices = false;
    g->m[(size_t)i].dev = d;
    if (hipSetDevice(d) != hipSuccess || hipStreamCreateWithFlags(&g->m[(size_t)i].stream, hipStreamDefault) != hipSuccess) {
      g_group_create_error = "HIP stream creation failed";
      apss_group_destroy(g);
      return APSS_E_DEVICE;
    }
  }
  *out = g;
  return APSS_OK;
}

void apss_group_destroy(apss_group *g) {
  if (!g) return;
  Rccl *r = g->exchange == APSS_EXCHANGE_RCCL ? load_rccl() : nullptr;
  for (apss_group::Member &M : g->m) {
    (void)hipSetDevice(M.dev);
    if (M.stream) (void)hipStreamSynchronize(M.stream);
    if (M.comm && r && r->lib) (void)r->CommDestroy(M.comm);
    if (M.h) apss_destroy(M.h);
    release(M.keys); release(M.sorted); release(M.uniq); release(M.count); release(M.tmp);
    release(M.uq); release(M.uc); release(M.partial); release(M.sum); release(M.stage);
  }
  if (!g->m.empty()) (void)hipSetDevice(g->m[0].dev);
  release(g->res_q); release(g->res_c); release(g->res_s); release(g->ext_q); release(g->ext_c);
  for (apss_group::Member &M : g->m)
    if (M.stream) {
      (void)hipSetDevice(M.dev);
      (void)hipStreamDestroy(M.stream);
    }
  delete g;
}

const char *apss_group_last_error(const apss_group *g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }

int32_t apss_group_set_term_cuts(apss_group *g, const int32_t *cuts) {
  if (!g || !cuts) return APSS_E_INVALID;
  if (g->created) return gfail(g, APSS_E_STATE, "apss_group_set_term_cuts: the members exist already (name the cuts before the first insert)");
  if (cuts[0] != 0 || cuts[g->T] != g->cfg.dim) return gfail(g, APSS_E_INVALID, "apss_group_set_term_cuts: cuts[0] = 0 and cuts[n_members] = dim");
  for (int i = 0; i < g->T; ++i)
    if (cuts[i] >= cuts[i + 1]) return gfail(g, APSS_E_INVALID, "apss_group_set_term_cuts: cuts must be strictly increasing");
  g->cuts.assign(cuts, cuts + g->T + 1);
  g->cuts_named = true;
  return APSS_OK;
}

int32_t apss_group_insert(apss_group *g, int64_t n, const int64_t *rowptr, const int32_t *indices, const double *values,
                          const int64_t *ext_ids) {
  if (!g) return APSS_E_INVALID;
  int32_t rc = validate_host(g, n, rowptr, indices, values, ext_ids);
  if (rc != APSS_OK) return rc;
  if (n == 0) return APSS_OK;
  return run_call(g, 0, host_batch(n, rowptr, indices, values, ext_ids), nullptr);
}

int32_t apss_group_query(apss_group *g, int64_t n, const int64_t *rowptr, const int32_t *indices, const double *values,
                         const int64_t *ext_ids, int64_t *n_results) {
  if (!g) return APSS_E_INVALID;
  int32_t rc = validate_host(g, n, rowptr, indices, values, ext_ids);
  if (rc != APSS_OK) return rc;
  return run_call(g, 1, host_batch(n, rowptr, indices, values, ext_ids), n_results);
}

int32_t apss_group_insert_and_query(apss_group *g, int64_t n, const int64_t *rowptr, const int32_t *indices, const double *values,
                                    const int64_t *ext_ids, int64_t *n_results) {
  if (!g) return APSS_E_INVALID;
  int32_t rc = validate_host(g, n, rowptr, indices, values, ext_ids);
  if (rc != APSS_OK) return rc;
  return run_call(g, 2, host_batch(n, rowptr, indices, values, ext_ids), n_results);
}

int32_t apss_group_insert_and_query_dev(apss_group *g, int64_t n, int64_t nnz, const int64_t *const *d_rowptr,
                                        const int32_t *const *d_indices, const float *const *d_values,
                                        const int64_t *const *d_ext_ids, int64_t *n_results) {
  if (!g) return APSS_E_INVALID;
  if (n < 0 || nnz < 0) return gfail(g, APSS_E_INVALID, "negative size");
  if (n > 0) {
    if (!d_rowptr || !d_indices || !d_values || !d_ext_ids) return gfail(g, APSS_E_INVALID, "null pointer table");
    for (int i = 0; i < g->T; ++i)
      if (!d_rowptr[i] || !d_ext_ids[i] || (nnz > 0 && (!d_indices[i] || !d_values[i]))) return gfail(g, APSS_E_INVALID, "null device pointer");
  }
  Batch b;
  b.n = n;
  b.nnz = nnz;
  b.d_rowptr = d_rowptr;
  b.d_indices = d_indices;
  b.d_values = d_values;
  b.d_ext = d_ext_ids;
  b.on_device = true;
  return run_call(g, 2, b, n_results);
}

int32_t apss_group_clear(apss_group *g) {
  if (!g) return APSS_E_INVALID;
  g->n_res = -1;
  g->results_in_handle = false;
  g->n_rows = 0;
  for (int i = 0; i < g->T; ++i) {
    apss_group::Member &M = g->m[(size_t)i];
    if (!M.h) continue;
    const int32_t rc = apss_clear(M.h);
    if (rc != APSS_OK) return gfail(g, rc, "member " + std::to_string(i) + ": " + apss_last_error(M.h));
  }
  return APSS_OK;
}

int32_t apss_group_result_count(const apss_group *g, int64_t *n_results) {
  if (!g || !n_results) return APSS_E_INVALID;
  if (g->n_res < 0) return APSS_E_STATE;
  *n_results = g->n_res;
  return APSS_OK;
}

int32_t apss_group_fetch_results(apss_group *g, int64_t offset, int64_t count, int64_t *out_q, int64_t *out_c, float *out_score) {
  if (!g) return APSS_E_INVALID;
  if (g->n_res < 0) return gfail(g, APSS_E_STATE, "no query has run on this group since the last insert");
  if (offset < 0 || count < 0 || offset + count > g->n_res) return gfail(g, APSS_E_INVALID, "fetch range out of bounds");
  if (count == 0) return APSS_OK;
  if (!out_q || !out_c || !out_score) return gfail(g, APSS_E_INVALID, "null output buffer");
  apss_group::Member &M = g->m[0];
  if (g->results_in_handle) {
    const int32_t rc = apss_fetch_results(M.h, offset, count, out_q, out_c, out_score);
    if (rc != APSS_OK) g->err = apss_last_error(M.h);
    return rc;
  }
  auto body = [&]() -> int32_t {
    GHIP(nullptr, M, hipSetDevice(M.dev));
    const int64_t *d_store = nullptr, *d_query = nullptr;
    int32_t rc = apss_ext_ids_dev(M.h, &d_store, &d_query);
    if (rc != APSS_OK || !d_store || !d_query) {
      M.err = "the members' external ids are gone (an insert since the last query-type call)";
      return APSS_E_STATE;
    }
    if ((rc = ensure(M, g->ext_q, (size_t)count)) != APSS_OK) return rc;
    if ((rc = ensure(M, g->ext_c, (size_t)count)) != APSS_OK) return rc;
    hipLaunchKernelGGL(k_gather_ext, dim3((unsigned)ceil_div(count, 256)), dim3(256), 0, M.stream, (const int32_t *)g->res_q.p + offset,
                       (const int32_t *)g->res_c.p + offset, d_query, d_store, count, g->ext_q.p, g->ext_c.p);
    GHIP(nullptr, M, hipGetLastError());
    GHIP(nullptr, M, hipMemcpyAsync(out_q, g->ext_q.p, (size_t)count * sizeof(int64_t), hipMemcpyDeviceToHost, M.stream));
    GHIP(nullptr, M, hipMemcpyAsync(out_c, g->ext_c.p, (size_t)count * sizeof(int64_t), hipMemcpyDeviceToHost, M.stream));
    GHIP(nullptr, M, hipMemcpyAsync(out_score, g->res_s.p + offset, (size_t)count * sizeof(float), hipMemcpyDeviceToHost, M.stream));
    GHIP(nullptr, M, hipStreamSynchronize(M.stream));
    return APSS_OK;
  };
  const int32_t rc = body();
  if (rc != APSS_OK) g->err = M.err;
  return rc;
}

int32_t apss_group_stats_get(apss_group *g, apss_group_stats *out) {
  if (!g || !out) return APSS_E_INVALID;
  const int32_t caller = out->struct_size;
  if (caller < (int32_t)(2 * sizeof(int32_t)) || caller > (1 << 16))
    return gfail(g, APSS_E_INVALID, "apss_group_stats.struct_size must be set to sizeof(apss_group_stats) before the call");
  const int32_t n = std::min<int32_t>(caller, (int32_t)sizeof(apss_group_stats));
  g->st.n_members = g->T;
  g->st.struct_size = n;
  std::memcpy(out, &g->st, (size_t)n);
  return APSS_OK;
}

int32_t apss_group_member_stats(apss_group *g, int32_t member, apss_stats *out) {
  if (!g || !out) return APSS_E_INVALID;
  if (member < 0 || member >= g->T) return gfail(g, APSS_E_INVALID, "no such member");
  if (!g->m[(size_t)member].h) return gfail(g, APSS_E_STATE, "the members are created by the group's first insert");
  const int32_t rc = apss_stats_get(g->m[(size_t)member].h, out);
  if (rc != APSS_OK) g->err = apss_last_error(g->m[(size_t)member].h);
  return rc;
}

}  // extern "C"
