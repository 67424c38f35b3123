// head_gemm_bench.hip -- stand-alone timing of k_head_gemm (apss_head.hpp) on random unit rows: TFLOP/s against the
// dense bf16 MFMA peak.  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o head_gemm_bench head_gemm_bench.hip
// Run:   ./head_gemm_bench [N=262144] [KH=256] [stored=1] [reps=3]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "head_gemm16.hpp"

using namespace apss;

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
      return 1;                                                                \
    }                                                                          \
  } while (0)

__global__ void k_fill(uint16_t *W, int64_t n, int kh, uint32_t seed) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  float v[256];
  float n2 = 0.f;
  for (int k = 0; k < kh; ++k) {
    uint32_t s = seed + (uint32_t)row * 0x9E3779B9u + (uint32_t)k * 0x85EBCA6Bu;  // murmur3 finaliser per (row, k)
    s ^= s >> 16; s *= 0x85EBCA6Bu; s ^= s >> 13; s *= 0xC2B2AE35u; s ^= s >> 16;
    const float x = ((s >> 8) & 0xffff) / 65536.0f;
    v[k] = (s & 3u) ? 0.f : x;  // a quarter of the entries are non-zero
    n2 += v[k] * v[k];
  }
  const float inv = n2 > 0.f ? 1.0f / sqrtf(n2) : 0.f;
  for (int k = 0; k < kh; ++k) W[head_chunk_off(row, k >> 3, kh) + (k & 7)] = f32_to_bf16_rn(v[k] * inv);
}

int main(int argc, char **argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 262144;
  const int kh = argc > 2 ? atoi(argv[2]) : 256;
  const int stored = argc > 3 ? atoi(argv[3]) : 1;
  const int reps = argc > 4 ? atoi(argv[4]) : 3;
  const int64_t rows_pad = (n + kHeadQBlock - 1) / kHeadQBlock * kHeadQBlock + kHeadCTile;
  uint16_t *W;
  int64_t *ext;
  int32_t *rq, *rc;
  float *rs;
  unsigned long long *ctr;
  const uint64_t cap = 1 << 24;
  CK(hipMalloc(&W, rows_pad * kh * 2));
  CK(hipMemset(W, 0, rows_pad * kh * 2));
  CK(hipMalloc(&ext, n * 8));
  CK(hipMalloc(&rq, cap * 4));
  CK(hipMalloc(&rc, cap * 4));
  CK(hipMalloc(&rs, cap * 4));
  CK(hipMalloc(&ctr, 64));
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, W, n, kh, 12345u);
  std::vector<int64_t> h_ext(n);
  for (int64_t i = 0; i < n; ++i) h_ext[i] = i;
  CK(hipMemcpy(ext, h_ext.data(), n * 8, hipMemcpyHostToDevice));
  HeadGemmArgs g{};
  g.Wq = W;
  g.Wc = W;
  g.wq_rows = rows_pad;
  g.n_rows = n;
  g.q_slot_base = stored ? 0 : -1;
  g.nq = (int32_t)n;
  g.qblock0 = 0;
  g.n_qblocks = (int32_t)((n + kHeadQBlock - 1) / kHeadQBlock);
  const int64_t ct = head_tile_rows(kh);
  g.n_ctiles = (int32_t)((n + ct - 1) / ct);
  int64_t panels = 8;
  while (panels * g.n_qblocks < 2048 && g.n_ctiles / (2 * panels) >= 32) panels *= 2;
  if (getenv("PANELS")) panels = atoi(getenv("PANELS"));
  g.n_panels = (int32_t)panels;
  g.q_ext = ext;
  g.c_ext = ext;
  g.thr = 0.8f;
  g.res_q = rq;
  g.res_c = rc;
  g.res_s = rs;
  g.res_cap = cap;
  g.counters = ctr;
  g.head_pairs = ctr + 4;
  g.kt = kh;
  g.chunk0 = 0;
  g.part = 0;
  g.n_parts = 1;
  const bool m16 = getenv("M16") != nullptr;  // the v_mfma_f32_16x16x32_bf16 form of the kernel
  const bool clk = getenv("CLK") != nullptr;  // diagnostic instantiation: in-kernel clock (s_memtime / s_memrealtime), M16 + KH 256 only
  double tiles = 0;
  for (int64_t b = 0; b < g.n_qblocks; ++b)
    tiles += stored ? (double)std::min<int64_t>(g.n_ctiles, (b + 1) * kHeadQBlock / ct) : (double)g.n_ctiles;
  const double flops = 2.0 * kh * tiles * ct * kHeadQBlock;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const dim3 grid((unsigned)((int64_t)g.n_qblocks * g.n_panels));
  unsigned long long *d_clk = nullptr;
  if (clk) {
    CK(hipMalloc(&d_clk, (size_t)grid.x * 16));
    CK(hipMemset(d_clk, 0, (size_t)grid.x * 16));
    g.clk = d_clk;
  }
  for (int r = 0; r < reps + 1; ++r) {
    CK(hipMemset(ctr, 0, 64));
    CK(hipEventRecord(e0, 0));
    if (getenv("NW4")) {  // experiment: two independent 4-wave workgroups per CU (256 query slots each, own barriers)
      HeadGemmArgs g4 = g;
      g4.n_qblocks = (int32_t)((n + 255) / 256);
      const dim3 grid4((unsigned)((int64_t)g4.n_qblocks * g4.n_panels));
      hipLaunchKernelGGL((k_head_gemm<256, true, 2, 4>), grid4, dim3(256), 0, 0, g4);
    } else if (clk) hipLaunchKernelGGL((k_head_gemm16<256, true, true>), grid, dim3(512), 0, 0, g);
    else if (m16 && kh == 64) hipLaunchKernelGGL(k_head_gemm16<64>, grid, dim3(512), 0, 0, g);
    else if (m16 && kh == 128) hipLaunchKernelGGL(k_head_gemm16<128>, grid, dim3(512), 0, 0, g);
    else if (m16 && getenv("NOCOUNT")) hipLaunchKernelGGL((k_head_gemm16<256, false>), grid, dim3(512), 0, 0, g);
    else if (m16) hipLaunchKernelGGL(k_head_gemm16<256>, grid, dim3(512), 0, 0, g);
    else if (getenv("NOCOUNT") && kh == 128 && getenv("PIPE3")) hipLaunchKernelGGL((k_head_gemm<128, false, 3>), grid, dim3(512), 0, 0, g);
    else if (getenv("NOCOUNT") && kh == 128) hipLaunchKernelGGL((k_head_gemm<128, false, 2>), grid, dim3(512), 0, 0, g);
    else if (getenv("PIPE4")) hipLaunchKernelGGL((k_head_gemm<256, true, 4>), grid, dim3(512), 0, 0, g);
    else if (getenv("PIPE2") && kh == 64) hipLaunchKernelGGL((k_head_gemm<64, true, 2>), grid, dim3(512), 0, 0, g);
    else if (getenv("PIPE2") && kh == 128) hipLaunchKernelGGL((k_head_gemm<128, true, 2>), grid, dim3(512), 0, 0, g);
    else if (getenv("PIPE2")) hipLaunchKernelGGL((k_head_gemm<256, true, 2>), grid, dim3(512), 0, 0, g);
    else if (kh == 64) hipLaunchKernelGGL(k_head_gemm<64>, grid, dim3(512), 0, 0, g);
    else if (kh == 128) hipLaunchKernelGGL(k_head_gemm<128>, grid, dim3(512), 0, 0, g);
    else if (getenv("NOCOUNT")) hipLaunchKernelGGL((k_head_gemm<256, false>), grid, dim3(512), 0, 0, g);
    else hipLaunchKernelGGL(k_head_gemm<256>, grid, dim3(512), 0, 0, g);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c[8];
    CK(hipMemcpy(c, ctr, 64, hipMemcpyDeviceToHost));
    if (r) printf("%s N=%lld KH=%d stored=%d panels=%lld grid=%u: %.3f ms  %.1f TFLOP/s (%.3f of 2500)  results=%llu positive=%llu\n",
                  m16 ? "16x16x32" : "32x32x16", (long long)n, kh, stored, (long long)panels, grid.x, ms, flops / ms / 1e9, flops / ms / 1e9 / 2500.0, c[0], c[4]);
  }
  if (clk) {
    // median over the workgroups that ran a sizeable loop (after `reps` back-to-back launches on random data)
    std::vector<unsigned long long> h((size_t)grid.x * 2);
    CK(hipMemcpy(h.data(), d_clk, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> mhz;
    for (size_t i = 0; i < grid.x; ++i)
      if (h[2 * i + 1] > 1000) mhz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
    std::sort(mhz.begin(), mhz.end());
    if (!mhz.empty())
      printf("in-kernel clock: median %.0f MHz (p10 %.0f, p90 %.0f) over %zu workgroups\n", mhz[mhz.size() / 2], mhz[mhz.size() / 10],
             mhz[mhz.size() * 9 / 10], mhz.size());
  }
  return 0;
}
