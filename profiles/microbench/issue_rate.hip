// How long does one wave take per "posting" of the filter's add pattern, and what does the exec masking of idle lanes cost?
// Build: hipcc -O3 --offload-arch=gfx950 -o issue_rate issue_rate.hip ; run: ./issue_rate
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int VARIANT>
__global__ __launch_bounds__(512) void k(const uint32_t *post, unsigned long long *out, int iters, float wq) {
  __shared__ uint32_t acc[16384 + 64];
  for (int i = threadIdx.x; i < 16384 + 64; i += blockDim.x) acc[i] = 0;
  __syncthreads();
  unsigned char *smem = reinterpret_cast<unsigned char *>(acc);
  const uint32_t thr1 = 200u;
  uint32_t cands = 0, crossings = 0;
  const uint32_t *p = post + threadIdx.x * 4;
  uint32_t pc[4], st[4];  // st: the lane's own sequence (never zeroed: idle lanes must not fall into a common sequence)
  for (int j = 0; j < 4; ++j) pc[j] = st[j] = p[j];
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    uint32_t o[4], pr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t w = pc[j];
      const float x = __builtin_fmaf(wq, __half2float(__ushort_as_half((unsigned short)(w >> 16))), 1.0f);
      pr[j] = (uint32_t)x;
      if (VARIANT == 0) {  // exec-masked idle lanes (the kernel's pattern)
        o[j] = thr1 + 1u;
        if (w) o[j] = atomicAdd(reinterpret_cast<uint32_t *>(smem + (w & 0xfffcu)), pr[j] << ((w << 3) & 31u));
      } else if (VARIANT == 1) {  // no masking at all
        o[j] = atomicAdd(reinterpret_cast<uint32_t *>(smem + (w & 0xfffcu)), pr[j] << ((w << 3) & 31u));
      } else if (VARIANT == 3) {  // idle lanes add to their own spare word (select on the address only); fixed up after the return
        const uint32_t addr = w ? (w & 0xfffcu) : 65536u + (threadIdx.x % 64) * 4u;
        o[j] = atomicAdd(reinterpret_cast<uint32_t *>(smem + addr), pr[j] << ((w << 3) & 31u));
      } else {  // idle lanes add 0 to a spare word (select on address and value)
        const uint32_t addr = w ? (w & 0xfffcu) : 65536u;
        const uint32_t val = w ? pr[j] << ((w << 3) & 31u) : 0u;
        o[j] = atomicAdd(reinterpret_cast<uint32_t *>(smem + addr), val);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t old = __builtin_amdgcn_ubfe(o[j], pc[j] << 3, 8u);
      if (VARIANT == 3) {
        old = pc[j] ? old : thr1 + 1u;
        cands += old == 0u ? 1u : 0u;  // per lane: no trip through the scalar unit
      } else {
        cands += (uint32_t)__popcll(__ballot(old == 0u));
      }
      crossings += thr1 - old < pr[j] ? 1u : 0u;
      st[j] = st[j] * 1664525u + 1013904223u;
      pc[j] = ((st[j] >> 9) & 7u) == 0u && VARIANT != 1 ? 0u : st[j] | 0x38000000u;  // an eighth of the lanes idle
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x % 64 == 0) {
    atomicAdd(&out[0], t1 - t0);
    atomicAdd(&out[1], (unsigned long long)cands + crossings);
  }
}

template <int V>
void run(const char *name, int wg, int threads, const uint32_t *dp, unsigned long long *dout, int iters) {
  hipMemset(dout, 0, 16);
  hipLaunchKernelGGL(k<V>, dim3(wg), dim3(threads), 0, 0, dp, dout, iters, 37.5f);
  hipDeviceSynchronize();
  unsigned long long h[2];
  hipMemcpy(h, dout, 16, hipMemcpyDeviceToHost);
  const double waves = (double)wg * threads / 64;
  printf("%-34s wg %4d x %4d threads: %.1f cycles per iteration (4 postings per lane) per wave\n", name, wg, threads, (double)h[0] / waves / iters);
}

int main() {
  std::vector<uint32_t> hp(1024 * 4);
  uint32_t s = 12345;
  for (auto &x : hp) { s = s * 1664525u + 1013904223u; x = s | 0x38000000u; }
  uint32_t *dp; unsigned long long *dout;
  hipMalloc(&dp, hp.size() * 4); hipMalloc(&dout, 16);
  hipMemcpy(dp, hp.data(), hp.size() * 4, hipMemcpyHostToDevice);
  const int iters = 20000;
  for (int threads : {64, 512}) {
    for (int wg : {1, 256, 512}) {
      run<0>("exec-masked idle lanes", wg, threads, dp, dout, iters);
      run<1>("no idle lanes", wg, threads, dp, dout, iters);
      run<2>("idle lanes -> spare word (selects)", wg, threads, dp, dout, iters);
      run<3>("idle lanes -> own spare word, VALU counts", wg, threads, dp, dout, iters);
    }
  }
  return 0;
}
