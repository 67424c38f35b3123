// Are returning global atomics cheaper when they stay in the issuing XCD's L2 (workgroup scope) than at device scope
// (memory side)?  And can a workgroup read which XCD it runs on (HW_REG_XCC_ID)?   hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int SCOPE>
__global__ void k_atomics(uint32_t *table, uint32_t mask, uint32_t *sink, int per_thread) {
  uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u, acc = 0;
  for (int i = 0; i < per_thread; ++i) {
    x = x * 1664525u + 1013904223u;
    acc += __hip_atomic_fetch_add(&table[(x >> 8) & mask], 1u, __ATOMIC_RELAXED, SCOPE);
  }
  if (acc == 0xffffffffu) sink[0] = acc;
}

__global__ void k_xcc(uint32_t *out) {
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);  // XCC_ID[3:0]
}

int main() {
  const uint32_t entries = 1u << 18;  // 1 MB of cursors, like one tile's tile_seg
  uint32_t *table, *sink, *xcc;
  hipMalloc(&table, entries * 4);
  hipMalloc(&sink, 4);
  hipMalloc(&xcc, 4096 * 4);
  hipMemset(table, 0, entries * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = 4096, threads = 256, per = 96;  // 1.0e8 atomics
  for (int scope = 0; scope < 2; ++scope) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (scope == 0) hipLaunchKernelGGL(k_atomics<__HIP_MEMORY_SCOPE_AGENT>, dim3(blocks), dim3(threads), 0, 0, table, entries - 1, sink, per);
      else hipLaunchKernelGGL(k_atomics<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(blocks), dim3(threads), 0, 0, table, entries - 1, sink, per);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%s scope: %.2f ms for %.2e returning atomics = %.2e /s\n", scope ? "workgroup" : "agent", ms,
             (double)blocks * threads * per, (double)blocks * threads * per / (ms * 1e-3));
    }
  }
  hipLaunchKernelGGL(k_xcc, dim3(4096), dim3(64), 0, 0, xcc);
  std::vector<uint32_t> h(4096);
  hipMemcpy(h.data(), xcc, 4096 * 4, hipMemcpyDeviceToHost);
  printf("xcc id of blocks 0..23:");
  for (int i = 0; i < 24; ++i) printf(" %u", h[i]);
  int hist[16] = {0};
  for (int i = 0; i < 4096; ++i) hist[h[i] & 15]++;
  printf("\nblocks per xcc:");
  for (int i = 0; i < 16; ++i) printf(" %d", hist[i]);
  int rr = 0;
  for (int i = 0; i < 4096; ++i) rr += (h[i] & 15) == (uint32_t)(i % 8);
  printf("\nblocks with xcc == blockIdx %% 8: %d of 4096\n", rr);
  return 0;
}
