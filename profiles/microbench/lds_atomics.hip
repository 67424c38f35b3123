// Microbenchmark: throughput of LDS read-modify-write primitives on gfx950 with the probe kernel's access shape
// (1024-thread workgroup, one per CU, 32768-word LDS array, per-lane random addresses).  Evidence for DESIGN.md
// "accumulator primitive"; not part of the product.
//   hipcc -O3 --offload-arch=gfx950 -o lds_atomics lds_atomics.hip && ./lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int CB = 32768, BLOCK = 1024, K = 8, ITERS = 2000;

__device__ __forceinline__ uint32_t hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

template <int OP, int PATTERN>
__global__ __launch_bounds__(BLOCK) void k(float *out) {
  extern __shared__ float acc[];
  for (int i = threadIdx.x; i < CB; i += BLOCK) acc[i] = 0.f;
  __syncthreads();
  uint32_t a[K];
  for (int j = 0; j < K; ++j) {
    uint32_t h = hash(threadIdx.x * 131u + j * 7919u + blockIdx.x * 104729u);
    if (PATTERN == 0) a[j] = h % CB;                                      // random
    else if (PATTERN == 1) a[j] = ((h % (CB / 64)) * 64 + (threadIdx.x % 64)); // conflict-free: lane i -> bank i
    else a[j] = (h % (CB / 1024)) * 1024 + threadIdx.x;                   // fully linear
  }
  float s = 0.f;
  uint32_t *iacc = reinterpret_cast<uint32_t *>(acc);
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if (OP == 0) s += atomicAdd(&acc[a[j]], 0.001f);                 // ds_add_rtn_f32
      else if (OP == 1) atomicAdd(&acc[a[j]], 0.001f);                 // ds_add_f32
      else if (OP == 2) s += (float)atomicAdd(&iacc[a[j]], 3u);        // ds_add_rtn_u32
      else if (OP == 3) atomicAdd(&iacc[a[j]], 3u);                    // ds_add_u32
      else if (OP == 4) s += __uint_as_float(atomicExch(&iacc[a[j]], 0u)); // ds_wrxchg_rtn_b32
      else if (OP == 5) acc[a[j]] = s;                                 // ds_write_b32
      else if (OP == 6) s += acc[a[j]];                                // ds_read_b32
      else if (OP == 7) s += __uint_as_float(atomicMax(&iacc[a[j]], (uint32_t)it)); // ds_max_rtn_u32
    }
    if (OP == 5) s += 1.0f;
  }
  if (s == 12345.678f) out[0] = s;
  __syncthreads();
  if (threadIdx.x == 0 && acc[5] == 999.f) out[1] = acc[5];
}

template <int OP, int PATTERN>
void run(const char *name, float *d_out, double clk_ghz) {
  auto kern = k<OP, PATTERN>;
  hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, CB * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<256, BLOCK, CB * 4>>>(d_out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<256, BLOCK, CB * 4>>>(d_out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double wave_instrs_per_cu = (double)ITERS * K * (BLOCK / 64);
  const double cyc = ms * 1e-3 * clk_ghz * 1e9 / wave_instrs_per_cu;
  printf("%-34s %8.3f ms  %7.1f cycles/wave-instr/CU  %6.2f lane-ops/clk/CU  (%.2e lane-ops/s chip)\n", name, ms, cyc, 64.0 / cyc,
         256.0 * wave_instrs_per_cu * 64 / (ms * 1e-3));
}

int main() {
  float *d; hipMalloc(&d, 64);
  int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  const double ghz = clk_khz / 1e6;
  printf("clock %.2f GHz (nominal); cycles quoted at nominal clock\n", ghz);
#define RUN3(OP, NAME) run<OP, 0>(NAME " random", d, ghz); run<OP, 1>(NAME " bank-per-lane", d, ghz); run<OP, 2>(NAME " linear", d, ghz);
  RUN3(0, "ds_add_rtn_f32")
  RUN3(1, "ds_add_f32")
  RUN3(2, "ds_add_rtn_u32")
  RUN3(3, "ds_add_u32")
  RUN3(4, "ds_wrxchg_rtn_b32")
  RUN3(5, "ds_write_b32")
  RUN3(6, "ds_read_b32")
  RUN3(7, "ds_max_rtn_u32")
  return 0;
}
