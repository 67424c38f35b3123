// Calibration of rocprofv3 FETCH_SIZE on gfx950 for the probe kernel's access shape (8 B per lane, 8-lane groups
// reading 64-B chunks at 8-B-aligned pseudo-random offsets) against a linear 8 B/lane stream and a 16 B/lane
// stream, each over a 2 GiB buffer read exactly once (far beyond the 256 MiB Infinity Cache).
//   hipcc -O3 --offload-arch=gfx950 -o fetch_calib fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr size_t BYTES = 2ull << 30;
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__global__ void k_linear8(const uint2 *p, size_t n, unsigned *out) {
  unsigned s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint2 v = p[i]; s += v.x ^ v.y; }
  if (s == 0x12345u) out[0] = s;
}
__global__ void k_linear16(const uint4 *p, size_t n, unsigned *out) {
  unsigned s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; s += v.x ^ v.y ^ v.z ^ v.w; }
  if (s == 0x12345u) out[0] = s;
}
// chunk c (64 B worth of postings) sits at 8-B-aligned offset: a pseudo-random position + random 8-B phase
__global__ void k_chunks(const uint2 *p, size_t n_chunks, unsigned *out) {
  unsigned s = 0;
  const size_t groups = (size_t)gridDim.x * blockDim.x / 8;
  for (size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 8; g < n_chunks; g += groups) {
    const uint32_t h = hash((uint32_t)g * 2654435761u + 12345u);
    const size_t chunk = (size_t)h % (n_chunks - 1);
    const size_t phase = (hash(h) % 8);
    uint2 v = p[chunk * 8 + phase + (threadIdx.x % 8)];
    s += v.x ^ v.y;
  }
  if (s == 0x12345u) out[0] = s;
}
int main() {
  uint2 *d; unsigned *o;
  (void)hipMalloc(&d, BYTES + 4096); (void)hipMalloc(&o, 64); (void)hipMemset(d, 1, BYTES + 4096);
  (void)hipDeviceSynchronize();
  k_linear8<<<4096, 256>>>(d, BYTES / 8, o);
  k_linear16<<<4096, 256>>>((const uint4 *)d, BYTES / 16, o);
  k_chunks<<<4096, 256>>>(d, BYTES / 64, o);
  (void)hipDeviceSynchronize();
  printf("each kernel requested %zu bytes\n", BYTES);
  return 0;
}
