// head_gemm16.hpp -- the dense-head contraction on v_mfma_f32_16x16x32_bf16: a measured-and-rejected variant of
// k_head_gemm (all-pairs-similarity_amd/csrc/apss_head.hpp), kept with the microbenchmark that measured it
// (profiles/r03_head_gemm.md): equal at KH = 256 (0.65 vs 0.665 of the bf16 peak), slower at KH = 128 / 64 (0.51 / 0.32 vs
// 0.58 / 0.46), and its diagnostic instantiation (CLK) is what measured the in-kernel clock: 2.39 GHz -- the chip does not
// lower its clock under this kernel, so the MFMA shape has no DVFS give-back to win.  Not part of the library.
#pragma once
#include "../../all-pairs-similarity_amd/csrc/apss_head.hpp"

namespace apss {

// ---------------------------------------------------------------------------------------------------------
// k_head_gemm16: the same contraction on v_mfma_f32_16x16x32_bf16.  Same workgroup shape, same tile stream, same LDS image,
// same output tile per wave (64 query slots x 64 candidates per tile) and the same cycles per flop as k_head_gemm -- what
// differs is the clock the chip holds under the load (MI355X_MICROARCH.md, DVFS give-back item 7: the 16x16x32 loop delivers
// 1.12-1.15x the FLOP/s of the 32x32x16 loop on random data at equal cycles).  Fragments: lane (r16 = lane % 16, kg = lane /
// 16) holds, of A, row 16 m + r16 of the wave's slots, k = 32 kk + 8 kg .. + 8 (chunk 4 kk + kg of its row) -- 16 fragments
// of 8 bf16 per 16-row block, 4 blocks, 128 VGPRs at KH = 256 as before -- and reads, of B, candidate row 16 n + r16 of the
// tile, the same chunk: `ds_read_b128` at lane_base + kk * 4096 + n * 256, a 16-lane group reading 256 contiguous bytes.
// D block (m, n): lane holds candidate column 16 n + r16, slots 16 m + 4 kg + i, i = 0..3.  The half-tile software
// pipeline is the same: the two 32-candidate halves (n = 0, 1 | n = 2, 3) alternate between two accumulator sets of
// 4 x 2 blocks; while the matrix cores work on one half the vector unit scans the other.
typedef __attribute__((ext_vector_type(4))) float apss_f32x4;

template <int KH, bool COUNT = true, bool CLK = false>
__global__ __launch_bounds__(512, 2) void k_head_gemm16(const HeadGemmArgs a) {
  constexpr int KS = KH / 32;                 // k-steps of the 16x16x32 MFMA
  constexpr int SPK = 32 / KS;                // epilogue scan steps (one accumulator element each) per k-step of the other half
  static_assert(SPK * KS == 32, "KH is 64, 128 or 256");
  constexpr int ROWB = KH * 2;
  constexpr int CPR = KH / 8;
  constexpr int TILEB = kHeadCTile * ROWB;    // bytes per tile of this block (contiguous in HBM inside the row's kt-wide tile)
  constexpr int PIECES = TILEB / 1024;
  constexpr int PPW = PIECES / 8;
  static_assert(PIECES % 8 == 0, "every wave copies the same number of pieces");
  __shared__ __attribute__((aligned(1024))) unsigned char ldsb[2 * TILEB];
  __shared__ float scratch[8 * 16 * kWave];   // reporting path only

  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid / kWave), ln = tid % kWave;
  const int r16 = ln & 15, kg = ln >> 4;
  const int panel = blockIdx.x % a.n_panels;
  const int qb = a.n_qblocks - 1 - (int)(blockIdx.x / a.n_panels);
  const bool stored = a.q_slot_base >= 0;
  const int64_t B0 = a.qblock0 + (int64_t)qb * kHeadQBlock;
  const int64_t qs0 = stored ? a.q_slot_base : 0;
  const int n_parts = max(a.n_parts, 1);
  const int t_lo = a.part + n_parts * panel, t_step = n_parts * a.n_panels;
  int t_hi = a.n_ctiles;
  if (stored) t_hi = min(t_hi, (int)((B0 + kHeadQBlock) / kHeadCTile));
  if (t_lo >= t_hi) return;

  // ---- A fragments ----
  apss_bf16x8 af[4][KS];
  const int64_t wslot0 = B0 + 64 * wv;
  bool wave_live = false;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int64_t s = wslot0 + 16 * m + r16;
    const int64_t row = s - qs0;
    const bool ok = row >= 0 && row < a.nq && s < a.wq_rows;
    wave_live |= ok;
    const uint4 *src = reinterpret_cast<const uint4 *>(a.Wq) + ((ok ? wslot0 : 0) / kHeadCTile) * (int64_t)(a.kt / 8 * kHeadCTile) +
                       (int64_t)a.chunk0 * kHeadCTile + 16 * m + r16;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (ok) v = src[(4 * kk + kg) * kHeadCTile];
      af[m][kk] = __builtin_bit_cast(apss_bf16x8, v);
    }
  }
  wave_live = __any(wave_live);

  auto copy_tile = [&](const int t, const int buf) {
    const unsigned char *tsrc = reinterpret_cast<const unsigned char *>(a.Wc) + (int64_t)t * ((int64_t)kHeadCTile * a.kt * 2) +
                                (int64_t)a.chunk0 * (kHeadCTile * 16) + ln * 16;
#pragma unroll
    for (int p = 0; p < PPW; ++p) {
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void *)(tsrc + (wv * PPW + p) * 1024),
          (__attribute__((address_space(3))) void *)(ldsb + buf * TILEB + (wv * PPW + p) * 1024), 16, 0, 0);
    }
  };
  const uint32_t rd_lane = (uint32_t)(kg * 1024 + r16 * 16);

  unsigned long long n_pos = 0;
  float *const sc = scratch + wv * (16 * kWave);

  struct Half {
    float mx;
    uint32_t pos[2];  // positive elements per candidate column of the lane (n = 0, 1 of the half)
  };
  // element e = 0..31 of a half's accumulators: block m = e >> 3, column block n = (e >> 2) & 1, row i = e & 3
  auto scan = [&](Half &hf, const apss_f32x4 (&ac)[4][2], const int e) {
    const float v = ac[e >> 3][(e >> 2) & 1][e & 3];
    hf.mx = fmaxf(hf.mx, v);
    if (COUNT) hf.pos[(e >> 2) & 1] += v > 0.f ? 1u : 0u;
  };
  auto finish = [&](Half &hf, const apss_f32x4 (&ac)[4][2], const int64_t cb_row0) {
    const bool below = stored && cb_row0 + 32 <= B0;
    const int64_t c0 = cb_row0 + r16, c1 = c0 + 16;
    if (COUNT) n_pos += ((below && c0 >= qs0) ? 2u * hf.pos[0] : hf.pos[0]) + ((below && c1 >= qs0) ? 2u * hf.pos[1] : hf.pos[1]);
    if (__any(hf.mx >= a.thr)) {
      const int64_t cext0 = c0 < a.n_rows ? a.c_ext[c0] : 0, cext1 = c1 < a.n_rows ? a.c_ext[c1] : 0;
#pragma unroll
      for (int mp = 0; mp < 2; ++mp) {  // 16 of the 32 elements at a time through the wave-private scratch: the reporting loop is a real loop
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[e * kWave + ln] = ac[2 * mp + (e >> 3)][(e >> 2) & 1][e & 3];
#pragma unroll 1
        for (int e = 0; e < 16; ++e) {
          const float v = sc[e * kWave + ln];
          if (!__any(v >= a.thr)) continue;
          const int nn = (e >> 2) & 1;
          const int64_t c = nn ? c1 : c0;
          const int64_t s = wslot0 + 16 * (2 * mp + (e >> 3)) + 4 * kg + (e & 3);
          const int64_t qrow = s - qs0;
          bool ok = v >= a.thr && qrow >= 0 && qrow < a.nq && c < a.n_rows;
          if (ok) ok = a.q_ext[qrow] != (nn ? cext1 : cext0);  // self-exclusion by external id (IWA:91)
          const uint64_t o = wave_append(ok, &a.counters[kCtrResults]);
          if (ok && o < a.res_cap) {
            a.res_q[o] = (int32_t)qrow;
            a.res_c[o] = (int32_t)c;
            a.res_s[o] = v;
          }
          const bool ok2 = ok && below && c >= qs0;  // the mirrored pair
          const uint64_t o2 = wave_append(ok2, &a.counters[kCtrResults]);
          if (ok2 && o2 < a.res_cap) {
            a.res_q[o2] = (int32_t)(c - qs0);
            a.res_c[o2] = (int32_t)s;
            a.res_s[o2] = v;
          }
        }
      }
    }
    hf.mx = 0.f;
    hf.pos[0] = hf.pos[1] = 0;
  };
  auto ldfrag = [&](const unsigned char *tb, const int n, const int kk) {
    return __builtin_bit_cast(apss_bf16x8, *reinterpret_cast<const uint4 *>(tb + rd_lane + kk * 4096 + n * 256));
  };
  // the MFMAs of one half (column blocks n0, n0 + 1): 8 per k-step; the two B fragments of k-step kk + 1 are requested
  // before the MFMAs of k-step kk; `between(kk)` runs after them
  auto mma_half = [&](apss_f32x4 (&ac)[4][2], const unsigned char *tb, const int n0, auto &&between) {
    apss_bf16x8 b0 = ldfrag(tb, n0, 0), b1 = ldfrag(tb, n0 + 1, 0);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const apss_bf16x8 c0 = b0, c1 = b1;
      if (kk + 1 < KS) {
        b0 = ldfrag(tb, n0, kk + 1);
        b1 = ldfrag(tb, n0 + 1, kk + 1);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (kk == 0) {
          const apss_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
          ac[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][kk], c0, zero, 0, 0, 0);
          ac[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][kk], c1, zero, 0, 0, 0);
        } else {
          ac[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][kk], c0, ac[m][0], 0, 0, 0);
          ac[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][kk], c1, ac[m][1], 0, 0, 0);
        }
      }
      between(kk);
    }
  };

  apss_f32x4 acc0[4][2], acc1[4][2];
  Half h0{0.f, {0u, 0u}}, h1{0.f, {0u, 0u}};
  int64_t pend = -1;
  copy_tile(t_lo, 0);
  int buf = 0;
  // diagnostic build (profiles/microbench/head_gemm_bench.hip, CLK=1): the clock the chip holds inside this loop =
  // shader cycles (s_memtime) per 100-MHz tick (s_memrealtime); no stamp executes in the product instantiations
  unsigned long long clk_c0 = 0, clk_r0 = 0;
  if (CLK) {
    clk_c0 = __builtin_amdgcn_s_memtime();
    clk_r0 = __builtin_amdgcn_s_memrealtime();
  }
  for (int t = t_lo; t < t_hi; t += t_step, buf ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + t_step < t_hi) copy_tile(t + t_step, buf ^ 1);
    if (!wave_live) continue;
    const unsigned char *tb = ldsb + buf * TILEB;
    const int64_t t_row0 = (int64_t)t * kHeadCTile;
    if (pend >= 0) {
      mma_half(acc0, tb, 0, [&](const int kk) {
#pragma unroll
        for (int j = 0; j < SPK; ++j) scan(h1, acc1, kk * SPK + j);
      });
      finish(h1, acc1, pend);
    } else {
      mma_half(acc0, tb, 0, [&](const int) {});
    }
    mma_half(acc1, tb, 2, [&](const int kk) {
#pragma unroll
      for (int j = 0; j < SPK; ++j) scan(h0, acc0, kk * SPK + j);
    });
    finish(h0, acc0, t_row0);
    pend = t_row0 + 32;
  }
  if (wave_live && pend >= 0) {
#pragma unroll
    for (int e = 0; e < 32; ++e) scan(h1, acc1, e);
    finish(h1, acc1, pend);
  }
  if (CLK && tid == 0 && a.clk) {
    a.clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk_c0;
    a.clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
  }
  unsigned long long tot = n_pos;
  for (int o = kWave / 2; o; o >>= 1) tot += __shfl_xor(tot, o);
  if (ln == 0 && tot) atomicAdd(a.head_pairs, tot);
}

}  // namespace apss
