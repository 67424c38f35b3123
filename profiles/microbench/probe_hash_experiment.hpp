// EXPERIMENT, NOT BUILT: a filter kernel that takes one QUERY at a time against all tiles with its sums in an LDS hash table
// (instead of one round per (query, tile) over dense accumulators).  Wired into apss_hip.hip's probe() for the measurement
// (two ProbeArgs fields hash_tiles_per_pass / hash_slot_shift, an APSS_DEBUG token), parity-green on 149 GPU tests with the
// path forced -- and slower wherever it was tried (profiles/r03_probe_hash_experiment.log, one MI355X, filter kernel ms):
//   one shard of 8 of uniform C3           86.5 vs 23.6 (k_probe_even)
//   one shard of 8 of power-law C5 (N=2M)  91.7 vs 74.4 (8 tiles per pass; 133.8 at 4; 31 = one pass overflows the table)
//   C3 with Zipf(1) terms, whole step      195.6 vs 187.3
// A posting costs a CAS probe loop (serially dependent returning LDS atomics, the wave waits for its slowest lane) plus the
// add, against ONE non-returning-latency-hidden add in the tile kernels; the rounds it saves are cheaper than that.  The
// power-law shard was also less thin than assumed: 4,700 postings per query (15 chunks per round), not a few hundred.
// Removed from the library again; kept here as the record.
// k_probe_hash: the filter of the VERY THIN regime -- a query meets a handful of postings per tile (the tail of a term
// shard once the dense-head block has taken the frequent terms: ~7 terms per row and shard, ~10 postings per term and tile).
//
// The tile kernels (apss_even.hpp, k_probe_coarse) spend one ROUND per (query, tile): two workgroup barriers, a staged strip
// and a share of the accumulator clear, ~1800 cycles whatever the round holds.  At 70 postings a round that floor is all of
// the cost: power-law C5's 8-way shard, 31 tiles x 2M queries = 6.2e7 rounds, 75 ms for 1.4e8 posting visits.
//
// Here a workgroup takes one QUERY at a time against ALL tiles of the launch: the sums live in an open-addressing hash table
// in LDS keyed by the candidate's global slot (C entries of {key, 32-bit sum}), so the cost follows the postings the query
// meets, not the candidates it could meet.  Same coarse postings, same products (floor(w_q w_c S + 1)), same threshold and
// the same crossing rule as the tile kernels: the add that takes a candidate's sum across the threshold reports it (sums are
// monotone: non-negative weights only), every reported pair is re-scored exactly by k_rescore / the shard's phase 2.
//
// A query whose candidates would crowd the table is cut into PASSES over consecutive tile groups (the host sizes them from
// what the index build measured; the table is wiped between passes).  A pass that still fills the table raises
// kFlagHashFull: the host then runs the call again on the tile kernels and keeps this handle away from here.
//
// Reference semantics: the accumulation of IndexingWorkerActor.querySimilarItems (IndexingWorkerActor.scala:79-101) over the
// candidates of the query's posting lists, restricted to what can reach the threshold.
#pragma once

namespace apss {

constexpr unsigned long long kFlagHashFull = 2ull;  // counters[kCtrFlags]: a pass filled k_probe_hash's table

template <int BLOCK, int LOGC, bool SHARD>
__global__ __launch_bounds__(BLOCK) void k_probe_hash(const ProbeArgs a) {
  constexpr int C = 1 << LOGC;
  constexpr int ICAP = 1024;        // chunks of 16 postings queued per flush
  constexpr int NG = BLOCK / 16;    // 16-lane groups: one chunk each per step
  constexpr int UN = 4;             // chunks in flight per group
  constexpr int kFull = C - C / 4;  // keys per pass beyond which probing degenerates: reported, the handle leaves this kernel
  constexpr int kMaxProbe = 512;    // slots tried before a posting is given up (and the call run again elsewhere)
  __shared__ __attribute__((aligned(16))) uint32_t keys[C];  // candidate slot + 1; 0: empty
  __shared__ __attribute__((aligned(16))) uint32_t vals[C];  // its sum, coarse units
  __shared__ uint32_t it_chunk[ICAP];  // index of the chunk's first posting / 16 (segments are aligned to 32)
  __shared__ uint32_t it_w[ICAP];      // query weight x S (/ the query's shard ratio), float bits
  __shared__ uint32_t it_row0[ICAP];   // first candidate slot of the chunk's tile
  __shared__ uint32_t ctr[4];          // [0] chunks reserved for this flush, [1] keys of this pass
  __shared__ unsigned long long stat[2];

  const int tid = threadIdx.x;
  const int ln = tid % kWave;
  const int grp = tid / 16, gl = tid % 16;
  const int v0 = blockIdx.x * a.q_chunk;
  const int v1 = min(a.nq, v0 + a.q_chunk);
  const int tile_lo = a.tile0, tile_hi = a.tile0 + a.n_tiles;
  const int tpp = max(a.hash_tiles_per_pass, 1);
  const uint32_t shift = (uint32_t)a.hash_slot_shift;
  const float cxs = a.cx_scale;
  const int thr_c = (int)a.cx_theta - 2;  // (k_probe_coarse: the soundness argument of the coarse threshold)
  const uint32_t thr1 = (uint32_t)max(thr_c, 1) - 1u;

  for (int i = tid; i < C; i += BLOCK) {
    keys[i] = 0u;
    vals[i] = 0u;
  }
  if (tid < 4) ctr[tid] = 0u;
  if (tid < 2) stat[tid] = 0ull;
  unsigned long long my_visits = 0;
  uint32_t my_new = 0;  // (lane 0 of each wave counts its wave's new keys)
  bool full = false;
  __syncthreads();

  for (int q = v0; q < v1; ++q) {
    const int64_t qb = a.q_rowptr[q];
    const int nnz = (int)(a.q_rowptr[q + 1] - qb);
    const float qs = SHARD ? a.q_scale[q] : 1.0f;
    if (nnz <= 0 || !(qs > 0.f)) continue;  // (uniform: nothing of this query in the shard / the tail)
    const float wmul = SHARD ? cxs / qs : cxs;
    const int64_t qext = a.q_ext[q];
    for (int tp0 = tile_lo; tp0 < tile_hi; tp0 += tpp) {
      const int ntl = min(tpp, tile_hi - tp0);
      const int npairs = nnz * ntl;
      for (int pb = 0; pb < npairs; pb += BLOCK) {
        // one (term, tile) pair per thread: its segment, as chunks
        const int i = pb + tid;
        uint32_t first = 0, rem = 0, wbits = 0, row0 = 0;
        if (i < npairs) {
          const int tl = i / nnz, k = i - tl * nnz;
          const int tile = tp0 + tl;
          const uint32_t term = (uint32_t)a.q_idx[qb + k];
          const uint2 sg = a.tile_seg[(int64_t)tile * a.seg_stride + term];
          first = (uint32_t)((a.tile_post_base[tile] + (int64_t)sg.x) >> 4);
          rem = (sg.y + 15u) >> 4;
          wbits = __float_as_uint(a.q_val[qb + k] * wmul);
          row0 = (uint32_t)tile * (uint32_t)a.cb;
          my_visits += sg.y;
        }
        for (;;) {
          // queue as many of the pending chunks as the list takes (one LDS atomic per wave)
          const uint32_t incl = wave_incl_scan(rem);
          uint32_t base = 0;
          if (ln == kWave - 1) base = atomicAdd(&ctr[0], incl);
          const uint32_t j0 = (uint32_t)__builtin_amdgcn_readlane((int)base, kWave - 1) + incl - rem;
          const uint32_t fit = j0 < (uint32_t)ICAP ? min(rem, (uint32_t)ICAP - j0) : 0u;
          for (uint32_t c = 0; c < fit; ++c) {
            it_chunk[j0 + c] = first + c;
            it_w[j0 + c] = wbits;
            it_row0[j0 + c] = row0;
          }
          first += fit;
          rem -= fit;
          __syncthreads();
          const int n = (int)min(ctr[0], (uint32_t)ICAP);
          for (int t0 = 0; t0 < n; t0 += NG * UN) {
            uint32_t pc[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
              const int it = t0 + u * NG + grp;
              pc[u] = it < n ? a.post_c[((int64_t)it_chunk[it] << 4) + gl] : 0u;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
              const int it = min(t0 + u * NG + grp, ICAP - 1);
              const uint32_t w = pc[u];
              const bool live = w != 0u;  // (a posting word of zero is padding)
              const uint32_t cand = it_row0[it] + ((w & 0xffffu) >> shift);
              const float x = __builtin_fmaf(__uint_as_float(it_w[it]), __half2float(__ushort_as_half((unsigned short)(w >> 16))), 1.0f);
              const uint32_t p = (uint32_t)x;
              const uint32_t key = cand + 1u;
              uint32_t hslot = (key * 2654435761u) >> (32 - LOGC);
              bool isnew = false, placed = !live;
              for (int guard = 0; !placed && guard < kMaxProbe; ++guard) {
                const uint32_t old = atomicCAS(&keys[hslot], 0u, key);
                if (old == 0u || old == key) {
                  isnew = old == 0u;
                  placed = true;
                } else {
                  hslot = (hslot + 1u) & (uint32_t)(C - 1);
                }
              }
              full |= !placed;
              uint32_t oldv = thr1 + 1u;  // (never a crossing: thr1 - old wraps)
              if (live && placed) oldv = atomicAdd(&vals[hslot], p);
              const unsigned long long nm = __ballot(isnew);
              if (ln == 0 && nm) {
                my_new += (uint32_t)__popcll(nm);
                atomicAdd(&ctr[1], (uint32_t)__popcll(nm));
              }
              bool ok = live && placed && (thr1 - oldv) < p;
              if (ok) ok = a.ext_id[cand] != qext;
              const uint64_t o = wave_append(ok, &a.counters[kCtrResults]);
              if (ok && o < a.res_cap) {
                a.res_q[o] = q;
                a.res_c[o] = (int32_t)cand;
                a.res_s[o] = (float)(oldv + p) / cxs;  // coarse score at the crossing, replaced by the exact pass
              }
            }
          }
          const int more = __syncthreads_or(rem > 0u ? 1 : 0);
          if (tid == 0) ctr[0] = 0u;
          __syncthreads();
          if (!more) break;
        }
      }
      // end of the pass: wipe the table (every thread's adds landed before the last barrier)
      full |= ctr[1] > (uint32_t)kFull;
      for (int i = tid * 4; i < C; i += BLOCK * 4) {
        *reinterpret_cast<uint4 *>(keys + i) = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4 *>(vals + i) = make_uint4(0u, 0u, 0u, 0u);
      }
      __syncthreads();
      if (tid == 0) ctr[1] = 0u;  // (the next pass's first count follows a barrier)
    }
  }
  // statistics: posting visits, first touches (= candidate pairs), table-full flag
  for (int off = 32; off > 0; off >>= 1) my_visits += __shfl_down(my_visits, off);
  if (ln == 0) {
    atomicAdd(&stat[0], my_visits);
    atomicAdd(&stat[1], (unsigned long long)my_new);
  }
  if (__syncthreads_or(full ? 1 : 0) && tid == 0) atomicOr(&a.counters[kCtrFlags], kFlagHashFull);
  __syncthreads();
  if (tid == 0) {
    atomicAdd(&a.counters[kCtrVisits], stat[0]);
    atomicAdd(&a.counters[kCtrCands], stat[1]);
  }
}

}  // namespace apss
