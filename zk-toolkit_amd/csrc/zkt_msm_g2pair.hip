// G2 bucket accumulation with two lanes per task (the largest single item of a Groth16 proof).
//
// An XYZZ accumulator over Fq2 is 8 Fq = 112 registers and the mixed add needs ten more Fq temporaries: with one lane per task the
// kernel needs 256 VGPRs + 234 AGPRs.  Here the c0 halves of every Fq2 value live on the even lane of a pair and the c1 halves on the
// odd lane (tower.h, ZKT_FQ2_SPLIT): additions are lane-local, a product exchanges the operands by DPP and is one two-product multiply
// per lane.  The total multiply count is unchanged; what is gained is the working set (255 VGPRs + 56 AGPRs, no scratch) and twice
// the waves to schedule.  Measured on MI355X, 2^20-term G2 MSM over resident bases: accumulate 10.97 -> 9.2 ms, whole MSM
// 11.8 -> 9.95 ms.  Forcing two waves per SIMD (ZKT_G2PAIR_ATTR = amdgpu_waves_per_eu(2,2)) spills 560 B per lane, is not faster
// (9.4 ms) and starves the reduce-stage kernels of registers, so it is not the default.
// Memory layouts (window-multiple table, bucket sums, partials) are exactly those of the one-lane kernels in zkt_msm.hip, which
// still do the table build and the reduction.
//
// The split layout redefines zkt::Fq2, so this translation unit keeps its symbols in a namespace of its own.
#define ZKT_FQ2_SPLIT
#define zkt zkt_g2pair
#include "curve.h"
#include "zkt_internal.h"

namespace zkt {

__device__ inline Fq2 ld_half(const uint32_t* p) { Fq2 r; const uint32_t* q = p + (fq2_odd() ? FqC::N : 0);
#pragma unroll
  for (int i = 0; i < FqC::N; ++i) r.h.v[i] = q[i];
  return r; }
__device__ inline void st_half(uint32_t* p, const Fq2& a) { uint32_t* q = p + (fq2_odd() ? FqC::N : 0);
#pragma unroll
  for (int i = 0; i < FqC::N; ++i) q[i] = a.h.v[i]; }

// same task list, same entry order, same memory layout as k_accumulate<Fq2Ops> (zkt_msm.hip); lanes 2t and 2t+1 share task t
#ifndef ZKT_G2PAIR_ATTR
#define ZKT_G2PAIR_ATTR
#endif
__global__ void __launch_bounds__(64) ZKT_G2PAIR_ATTR
k_accumulate_g2_pair(const uint32_t* __restrict__ table, const uint32_t* __restrict__ entries, const uint32_t* __restrict__ offsets,
                     const uint2* __restrict__ order, const uint32_t* __restrict__ task_off, size_t nbuckets,
                     uint32_t* __restrict__ sums, uint32_t* __restrict__ partial) {
  typedef Fq2Ops F;
  constexpr int CW = 2 * FqC::N, XYW = 4 * CW;                      // words of one Fq2 coordinate / one XYZZ point in memory
  const size_t t = ((size_t)blockIdx.x * 64 + threadIdx.x) >> 1;
  if (t >= task_off[nbuckets]) return;
  const uint2 tk = order[t];
  const size_t b = tk.x;
  const uint32_t t0 = task_off[b], nt = task_off[b + 1] - t0;
  const uint32_t off = offsets[b], cnt = offsets[b + 1] - off;
  uint32_t beg = off, end = off + cnt;
  if (nt != 1) {                                   // piece tk.y of nt equal pieces, exactly as k_accumulate (zkt_msm.hip)
    const uint32_t q = cnt / nt, r = cnt - q * nt;
    beg = off + tk.y * q + (tk.y < r ? tk.y : r); end = beg + q + (tk.y < r ? 1u : 0u);
  }
  Xyzz<F> acc = xyzz_inf<F>();
  uint32_t ent = beg < end ? entries[beg] : 0, ent_next = beg + 1 < end ? entries[beg + 1] : 0;
  const uint32_t* p = table + (size_t)(ent & 0x7fffffffu) * (2 * CW);
#if !defined(ZKT_G2PAIR_NO_PREFETCH)
  Fq2 nx = ld_half(p), ny = ld_half(p + CW);                        // software-pipelined gather, as in the G1 kernel
  for (uint32_t e = beg; e < end; ++e) {
    Fq2 x = nx, y = ny;
    const bool negate = ent >> 31;
    if (e + 1 < end) {
      ent = ent_next;
      ent_next = e + 2 < end ? entries[e + 2] : 0;
      p = table + (size_t)(ent & 0x7fffffffu) * (2 * CW);
      nx = ld_half(p); ny = ld_half(p + CW);
    }
    if (negate) y = F::neg(y);
    acc = xyzz_add_aff<F>(acc, x, y);
  }
#else         // -DZKT_G2PAIR_NO_PREFETCH: no point held in registers across the add.  Measured (2^20-term G2 MSM, pipelined): 9.58 ms default, 9.87 ms this way at one wave
              // per SIMD, 9.51 ms with amdgpu_waves_per_eu(2,2) on top (118 dwords spilled) — no gain, not the default
  for (uint32_t e = beg; e < end; ++e) {
    Fq2 x = ld_half(p), y = ld_half(p + CW);
    const bool negate = ent >> 31;
    ent = ent_next;
    ent_next = e + 2 < end ? entries[e + 2] : 0;
    p = table + (size_t)(ent & 0x7fffffffu) * (2 * CW);
    if (negate) y = F::neg(y);
    acc = xyzz_add_aff<F>(acc, x, y);
  }
#endif
  uint32_t* out = nt == 1 ? sums + b * XYW : partial + (size_t)(t0 + tk.y) * XYW;
  st_half(out, acc.X); st_half(out + CW, acc.Y); st_half(out + 2 * CW, acc.ZZ); st_half(out + 3 * CW, acc.ZZZ);
}

// The same accumulation over the LAST LAYER of the affine rounds (zkt_msm_affine.hip): bucket b holds cnt[b] affine points side by side at off[b] — no entry list, no
// signs, an infinity byte per slot (P + (-P) met in a round).  Same task list semantics: piece tk.y of nt equal pieces.
__global__ void __launch_bounds__(64) ZKT_G2PAIR_ATTR
k_accumulate_g2_pair_direct(const uint32_t* __restrict__ pts, const uint8_t* __restrict__ inf, const uint32_t* __restrict__ offs, const uint32_t* __restrict__ cnts,
                            const uint2* __restrict__ order, const uint32_t* __restrict__ task_off, size_t nbuckets,
                            uint32_t* __restrict__ sums, uint32_t* __restrict__ partial) {
  typedef Fq2Ops F;
  constexpr int CW = 2 * FqC::N, XYW = 4 * CW;
  const size_t t = ((size_t)blockIdx.x * 64 + threadIdx.x) >> 1;
  if (t >= task_off[nbuckets]) return;
  const uint2 tk = order[t];
  const size_t b = tk.x;
  const uint32_t t0 = task_off[b], nt = task_off[b + 1] - t0;
  const uint32_t off = offs[b], cnt = cnts[b];
  uint32_t beg = off, end = off + cnt;
  if (nt != 1) {
    const uint32_t q = cnt / nt, r = cnt - q * nt;
    beg = off + tk.y * q + (tk.y < r ? tk.y : r); end = beg + q + (tk.y < r ? 1u : 0u);
  }
  Xyzz<F> acc = xyzz_inf<F>();
  for (uint32_t e = beg; e < end; ++e) {
    if (inf[e]) continue;                                            // pair-uniform: both lanes of a task read the same byte
    const uint32_t* p = pts + (size_t)e * (2 * CW);
    const Fq2 x = ld_half(p), y = ld_half(p + CW);
    acc = xyzz_add_aff<F>(acc, x, y);
  }
  uint32_t* out = nt == 1 ? sums + b * XYW : partial + (size_t)(t0 + tk.y) * XYW;
  st_half(out, acc.X); st_half(out + CW, acc.Y); st_half(out + 2 * CW, acc.ZZ); st_half(out + 3 * CW, acc.ZZZ);
}

}  // namespace zkt
#undef zkt

hipError_t zkt_launch_accumulate_g2_pair(const uint32_t* table, const uint32_t* entries, const uint32_t* offsets, const void* order, const uint32_t* task_off,
                                         size_t nbuckets, uint32_t* sums, uint32_t* partial, size_t max_tasks, hipStream_t s) {
  if (max_tasks == 0) return hipSuccess;
  hipLaunchKernelGGL(zkt_g2pair::k_accumulate_g2_pair, dim3((unsigned)((2 * max_tasks + 63) / 64)), dim3(64), 0, s, table, entries, offsets, (const uint2*)order, task_off,
                     nbuckets, sums, partial);
  return hipGetLastError();
}
hipError_t zkt_launch_accumulate_g2_pair_direct(const uint32_t* pts, const uint8_t* inf, const uint32_t* off, const uint32_t* cnt, const void* order, const uint32_t* task_off,
                                                size_t nbuckets, uint32_t* sums, uint32_t* partial, size_t max_tasks, hipStream_t s) {
  if (max_tasks == 0) return hipSuccess;
  hipLaunchKernelGGL(zkt_g2pair::k_accumulate_g2_pair_direct, dim3((unsigned)((2 * max_tasks + 63) / 64)), dim3(64), 0, s, pts, inf, off, cnt, (const uint2*)order, task_off,
                     nbuckets, sums, partial);
  return hipGetLastError();
}
