// Go / no-go experiment (round 4, review item 1): ONE round of a batched-affine pair tree against the XYZZ bucket accumulation,
// on the library's own field code (fp.h / curve.h, multiply inlined like the MSM objects).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DZKT_INLINE_MUL -DZKT_WORDSTEP_INV_FQ -I zk-toolkit_amd/csrc -o build/affine_round tools/ubench/affine_round.hip
//   ./build/affine_round [log2 pairs = 22.7 -> 6.8 M]        (profiles/r04_batched_affine_go_no_go.md has the numbers)
// Workload: `npairs` pair additions P1 + P2 of affine points gathered from a table of T points through an index list, as the first round of a pair tree over the
// bucket-sorted entries of a 2^20-term MSM would (T = 13 * 2^20 window multiples, 6.8 M pairs); a lane owns K consecutive pairs and shares ONE inversion
// among them (Montgomery's trick: forward pass = prefix products of the x-differences, word-step inversion, backward pass = slopes and sums).
// Variants: gathered (random indices) / linear (indices in order: what rounds 2.. read); K = 32 / 64 / 128; and the XYZZ mixed-addition chain over the same
// gather (2K entries per lane) as the same-box baseline.  A check kernel recomputes sampled pairs through xyzz_add_aff + one inversion each.
// The chord formula and the XYZZ formula are the same rational functions of the coordinates, so random field elements serve as "points".
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <algorithm>
#include <vector>
#include "abi.h"

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
using namespace zkt;

template <class F> struct CoordIO;
template <> struct CoordIO<FqOps> {
  static constexpr int CW = FqC::N;
  __device__ static Fq ld(const uint32_t* p) { return ld_raw<FqC>(p); }
  __device__ static void st(uint32_t* p, const Fq& a) { st_raw<FqC>(p, a); }
};
template <> struct CoordIO<Fq2Ops> {
  static constexpr int CW = 2 * FqC::N;
  __device__ static Fq2 ld(const uint32_t* p) { Fq2 r; r.c0 = ld_raw<FqC>(p); r.c1 = ld_raw<FqC>(p + FqC::N); return r; }
  __device__ static void st(uint32_t* p, const Fq2& a) { st_raw<FqC>(p, a.c0); st_raw<FqC>(p + FqC::N, a.c1); }
};

// random "points": limbs < 2^28, top limb < 2^16 (value < p)
__global__ void k_fill(uint32_t* t, size_t words, int cw, uint32_t seed) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= words) return;
  uint64_t z = (i + 1) * 0x9e3779b97f4a7c15ull + seed; z ^= z >> 31; z *= 0xbf58476d1ce4e5b9ull; z ^= z >> 29; z *= 0x94d049bb133111ebull; z ^= z >> 32;
  uint32_t v = (uint32_t)z & 0x0fffffffu;
  if ((i % 14) == 13) v &= 0xffffu;
  t[i] = v;
}
__global__ void k_entries(uint32_t* e, size_t n, uint32_t T, int linear, uint32_t seed) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t z = (i + 1) * 0x9e3779b97f4a7c15ull + seed; z ^= z >> 31; z *= 0xbf58476d1ce4e5b9ull; z ^= z >> 29; z *= 0x94d049bb133111ebull; z ^= z >> 32;
  e[i] = linear ? (uint32_t)(i % T) : (uint32_t)(z % T) | ((uint32_t)(z >> 63) << 31);
}

// prefix products: wave-interleaved, element = CW words, quad q of step i of block b at ((b*K + i)*Q + q)*256 + lane*4 words
template <class F> __device__ inline void st_pref(uint32_t* base, const typename F::E& v) {
  constexpr int CW = CoordIO<F>::CW, Q = (CW + 3) / 4;
  uint32_t w[Q * 4] = {};
  CoordIO<F>::st(w, v);
#pragma unroll
  for (int q = 0; q < Q; ++q) *reinterpret_cast<uint4*>(base + q * 256) = uint4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
}
template <class F> __device__ inline typename F::E ld_pref(const uint32_t* base) {
  constexpr int CW = CoordIO<F>::CW, Q = (CW + 3) / 4;
  uint32_t w[Q * 4];
#pragma unroll
  for (int q = 0; q < Q; ++q) { const uint4 t = *reinterpret_cast<const uint4*>(base + q * 256); w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w; }
  return CoordIO<F>::ld(w);
}

// one lane = K pair additions with one inversion.  MODE 0: both passes, 1: forward only (prefix products + their product), 2: backward only
template <class F, int K, int MODE>
__global__ void __launch_bounds__(64) k_aff_round(const uint32_t* __restrict__ table, const uint32_t* __restrict__ entries, size_t npairs,
                                                  uint32_t* __restrict__ pref, uint32_t* __restrict__ total, uint32_t* __restrict__ out) {
  typedef typename F::E E;
  constexpr int CW = CoordIO<F>::CW, PW = 2 * CW, Q = (CW + 3) / 4;
  const size_t lane = (size_t)blockIdx.x * 64 + threadIdx.x;
  const size_t p0 = lane * K;
  if (p0 >= npairs) return;                      // npairs is a multiple of 64 * K
  uint32_t* myp = pref + (size_t)blockIdx.x * K * Q * 256 + threadIdx.x * 4;
  const uint32_t* ent = entries + 2 * p0;
  E acc = F::one();
  if (MODE != 2) {
    uint32_t e1 = ent[0], e2 = ent[1];
    E nx1 = CoordIO<F>::ld(table + (size_t)(e1 & 0x7fffffffu) * PW), nx2 = CoordIO<F>::ld(table + (size_t)(e2 & 0x7fffffffu) * PW);
    for (int i = 0; i < K; ++i) {
      const E x1 = nx1, x2 = nx2;
      if (i + 1 < K) {
        e1 = ent[2 * i + 2]; e2 = ent[2 * i + 3];
        nx1 = CoordIO<F>::ld(table + (size_t)(e1 & 0x7fffffffu) * PW); nx2 = CoordIO<F>::ld(table + (size_t)(e2 & 0x7fffffffu) * PW);
      }
      st_pref<F>(myp + (size_t)i * Q * 256, acc);
      acc = F::mul(acc, F::sub(x2, x1));
    }
    if (MODE == 1) { st_pref<F>(total + (size_t)blockIdx.x * Q * 256 + threadIdx.x * 4, acc); return; }
  } else acc = ld_pref<F>(total + (size_t)blockIdx.x * Q * 256 + threadIdx.x * 4);
  E inv = F::inv(acc);
  for (int i = K - 1; i >= 0; --i) {
    const uint32_t e1 = ent[2 * i], e2 = ent[2 * i + 1];
    const uint32_t* q1 = table + (size_t)(e1 & 0x7fffffffu) * PW; const uint32_t* q2 = table + (size_t)(e2 & 0x7fffffffu) * PW;
    const E x1 = CoordIO<F>::ld(q1), x2 = CoordIO<F>::ld(q2);
    E y1 = CoordIO<F>::ld(q1 + CW), y2 = CoordIO<F>::ld(q2 + CW);
    const E pp = ld_pref<F>(myp + (size_t)i * Q * 256);
    if (e1 >> 31) y1 = F::neg(y1);
    if (e2 >> 31) y2 = F::neg(y2);
    const E d = F::sub(x2, x1);
    const E dinv = F::mul(inv, pp);
    inv = F::mul(inv, d);
    const E lam = F::mul(F::sub(y2, y1), dinv);
    const E x3 = F::sub(F::sub(F::sqr(lam), x1), x2);
    const E y3 = F::sub(F::mul(lam, F::sub(x1, x3)), y1);
    uint32_t* o = out + (p0 + i) * PW;
    CoordIO<F>::st(o, x3); CoordIO<F>::st(o + CW, y3);
  }
}

// baseline: the XYZZ mixed-addition chain of k_accumulate over the same gather, 2K entries per lane (software-pipelined like the product kernel)
template <class F, int K>
__global__ void __launch_bounds__(64) k_xyzz_chain(const uint32_t* __restrict__ table, const uint32_t* __restrict__ entries, size_t npairs, uint32_t* __restrict__ out) {
  typedef typename F::E E;
  constexpr int CW = CoordIO<F>::CW, PW = 2 * CW;
  const size_t lane = (size_t)blockIdx.x * 64 + threadIdx.x;
  const size_t p0 = lane * K;
  if (p0 >= npairs) return;
  const uint32_t* ent = entries + 2 * p0;
  Xyzz<F> acc = xyzz_inf<F>();
  uint32_t e = ent[0];
  const uint32_t* p = table + (size_t)(e & 0x7fffffffu) * PW;
  E nx = CoordIO<F>::ld(p), ny = CoordIO<F>::ld(p + CW);
  for (int i = 0; i < 2 * K; ++i) {
    E x = nx, y = ny; const bool negate = e >> 31;
    if (i + 1 < 2 * K) { e = ent[i + 1]; p = table + (size_t)(e & 0x7fffffffu) * PW; nx = CoordIO<F>::ld(p); ny = CoordIO<F>::ld(p + CW); }
    if (negate) y = F::neg(y);
    acc = xyzz_add_aff<F>(acc, x, y);
  }
  uint32_t* o = out + lane * 4 * CW;
  CoordIO<F>::st(o, acc.X); CoordIO<F>::st(o + CW, acc.Y); CoordIO<F>::st(o + 2 * CW, acc.ZZ); CoordIO<F>::st(o + 3 * CW, acc.ZZZ);
}

// check: pair j through xyzz_add_aff + normalisation, compared with the batched result (canonical words)
template <class F> __device__ inline bool same_elem(const typename F::E& a, const typename F::E& b) { return F::eq(a, b); }
template <class F>
__global__ void __launch_bounds__(64) k_check(const uint32_t* __restrict__ table, const uint32_t* __restrict__ entries, size_t npairs, size_t step, const uint32_t* __restrict__ out, uint32_t* __restrict__ bad) {
  typedef typename F::E E;
  constexpr int CW = CoordIO<F>::CW, PW = 2 * CW;
  const size_t j = ((size_t)blockIdx.x * 64 + threadIdx.x) * step;
  if (j >= npairs) return;
  const uint32_t e1 = entries[2 * j], e2 = entries[2 * j + 1];
  const uint32_t* q1 = table + (size_t)(e1 & 0x7fffffffu) * PW; const uint32_t* q2 = table + (size_t)(e2 & 0x7fffffffu) * PW;
  E x1 = CoordIO<F>::ld(q1), y1 = CoordIO<F>::ld(q1 + CW), x2 = CoordIO<F>::ld(q2), y2 = CoordIO<F>::ld(q2 + CW);
  if (e1 >> 31) y1 = F::neg(y1);
  if (e2 >> 31) y2 = F::neg(y2);
  Xyzz<F> a{x1, y1, F::one(), F::one()};
  const Aff<F> r = xyzz_to_aff<F>(xyzz_add_aff<F>(a, x2, y2));
  const E x3 = CoordIO<F>::ld(out + j * PW), y3 = CoordIO<F>::ld(out + j * PW + CW);
  if (!same_elem<F>(r.x, x3) || !same_elem<F>(r.y, y3)) atomicAdd(bad, 1u);
}

static float time_ms(hipEvent_t e0, hipEvent_t e1) { float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); return ms; }

template <class F, int K> void run(const char* group, size_t npairs_req, uint32_t T, int linear) {
  constexpr int CW = CoordIO<F>::CW, PW = 2 * CW, Q = (CW + 3) / 4;
  const size_t per_block = (size_t)64 * K;
  const size_t nblk = (npairs_req + per_block - 1) / per_block, npairs = nblk * per_block;
  uint32_t *table, *entries, *pref, *total, *out, *xout, *bad;
  CHK(hipMalloc(&table, (size_t)T * PW * 4)); CHK(hipMalloc(&entries, npairs * 2 * 4));
  CHK(hipMalloc(&pref, nblk * K * Q * 256 * 4)); CHK(hipMalloc(&total, nblk * Q * 256 * 4));
  CHK(hipMalloc(&out, npairs * PW * 4)); CHK(hipMalloc(&xout, nblk * 64 * 4 * CW * 4)); CHK(hipMalloc(&bad, 4));
  const size_t tw = (size_t)T * PW;
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((tw + 255) / 256)), dim3(256), 0, 0, table, tw, CW, 7u);
  hipLaunchKernelGGL(k_entries, dim3((unsigned)((npairs * 2 + 255) / 256)), dim3(256), 0, 0, entries, npairs * 2, T, linear, 11u);
  CHK(hipMemset(bad, 0, 4)); CHK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  float best[5] = {1e9f, 1e9f, 1e9f, 1e9f, 1e9f};
  for (int rep = 0; rep < 4; ++rep) {
    CHK(hipEventRecord(e0)); hipLaunchKernelGGL((k_aff_round<F, K, 0>), dim3((unsigned)nblk), dim3(64), 0, 0, table, entries, npairs, pref, total, out); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    best[0] = std::min(best[0], time_ms(e0, e1));
    CHK(hipEventRecord(e0)); hipLaunchKernelGGL((k_aff_round<F, K, 1>), dim3((unsigned)nblk), dim3(64), 0, 0, table, entries, npairs, pref, total, out); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    best[1] = std::min(best[1], time_ms(e0, e1));
    CHK(hipEventRecord(e0)); hipLaunchKernelGGL((k_aff_round<F, K, 2>), dim3((unsigned)nblk), dim3(64), 0, 0, table, entries, npairs, pref, total, out); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    best[2] = std::min(best[2], time_ms(e0, e1));
    CHK(hipEventRecord(e0)); hipLaunchKernelGGL((k_xyzz_chain<F, K>), dim3((unsigned)nblk), dim3(64), 0, 0, table, entries, npairs, xout); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    best[3] = std::min(best[3], time_ms(e0, e1));
  }
  CHK(hipGetLastError());
  // the split kernels ran last: `out` holds their result; check a sample of 64 K pairs
  const size_t step = npairs / 65536 ? npairs / 65536 : 1;
  hipLaunchKernelGGL(k_check<F>, dim3(1024), dim3(64), 0, 0, table, entries, npairs, step, out, bad); CHK(hipDeviceSynchronize());
  uint32_t hbad = 0; CHK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
  const double adds = (double)npairs, xadds = 2.0 * (double)npairs;
  printf("%s %-7s K=%3d pairs=%9zu | affine fused %7.3f ms (%6.1f ps/add) | fwd %7.3f + bwd %7.3f = %7.3f ms (%6.1f ps/add) | XYZZ chain over the same %zu entries %7.3f ms (%6.1f ps/add) | mismatches %u\n",
         group, linear ? "linear" : "gather", K, npairs, best[0], best[0] * 1e9 / adds, best[1], best[2], best[1] + best[2], (best[1] + best[2]) * 1e9 / adds, 2 * npairs, best[3], best[3] * 1e9 / xadds, hbad);
  fflush(stdout);
  CHK(hipFree(table)); CHK(hipFree(entries)); CHK(hipFree(pref)); CHK(hipFree(total)); CHK(hipFree(out)); CHK(hipFree(xout)); CHK(hipFree(bad));
}

int main(int argc, char** argv) {
  const double lg = argc > 1 ? atof(argv[1]) : 0.0;
  const size_t npairs = lg > 0 ? (size_t)pow(2.0, lg) : (size_t)13 * (1u << 19);      // 6.8 M: round 1 of a 2^20-term MSM with 13 windows
  const int which = argc > 2 ? atoi(argv[2]) : 3;                                      // bit 0: G1, bit 1: G2
  const int only_k = argc > 3 ? atoi(argv[3]) : 0;
  const uint32_t T = 13u << 20;
  hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
  printf("# tools/ubench/affine_round.hip on %s: %zu pair additions per launch, table of %u points; ps/add = kernel time / additions (XYZZ: one mixed addition per entry)\n", p.name, npairs, T);
  if (which & 1) {
    if (!only_k || only_k == 32) { run<FqOps, 32>("G1", npairs, T, 0); }
    if (!only_k || only_k == 64) { run<FqOps, 64>("G1", npairs, T, 0); run<FqOps, 64>("G1", npairs, T, 1); }
    if (!only_k || only_k == 128) { run<FqOps, 128>("G1", npairs, T, 0); }
  }
#ifndef NO_G2
  if (which & 2) {
    if (!only_k || only_k == 32) { run<Fq2Ops, 32>("G2", npairs, T, 0); run<Fq2Ops, 32>("G2", npairs, T, 1); }
    if (!only_k || only_k == 64) { run<Fq2Ops, 64>("G2", npairs, T, 0); }
  }
#endif
  return 0;
}
