// Batched group kernels: one point operation per lane (rows a7, a8, a16 of SURVEY §8).
//   impl_affine_add!        curves/macros.rs:34-163
//   impl_scalar_mul_point!  curves/macros.rs:1-32
//   Neg                     bls12_381/g1_point.rs:177-195, g2_point.rs:91-107
// Points arrive and leave as canonical affine structs (include/zkt.h); inside a lane the
// point is Jacobian over Montgomery residues and is normalised once at the end.
#include "abi.h"
#include "zkt_internal.h"

namespace zkt {

static inline unsigned nblk(size_t n, int tpb) { return (unsigned)((n + tpb - 1) / tpb); }

template <class F>
__global__ void __launch_bounds__(64) k_group_add(const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n) {   // out may alias a
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  constexpr int W = PtIO<F>::WORDS;
  Aff<F> p = PtIO<F>::ld(a + i * W), q = PtIO<F>::ld(b + i * W);
  PtIO<F>::st(out + i * W, jac_to_aff(jac_add_aff(jac_from_aff(p), q)));
}
template <class F>
__global__ void __launch_bounds__(64) k_group_neg(const uint32_t* __restrict__ a, uint32_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  constexpr int W = PtIO<F>::WORDS;
  Aff<F> p = PtIO<F>::ld(a + i * W);
  if (!p.inf) p.y = F::neg(p.y);
  PtIO<F>::st(out + i * W, p);
}
template <class F>
__global__ void __launch_bounds__(64) k_group_mul(const uint32_t* __restrict__ pts, int pt_stride, const uint32_t* __restrict__ scalars, int kw, int k_stride,
                                                  uint32_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  constexpr int W = PtIO<F>::WORDS;
  Aff<F> p = PtIO<F>::ld(pts + i * (size_t)pt_stride);   // pt_stride = W, or 0 for one fixed base (g * y_i, crs.rs:85-135)
  const uint32_t* k = scalars + i * (size_t)k_stride;   // k_stride = kw, or 0 for one scalar for every point (gg * x, bulletproofs.rs:44)
  // MSB-first double-and-add (see curve.h::scalar_mul_aff), scalar read from global per bit word
  Jac<F> acc = jac_inf<F>();
  bool started = false;
  if (!p.inf) {
    for (int w = kw - 1; w >= 0; --w) {
      uint32_t word = k[w];
      for (int bit = 31; bit >= 0; --bit) {
        if (started) acc = jac_dbl(acc);
        if ((word >> bit) & 1) { acc = jac_add_aff(acc, p); started = true; }
      }
    }
  }
  PtIO<F>::st(out + i * W, jac_to_aff(acc));
}

template <class F> static hipError_t add_t(const uint32_t* a, const uint32_t* b, uint32_t* o, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_group_add<F>, dim3(nblk(n, 64)), dim3(64), 0, s, a, b, o, n); return hipGetLastError(); }
template <class F> static hipError_t neg_t(const uint32_t* a, uint32_t* o, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_group_neg<F>, dim3(nblk(n, 64)), dim3(64), 0, s, a, o, n); return hipGetLastError(); }
template <class F> static hipError_t mul_t(const uint32_t* p, bool fixed, const uint32_t* k, int kw, bool fixed_k, uint32_t* o, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_group_mul<F>, dim3(nblk(n, 64)), dim3(64), 0, s, p, fixed ? 0 : PtIO<F>::WORDS, k, kw, fixed_k ? 0 : kw, o, n); return hipGetLastError(); }

hipError_t launch_group_add(int grp, const uint32_t* a, const uint32_t* b, uint32_t* o, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  switch (grp) { case G_G1: return add_t<FqOps>(a, b, o, n, s); case G_G2: return add_t<Fq2Ops>(a, b, o, n, s); case G_SECP: return add_t<SpOps>(a, b, o, n, s); }
  return hipErrorInvalidValue;
}
hipError_t launch_group_neg(int grp, const uint32_t* a, uint32_t* o, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  switch (grp) { case G_G1: return neg_t<FqOps>(a, o, n, s); case G_G2: return neg_t<Fq2Ops>(a, o, n, s); case G_SECP: return neg_t<SpOps>(a, o, n, s); }
  return hipErrorInvalidValue;
}
hipError_t launch_group_mul(int grp, const uint32_t* p, const uint32_t* k, int kw, uint32_t* o, size_t n, hipStream_t s, bool fixed_base, bool fixed_scalar) {
  if (n == 0) return hipSuccess;
  switch (grp) { case G_G1: return mul_t<FqOps>(p, fixed_base, k, kw, fixed_scalar, o, n, s); case G_G2: return mul_t<Fq2Ops>(p, fixed_base, k, kw, fixed_scalar, o, n, s); case G_SECP: return mul_t<SpOps>(p, fixed_base, k, kw, fixed_scalar, o, n, s); }
  return hipErrorInvalidValue;
}
// sum of n affine points by rounds of pairwise additions (in place, `pts` is clobbered; result in pts[0]).
// Straightforward O(n) adds, log2(n) launches — the G2 / secp256k1 sums behind eval_with_g2_hidings
// (polynomial.rs:283-293) and (AffinePoints * PrimeFieldElems).sum() (secp256k1/affine_points.rs:25-31,123-144).
hipError_t launch_group_sum_inplace(int grp, uint32_t* pts, size_t n, hipStream_t s) {
  const size_t W = grp == G_G1 ? ABI_G1_WORDS : grp == G_G2 ? ABI_G2_WORDS : ABI_SECP_WORDS;
  while (n > 1) {
    size_t half = n / 2, odd = n & 1;
    // pts[i] += pts[half + odd + i] for i < half; the middle element (if odd) stays in place
    hipError_t e = launch_group_add(grp, pts, pts + (half + odd) * W, pts, half, s);
    if (e != hipSuccess) return e;
    n = half + odd;
  }
  return hipSuccess;
}

}  // namespace zkt
