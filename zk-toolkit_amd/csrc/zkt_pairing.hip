// Batched Tate pairing kernel: one pairing per lane (rows a10–a13 of SURVEY §8).
//   Pairing::tate   bls12_381/pairing.rs:86-100
// The algorithm and why it is bit-identical to the reference are in pairing.h.
#include "abi.h"
#include "zkt_internal.h"

namespace zkt {

__global__ void __launch_bounds__(64) k_tate(const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g2,
                                             uint32_t* __restrict__ out, size_t n, unsigned long long* err) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  Aff<FqOps> p = PtIO<FqOps>::ld(g1 + i * ABI_G1_WORDS);
  Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(g2 + i * ABI_G2_WORDS);
  if (p.inf || q.inf) {   // RationalFunction::new_* / eval_with_* panic on infinity (rational_function.rs:36,59)
    atomicMin(err, (unsigned long long)i);
    return;
  }
  Fq12 f = miller_g1_g2(p.x, p.y, q.x, q.y);
  st_fq12(out + i * 144, final_exponentiation(f));
}

// raw Miller values / Weil (row a14): which = 0 calc_g1_g2, 1 calc_g2_g1, 2 weil (pairing.rs:54-55,75-84)
__global__ void __launch_bounds__(64) k_miller_exact(int which, const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g2,
                                                     uint32_t* __restrict__ out, size_t n, unsigned long long* err) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  Aff<FqOps> p = PtIO<FqOps>::ld(g1 + i * ABI_G1_WORDS);
  Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(g2 + i * ABI_G2_WORDS);
  if (p.inf || q.inf) { atomicMin(err, (unsigned long long)i); return; }
  Fq12 r;
  if (which == 0) r = miller_g1_g2_exact(p.x, p.y, q.x, q.y);
  else if (which == 1) r = miller_g2_g1_exact(q.x, q.y, p.x, p.y);
  else r = fq12_mul(miller_g1_g2_exact(p.x, p.y, q.x, q.y), fq12_inv(miller_g2_g1_exact(q.x, q.y, p.x, p.y)));
  st_fq12(out + i * 144, r);
}
hipError_t launch_miller_exact(int which, const uint32_t* g1, const uint32_t* g2, uint32_t* out, size_t n, unsigned long long* err, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_miller_exact, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, which, g1, g2, out, n, err);
  return hipGetLastError();
}

hipError_t launch_tate(const uint32_t* g1, const uint32_t* g2, uint32_t* out, size_t n, unsigned long long* err, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_tate, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, g1, g2, out, n, err);
  return hipGetLastError();
}

}  // namespace zkt
