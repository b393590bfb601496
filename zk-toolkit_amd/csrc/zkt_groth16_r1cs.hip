// Groth16 at scale on the reference's own evaluation domain {1..n}  (SURVEY §8 row f-3).
//
// The reference interpolates every wire's column of the R1CS over x = 1..n (QAP::build_polynomial, qap/qap.rs:33-97),
// keeps (m+1) dense polynomials, and the prover evaluates each "in the exponent" (prover.rs:103-117) and divides
// a*b - c by t in coefficient form (prover.rs:64-71): O(m n) group operations and O(n^2) field work, with (m+1) n
// coefficients resident — 2^40 of them at the 2^20-constraint configuration.  The proof points only depend on the
// polynomials as functions, so the same group elements are produced here without ever forming a coefficient:
//
//   sum_i a_i u_i(x)        = sum_j (A w)_j L_j(x)                   L_j = Lagrange basis of {1..n}
//   sum_k h_k x^k t(x)/delta = sum_s h(n+s) Lambda_s(x) t(x)/delta    Lambda_s = Lagrange basis of E = {n+1..2n-1}
//   p(n+s) for p in {a,b,c} from p(1..n):  p(n+s) = t(n+s) * sum_j [p(j)/t'(j)] / (n+s-j)    — one cyclic convolution
//                                                                       with the kernel 1/d (Fr NTT, size >= 2n)
//   h(n+s) = (a b - c)(n+s) / t(n+s)
//
// and the single scalar multiplications of prover.rs:118-140 fold into the sums: with A = alpha + sum_A + r delta and
// B_g1 = beta + sum_B + s delta,  s A + r B_g1 - r s delta = s alpha + r beta + r s delta + sum_j (s (A w)_j + r (B w)_j) L_j(x).
// A proof is therefore 3 sparse mat-vecs, 3+3 NTTs and exactly three MSMs over device-resident bases derived from the trapdoor:
//   A:  [L_j(x)]_1 | alpha, delta                                      scalars (A w)_j | 1, r
//   B:  [L_j(x)]_2 | beta, delta                                       scalars (B w)_j | 1, s
//   C:  [L_j(x)]_1 | uvw_wit | [Lambda_s(x) t(x)/delta]_1 | alpha, beta, delta
//                                                                      scalars s (A w)_j + r (B w)_j | a_i | h(n+s) | s, r, r s
// whose affine outputs ARE the proof points — no per-proof single-lane scalar multiplication remains.
// The C sum is evaluated as TWO resident MSMs (round 3): C1 over [L] | uvw_wit | alpha, beta, delta, whose scalars exist as soon as the mat-vecs are done,
// and C2 over [Lambda t/delta] (n - 1 terms), the only part that waits for the quotient's six NTTs.  C1 (2n + 3 terms) then runs beside A and B under the
// NTT chain, and what follows the chain is a third of the old C sum: sort 4.0 -> 1.3 ms and accumulate 6.5 -> 2.3 ms off a single proof's critical path
// (profiles/r03_groth16_timeline.txt).  C = C1 + C2 is one Jacobian addition on the device.  tests/test_r1cs_domain_math.py checks
// the identity in python integers; tests/test_gpu_groth16_r1cs.py checks the proof points bit-for-bit against the oracle's
// restatement of the reference prover on the dense QAP.
#include <vector>
#include <memory>
#include <mutex>
#include <cstring>
#include <cstdio>
#include "abi.h"
#include "zkt_internal.h"
#include "../../include/zkt.h"

namespace zkt {
typedef FrC C;
typedef Fp<FrC> Fr;
static constexpr int FW = 8;          // u32 words of an Fr element (canonical and Montgomery alike)

__device__ inline Fr ldm(const uint32_t* p) { return ld_raw<C>(p); }       // Montgomery in memory
__device__ inline void stm(uint32_t* p, const Fr& a) { st_raw<C>(p, a); }
__device__ inline Fr fr_small(uint32_t k) { uint32_t w[8] = {k, 0, 0, 0, 0, 0, 0, 0}; return fp_from_words<C>(w); }

// ---- elementwise helpers ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_to_mont(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  stm(out + i * FW, ld_fp<C>(in + i * FW));
}
__global__ void __launch_bounds__(256) k_from_mont(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  st_fp<C>(out + i * FW, ldm(in + i * FW));
}
__global__ void __launch_bounds__(256) k_inv(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n, unsigned long long* err) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  Fr v = ldm(in + i * FW);
  if (fp_is_zero(v)) { atomicMin(err, (unsigned long long)i); stm(out + i * FW, v); return; }
  stm(out + i * FW, fp_inv(v));
}
// out[0] = 1, out[k] = k  (prefix product = k!)
__global__ void __launch_bounds__(256) k_iota(uint32_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  stm(out + i * FW, fr_small(i ? (uint32_t)i : 1u));
}
// out[j-1] = x - j, j = 1..cnt
__global__ void __launch_bounds__(256) k_x_minus(const uint32_t* __restrict__ consts, uint32_t* __restrict__ out, size_t cnt) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= cnt) return;
  stm(out + i * FW, fp_sub(ldm(consts), fr_small((uint32_t)(i + 1))));
}
// out[0] = 1, out[k>0] = w   (prefix product = w^k)
__global__ void __launch_bounds__(256) k_fill_pow(const uint32_t* __restrict__ w, uint32_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  stm(out + i * FW, i ? ldm(w) : fp_one<C>());
}

// ---- inclusive prefix product (factorials, prod (x - j), twiddle tables) ------------------------------------------
static constexpr int SC_TPB = 256, SC_ITEMS = 8, SC_TILE = SC_TPB * SC_ITEMS;
__global__ void __launch_bounds__(SC_TPB) k_scanmul_tile(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n, uint32_t* __restrict__ tile_total) {
  __shared__ uint32_t lds[SC_TPB * FW];
  const int t = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * SC_TILE + (size_t)t * SC_ITEMS;
  Fr loc[SC_ITEMS]; Fr run = fp_one<C>();
#pragma unroll
  for (int k = 0; k < SC_ITEMS; ++k) { if (base + k < n) run = fp_mul(run, ldm(in + (base + k) * FW)); loc[k] = run; }
  stm(lds + t * FW, run); __syncthreads();
  for (int d = 1; d < SC_TPB; d <<= 1) {                 // Hillis-Steele over the 256 thread totals
    Fr o = t >= d ? ldm(lds + (t - d) * FW) : fp_one<C>(); __syncthreads();
    if (t >= d) { run = fp_mul(o, run); stm(lds + t * FW, run); } __syncthreads();
  }
  Fr excl = t ? ldm(lds + (t - 1) * FW) : fp_one<C>();
#pragma unroll
  for (int k = 0; k < SC_ITEMS; ++k) if (base + k < n) stm(out + (base + k) * FW, fp_mul(excl, loc[k]));
  if (t == SC_TPB - 1) stm(tile_total + (size_t)blockIdx.x * FW, run);
}
__global__ void __launch_bounds__(256) k_scanmul_apply(uint32_t* __restrict__ data, size_t n, const uint32_t* __restrict__ tile_prefix) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  size_t tile = i / SC_TILE; if (tile == 0) return;
  stm(data + i * FW, fp_mul(ldm(tile_prefix + (tile - 1) * FW), ldm(data + i * FW)));
}

// ---- CSR mat-vec in Fr (Montgomery): out[r] = sum_k val[k] * vec[idx[k]] ---------------------------------------------
// One lane per row; rows longer than SPMV_LONG (the transposed "one" wire of setup has a term in every constraint)
// are left to k_spmv_long: one block per listed row, strided partial sums and an LDS tree.
static constexpr uint32_t SPMV_LONG = 4096;
__global__ void __launch_bounds__(256) k_spmv(const uint32_t* __restrict__ ptr, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ val,
                                              const uint32_t* __restrict__ vec, uint32_t* __restrict__ out, size_t rows) {
  size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; if (r >= rows) return;
  const uint32_t beg = ptr[r], end = ptr[r + 1];
  if (end - beg > SPMV_LONG) return;
  Fr acc = fp_zero<C>();
  for (uint32_t k = beg; k < end; ++k) acc = fp_add(acc, fp_mul(ldm(val + (size_t)k * FW), ldm(vec + (size_t)idx[k] * FW)));
  stm(out + r * FW, acc);
}
__global__ void __launch_bounds__(256) k_spmv_long(const uint32_t* __restrict__ ptr, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ val,
                                                   const uint32_t* __restrict__ vec, uint32_t* __restrict__ out, const uint32_t* __restrict__ long_rows) {
  __shared__ uint32_t lds[256 * FW];
  const uint32_t r = long_rows[blockIdx.x], t = threadIdx.x;
  Fr acc = fp_zero<C>();
  for (uint32_t k = ptr[r] + t; k < ptr[r + 1]; k += 256) acc = fp_add(acc, fp_mul(ldm(val + (size_t)k * FW), ldm(vec + (size_t)idx[k] * FW)));
  stm(lds + t * FW, acc); __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) {
    if ((int)t < d) { acc = fp_add(acc, ldm(lds + (t + d) * FW)); stm(lds + t * FW, acc); }
    __syncthreads();
  }
  if (t == 0) stm(out + (size_t)r * FW, acc);
}

// ---- setup scalars ----------------------------------------------------------------------------------------
// consts layout (Montgomery, 8 words each)
enum { K_X = 0, K_ALPHA, K_BETA, K_GINV, K_DINV, K_TX, K_TE, K_HB, K_OMEGA, K_OMEGA_INV, K_NINV, K_COUNT };
__global__ void k_setup_consts(const uint32_t* __restrict__ trap /*alpha,beta,gamma,delta,x canonical*/, uint32_t* __restrict__ consts, int logN) {
  if (threadIdx.x || blockIdx.x) return;
  stm(consts + K_ALPHA * FW, ld_fp<C>(trap)); stm(consts + K_BETA * FW, ld_fp<C>(trap + 8));
  stm(consts + K_GINV * FW, fp_inv(ld_fp<C>(trap + 16))); stm(consts + K_DINV * FW, fp_inv(ld_fp<C>(trap + 24)));
  stm(consts + K_X * FW, ld_fp<C>(trap + 32));
  uint32_t rw[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) rw[i] = fr_root_word(i);
  Fr w = fp_from_words<C>(rw);
  for (int k = FR_TWO_ADICITY; k > logN; --k) w = fp_sqr(w);                      // order 2^logN
  stm(consts + K_OMEGA * FW, w); stm(consts + K_OMEGA_INV * FW, fp_inv(w));
  Fr two = fr_small(2), nn = fp_one<C>();
  for (int k = 0; k < logN; ++k) nn = fp_mul(nn, two);
  stm(consts + K_NINV * FW, fp_inv(nn));
}
// after the prefix product of (x - j): t(x) = pre[n-1], T_E(x) = pre[2n-2] / pre[n-1], hb = t(x) T_E(x) / delta
__global__ void k_setup_consts2(const uint32_t* __restrict__ pre, size_t n, uint32_t* __restrict__ consts) {
  if (threadIdx.x || blockIdx.x) return;
  Fr tx = ldm(pre + (n - 1) * FW);
  Fr te = n >= 2 ? fp_mul(ldm(pre + (2 * n - 2) * FW), fp_inv(tx)) : fp_one<C>();
  stm(consts + K_TX * FW, tx); stm(consts + K_TE * FW, te);
  stm(consts + K_HB * FW, fp_mul(fp_mul(tx, te), ldm(consts + K_DINV * FW)));
}
// j = 1..n: cinv[j-1] = 1/t'(j) = (-1)^(n-j) / ((j-1)! (n-j)!);  L[j-1] = t(x) / (x - j) * cinv[j-1]
__global__ void __launch_bounds__(256) k_lagrange(const uint32_t* __restrict__ consts, const uint32_t* __restrict__ invfact, const uint32_t* __restrict__ xinv,
                                                  size_t n, uint32_t* __restrict__ cinv, uint32_t* __restrict__ L_mont, uint32_t* __restrict__ L_canon) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;     // j = i + 1
  Fr c = fp_mul(ldm(invfact + i * FW), ldm(invfact + (n - 1 - i) * FW));
  if ((n - 1 - i) & 1) c = fp_neg(c);
  stm(cinv + i * FW, c);
  Fr l = fp_mul(fp_mul(ldm(consts + K_TX * FW), ldm(xinv + i * FW)), c);
  stm(L_mont + i * FW, l); st_fp<C>(L_canon + i * FW, l);
}
// s = 1..n-1: hb[s-1] = t(x)/delta * T_E(x) / (x - (n+s)) * (-1)^(n-1-s) / ((s-1)! (n-1-s)!)   (canonical: fixed-base scalars)
//             P[s-1] = t(n+s) = (n+s-1)! / (s-1)!                                               (Montgomery)
__global__ void __launch_bounds__(256) k_hbasis(const uint32_t* __restrict__ consts, const uint32_t* __restrict__ fact, const uint32_t* __restrict__ invfact,
                                                const uint32_t* __restrict__ xinv, size_t n, uint32_t* __restrict__ hb_canon, uint32_t* __restrict__ P) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i + 1 >= n) return;   // s = i + 1
  Fr c = fp_mul(ldm(invfact + i * FW), ldm(invfact + (n - 2 - i) * FW));
  if ((n - 2 - i) & 1) c = fp_neg(c);
  st_fp<C>(hb_canon + i * FW, fp_mul(fp_mul(ldm(consts + K_HB * FW), ldm(xinv + (n + i) * FW)), c));
  stm(P + i * FW, fp_mul(ldm(fact + (n + i) * FW), ldm(invfact + i * FW)));
}
// The quotient's convolution, cut into blocks (overlap-save).  A rank needs S_p(s) = sum_{j=1..n} f_j / (n+s-j) for the `cnt` values
// s = s0 .. s0+cnt-1 whose bases [Lambda_s t/delta] it holds.  The inputs j = 1..n are cut into Q blocks of Bi (a power of two >= cnt); block q meets the
// kernel slice g[base_q + e], base_q = n + s0 - 1 - q Bi, e = -(Bi-1) .. cnt-1, in a cyclic convolution of size M = 2 Bi (negative e at M + e), and the Q
// products are summed in the spectrum, so a rank runs Q forward transforms of size M per polynomial and ONE inverse — no exchange between ranks, and with
// one rank (cnt = n-1, Q = 1, M >= 2n) exactly the single convolution of size >= 2n.  g[d] = 1/d = (d-1)!/d! for 1 <= d <= 2n-1, 0 elsewhere (d <= 0 only
// ever meets the zero padding of the last block).  Layout [q][M].
__global__ void __launch_bounds__(256) k_recip_blocks(const uint32_t* __restrict__ fact, const uint32_t* __restrict__ invfact, size_t n, size_t s0, size_t cnt,
                                                      size_t Bi, size_t Q, uint32_t* __restrict__ g) {
  const size_t M = 2 * Bi, i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= Q * M) return;
  const size_t q = i / M, e = i % M;
  const long long base = (long long)(n + s0 - 1) - (long long)(q * Bi);
  long long d = 0;
  if (e < cnt) d = base + (long long)e; else if (e > M - Bi) d = base - (long long)(M - e);
  stm(g + i * FW, (d >= 1 && d <= (long long)(2 * n - 1)) ? fp_mul(ldm(fact + (d - 1) * FW), ldm(invfact + d * FW)) : fp_zero<C>());
}
// y_i = (beta u_i(x) + alpha v_i(x) + w_i(x)) / (gamma | delta), canonical  (crs.rs:66-84)
__global__ void __launch_bounds__(256) k_uvw(const uint32_t* __restrict__ consts, const uint32_t* __restrict__ ue, const uint32_t* __restrict__ ve, const uint32_t* __restrict__ we,
                                             size_t l, size_t rows, uint32_t* __restrict__ y_canon) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= rows) return;
  Fr s = fp_add(fp_add(fp_mul(ldm(consts + K_BETA * FW), ldm(ue + i * FW)), fp_mul(ldm(consts + K_ALPHA * FW), ldm(ve + i * FW))), ldm(we + i * FW));
  st_fp<C>(y_canon + i * FW, fp_mul(s, ldm(consts + (i <= l ? K_GINV : K_DINV) * FW)));
}

// ---- Fr NTT of size N = 2^logN ------------------------------------------------------------------------------------
// forward = decimation in frequency (natural order in, bit-reversed out); inverse = decimation in time on the
// bit-reversed spectrum (natural order out), so no reordering pass exists; the 1/N lives in the precomputed kernel spectrum.
// HBM-bound (one butterfly = one Fr multiply per 64 B moved), so stages are fused through LDS: a launch runs `cnt`
// consecutive radix-2 stages (butterfly distances 2^lo .. 2^(lo+cnt-1)) on a tile of 2^cnt strided rows x 2^cbits
// adjacent columns (<= 1024 elements, 32 KB), every element read and written once per launch, rows of >= 128 B contiguous.
// logN = 21 is three launches per transform (10 + 8 + 3 stages) instead of 21.
// A launch covers any number of transforms of the same size: consecutive ones simply continue blockIdx.x (the twiddle of a butterfly depends on its position
// inside its group only), blockIdx.y steps over arrays `ystride` elements apart.  `mulvec` is indexed like the consecutive transforms of one array.
// (one-wave workgroups for the transform were measured at the end of round 3, in case its four-wave workgroups were what starved beside an accumulate grid: 9.1 ms against 7.6
// for a shard of a proof, 40 against 50 proofs/s on one GPU — not that)
static constexpr int NTT_TILE_LOG = 10, NTT_TPB = 256;
template <bool DIF>
__global__ void __launch_bounds__(NTT_TPB) k_ntt_group(uint32_t* __restrict__ a, int logN, int lo, int cnt, int cbits, const uint32_t* __restrict__ tw,
                                                       const uint32_t* __restrict__ mulvec, size_t ystride) {
  __shared__ uint32_t lds[(1 << NTT_TILE_LOG) * FW];
  a += (size_t)blockIdx.y * ystride * FW;                         // grid.y: independent arrays (the three polynomials), sharing `mulvec`
  const int tile = 1 << (cnt + cbits), cmask = (1 << cbits) - 1;
  const size_t tiles_per_hi = (size_t)1 << (lo - cbits);
  const size_t hi = blockIdx.x / tiles_per_hi, c0 = (blockIdx.x % tiles_per_hi) << cbits;
  const size_t gbase = (hi << (lo + cnt)) | c0;
  for (int e = threadIdx.x; e < tile; e += NTT_TPB) {
    const size_t g = gbase | ((size_t)(e >> cbits) << lo) | (size_t)(e & cmask);
    const uint4* src = reinterpret_cast<const uint4*>(a + g * FW);
    uint4* dst = reinterpret_cast<uint4*>(lds + e * FW);
    dst[0] = src[0]; dst[1] = src[1];
  }
  __syncthreads();
  for (int st = 0; st < cnt; ++st) {
    const int t = DIF ? cnt - 1 - st : st;                       // local butterfly distance 2^t rows
    for (int b = threadIdx.x; b < tile / 2; b += NTT_TPB) {
      const int cc = b & cmask, kb = b >> cbits;
      const int k0 = ((kb >> t) << (t + 1)) | (kb & ((1 << t) - 1));
      const int e0 = (k0 << cbits) | cc, e1 = e0 + (1 << (t + cbits));
      const size_t j = ((size_t)(k0 & ((1 << t) - 1)) << lo) | c0 | (size_t)cc;      // position inside the butterfly group
      const Fr w = ldm(tw + (j << (logN - 1 - lo - t)) * FW);
      Fr u = ldm(lds + e0 * FW), v = ldm(lds + e1 * FW);
      if (DIF) { stm(lds + e0 * FW, fp_add(u, v)); stm(lds + e1 * FW, fp_mul(fp_sub(u, v), w)); }
      else { v = fp_mul(v, w); stm(lds + e0 * FW, fp_add(u, v)); stm(lds + e1 * FW, fp_sub(u, v)); }
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < tile; e += NTT_TPB) {
    const size_t g = gbase | ((size_t)(e >> cbits) << lo) | (size_t)(e & cmask);
    if (mulvec) stm(a + g * FW, fp_mul(ldm(lds + e * FW), ldm(mulvec + g * FW)));      // fused pointwise product with a spectrum
    else {
      const uint4* src = reinterpret_cast<const uint4*>(lds + e * FW);
      uint4* dst = reinterpret_cast<uint4*>(a + g * FW);
      dst[0] = src[0]; dst[1] = src[1];
    }
  }
}
__global__ void __launch_bounds__(256) k_scale_all(uint32_t* __restrict__ a, const uint32_t* __restrict__ s, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  stm(a + i * FW, fp_mul(ldm(a + i * FW), ldm(s)));
}

// ---- prover Fr stage -----------------------------------------------------------------------------------------
// X[p][q][e] = f_p(q Bi + 1 + e) = p(j) / t'(j) at j = q Bi + 1 + e for e < Bi and j <= n, zero elsewhere (the second half of every block of M = 2 Bi, the tail of the last block)
__global__ void __launch_bounds__(256) k_prep_blocks(const uint32_t* __restrict__ z0, const uint32_t* __restrict__ z1, const uint32_t* __restrict__ z2,
                                                     const uint32_t* __restrict__ cinv, size_t n, size_t Bi, size_t Q, uint32_t* __restrict__ X) {
  const size_t M = 2 * Bi, i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= Q * M) return;
  const size_t q = i / M, e = i % M, j0 = q * Bi + e;            // j - 1
  const uint32_t* z = blockIdx.y == 0 ? z0 : blockIdx.y == 1 ? z1 : z2;
  stm(X + ((size_t)blockIdx.y * Q * M + i) * FW, (e < Bi && j0 < n) ? fp_mul(ldm(z + j0 * FW), ldm(cinv + j0 * FW)) : fp_zero<C>());
}
// the Q block spectra (already multiplied by their kernel spectra) summed into block 0 of every polynomial
__global__ void __launch_bounds__(256) k_sum_blocks(uint32_t* __restrict__ X, size_t M, size_t Q) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; if (e >= M) return;
  uint32_t* x = X + ((size_t)blockIdx.y * Q * M + e) * FW;
  Fr acc = ldm(x);
  for (size_t q = 1; q < Q; ++q) acc = fp_add(acc, ldm(x + q * M * FW));
  stm(x, acc);
}
// h(n+s) = (P Sa * P Sb - P Sc) / t(n+s) with t(n+s) = P  =>  P Sa Sb - Sc,  s = s0 + i, i < cnt, canonical for the MSM.  S_p(s0 + i) = X[p][0][i]; P and h are indexed by s - 1.
__global__ void __launch_bounds__(256) k_hvals(const uint32_t* __restrict__ X, size_t pstride, const uint32_t* __restrict__ P, size_t s0, size_t cnt, uint32_t* __restrict__ h_canon) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= cnt) return;
  const size_t k = s0 - 1 + i;
  Fr h = fp_sub(fp_mul(fp_mul(ldm(P + k * FW), ldm(X + i * FW)), ldm(X + (pstride + i) * FW)), ldm(X + (2 * pstride + i) * FW));
  st_fp<C>(h_canon + k * FW, h);
}
// scalar vectors of the three MSMs (canonical).  rs = {r, s} canonical.
//   sA = [Az | 1 | r]   sB = [Bz | 1 | s]   sC = [s Az + r Bz | wires[l+1..m] | s | r | r s || h]   (C1 || C2; h is written in place by k_hvals)
__global__ void __launch_bounds__(256) k_prove_scalars(const uint32_t* __restrict__ Az, const uint32_t* __restrict__ Bz, const uint32_t* __restrict__ wires_c,
                                                       const uint32_t* __restrict__ rs, size_t n, size_t l, size_t m,
                                                       uint32_t* __restrict__ sA, uint32_t* __restrict__ sB, uint32_t* __restrict__ sC) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t nw = m - l;
  if (i < n) {
    Fr a = ldm(Az + i * FW), b = ldm(Bz + i * FW), r = ld_fp<C>(rs), s = ld_fp<C>(rs + FW);
    st_fp<C>(sA + i * FW, a); st_fp<C>(sB + i * FW, b);
    st_fp<C>(sC + i * FW, fp_add(fp_mul(s, a), fp_mul(r, b)));
  } else if (i < n + nw) {
    const uint32_t* src = wires_c + (l + 1 + (i - n)) * FW;
#pragma unroll
    for (int k = 0; k < FW; ++k) sC[i * FW + k] = src[k];
  } else if (i == n + nw) {                                   // the constant tails
    Fr r = ld_fp<C>(rs), s = ld_fp<C>(rs + FW);
    uint32_t* tA = sA + n * FW; uint32_t* tB = sB + n * FW; uint32_t* tC = sC + (n + nw) * FW;
    st_fp<C>(tA, fp_one<C>()); st_fp<C>(tA + FW, r);
    st_fp<C>(tB, fp_one<C>()); st_fp<C>(tB + FW, s);
    st_fp<C>(tC, s); st_fp<C>(tC + FW, r); st_fp<C>(tC + 2 * FW, fp_mul(r, s));
  }
}
}  // namespace zkt

using namespace zkt;

namespace {
#define RCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "[zkt] HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); return ZKT_ERR_DEVICE; } } while (0)
#define ZCHK(x) do { int _rc = (x); if (_rc != ZKT_OK) return _rc; } while (0)
inline unsigned nb(size_t n) { return (unsigned)((n + 255) / 256); }
const size_t G1B = sizeof(zkt_g1_affine), G2B = sizeof(zkt_g2_affine), FRB = 32;
const uint64_t G1_GEN[13] = {0xfb3af00adb22c6bbull, 0x6c55e83ff97a1aefull, 0xa14e3a3f171bac58ull, 0xc3688c4f9774b905ull, 0x2695638c4fa9ac0full, 0x17f1d3a73197d794ull,
                             0x0caa232946c5e7e1ull, 0xd03cc744a2888ae4ull, 0x00db18cb2c04b3edull, 0xfcf5e095d5d00af6ull, 0xa09e30ed741d8ae4ull, 0x08b3f481e3aaa0f1ull, 0};   // g1_point.rs:38-47
const uint64_t G2_GEN[25] = {0xe5ac7d055d042b7eull, 0x334cf11213945d57ull, 0xb5da61bbdc7f5049ull, 0x596bd0d09920b61aull, 0x7dacd3a088274f65ull, 0x13e02b6052719f60ull,
                             0xd48056c8c121bdb8ull, 0x0bac0326a805bbefull, 0xb4510b647ae3d177ull, 0xc6e47ad4fa403b02ull, 0x260805272dc51051ull, 0x024aa2b2f08f0a91ull,
                             0xaaa9075ff05f79beull, 0x3f370d275cec1da1ull, 0x267492ab572e99abull, 0xcb3e287e85a763afull, 0x32acd2b02bc28b99ull, 0x0606c4a02ea734ccull,
                             0xe193548608b82801ull, 0x923ac9cc3baca289ull, 0x6d429a695160d12cull, 0xadfd9baa8cbdd3a7ull, 0x8cc9cdc6da2e351aull, 0x0ce5d527727d6e11ull, 0};   // g2_point.rs:36-46

struct DBuf {                       // owning device buffer
  void* p = nullptr; size_t bytes = 0;
  int alloc(size_t b) { bytes = b ? b : 32; return hipMalloc(&p, bytes) == hipSuccess ? ZKT_OK : ZKT_ERR_DEVICE; }
  void release() { if (p) hipFree(p); p = nullptr; }
  ~DBuf() { release(); }
  uint32_t* w() const { return (uint32_t*)p; }
  DBuf() = default; DBuf(const DBuf&) = delete; DBuf& operator=(const DBuf&) = delete;
};
struct Csr { DBuf ptr, idx, val, long_rows; size_t rows = 0, nnz = 0, n_long = 0; };

// inclusive prefix product, in place allowed (in == out); two levels of tiles cover 2048^2 elements
int scan_mul(const uint32_t* in, uint32_t* out, size_t n, hipStream_t s) {
  if (n == 0) return ZKT_OK;
  const size_t tiles = (n + SC_TILE - 1) / SC_TILE;
  if (tiles > (size_t)SC_TILE) return ZKT_ERR_SHAPE;
  DBuf tot, tot2; ZCHK(tot.alloc(tiles * FRB)); ZCHK(tot2.alloc(FRB));
  hipLaunchKernelGGL(k_scanmul_tile, dim3((unsigned)tiles), dim3(SC_TPB), 0, s, in, out, n, tot.w());
  if (tiles > 1) {
    hipLaunchKernelGGL(k_scanmul_tile, dim3(1), dim3(SC_TPB), 0, s, (const uint32_t*)tot.w(), tot.w(), tiles, tot2.w());
    hipLaunchKernelGGL(k_scanmul_apply, dim3(nb(n)), dim3(256), 0, s, out, n, (const uint32_t*)tot.w());
  }
  RCHK(hipStreamSynchronize(s));     // the tile totals die with this frame
  return ZKT_OK;
}
// stage groups: the contiguous one first (distances 1..2^(c0-1)), then strided groups of <= 8 stages with >= 4 adjacent columns
struct NttGroup { int lo, cnt, cbits; };
int ntt_groups(int logN, NttGroup* g) {
  int k = 0, c0 = logN < NTT_TILE_LOG ? logN : NTT_TILE_LOG;
  g[k++] = {0, c0, 0};
  for (int lo = c0, rem = logN - c0; rem > 0;) {
    int cnt = rem < 8 ? rem : 8, cb = NTT_TILE_LOG - cnt; if (cb > lo) cb = lo;
    g[k++] = {lo, cnt, cb}; lo += cnt; rem -= cnt;
  }
  return k;
}
// forward: natural -> bit-reversed; if `mulvec`, the spectrum is multiplied by it on the way out.  `batch` consecutive transforms per array, `ny` arrays `ystride` elements apart.
int ntt_forward(uint32_t* a, int logN, const uint32_t* tw, const uint32_t* mulvec, hipStream_t s, size_t batch = 1, unsigned ny = 1, size_t ystride = 0) {
  NttGroup g[8]; const int k = ntt_groups(logN, g);
  for (int i = k - 1; i >= 0; --i)
    hipLaunchKernelGGL(k_ntt_group<true>, dim3((unsigned)((batch << logN) >> (g[i].cnt + g[i].cbits)), ny), dim3(NTT_TPB), 0, s, a, logN, g[i].lo, g[i].cnt, g[i].cbits, tw,
                       i == 0 ? mulvec : (const uint32_t*)nullptr, ystride);
  RCHK(hipGetLastError()); return ZKT_OK;
}
int ntt_inverse(uint32_t* a, int logN, const uint32_t* twinv, hipStream_t s, size_t batch = 1, unsigned ny = 1, size_t ystride = 0) {
  NttGroup g[8]; const int k = ntt_groups(logN, g);
  for (int i = 0; i < k; ++i)
    hipLaunchKernelGGL(k_ntt_group<false>, dim3((unsigned)((batch << logN) >> (g[i].cnt + g[i].cbits)), ny), dim3(NTT_TPB), 0, s, a, logN, g[i].lo, g[i].cnt, g[i].cbits, twinv,
                       (const uint32_t*)nullptr, ystride);
  RCHK(hipGetLastError()); return ZKT_OK;
}
void spmv(const Csr& M, const uint32_t* vec, uint32_t* out, hipStream_t s) {
  hipLaunchKernelGGL(k_spmv, dim3(nb(M.rows)), dim3(256), 0, s, (const uint32_t*)M.ptr.w(), (const uint32_t*)M.idx.w(), (const uint32_t*)M.val.w(), vec, out, M.rows);
  if (M.n_long) hipLaunchKernelGGL(k_spmv_long, dim3((unsigned)M.n_long), dim3(256), 0, s, (const uint32_t*)M.ptr.w(), (const uint32_t*)M.idx.w(), (const uint32_t*)M.val.w(), vec, out, (const uint32_t*)M.long_rows.w());
}
// host CSR (reference order: one sparse row per constraint, r1cs.rs / constraint.rs:5-9) -> device CSR and its transpose
int upload_csr(const zkt_sparse_rows* M, size_t n, size_t cols, Csr& rowwise, Csr& colwise, hipStream_t s) {
  const size_t nnz = (size_t)M->rowptr[n];
  if (nnz >= 0xffffffffull || M->rowptr[0] != 0) return ZKT_ERR_SHAPE;
  std::vector<uint32_t> rp(n + 1), cp(cols + 1, 0), ridx(nnz ? nnz : 1), order(nnz ? nnz : 1);
  for (size_t j = 0; j <= n; ++j) { if (j && M->rowptr[j] < M->rowptr[j - 1]) return ZKT_ERR_SHAPE; rp[j] = (uint32_t)M->rowptr[j]; }
  for (size_t k = 0; k < nnz; ++k) { if (M->col[k] >= cols) return ZKT_ERR_SHAPE; cp[M->col[k] + 1]++; }
  for (size_t i = 0; i < cols; ++i) cp[i + 1] += cp[i];
  std::vector<uint32_t> cur(cp.begin(), cp.end() - 1);
  std::vector<uint64_t> tval((nnz ? nnz : 1) * 4);
  for (size_t j = 0; j < n; ++j)
    for (uint32_t k = rp[j]; k < rp[j + 1]; ++k) { uint32_t d = cur[M->col[k]]++; ridx[d] = (uint32_t)j; memcpy(&tval[(size_t)d * 4], &M->val[(size_t)k * 4], 32); }
  auto put = [&](Csr& c, const std::vector<uint32_t>& p, const uint32_t* idx, const uint64_t* val, size_t rows) -> int {
    c.rows = rows; c.nnz = nnz;
    std::vector<uint32_t> lr;
    for (size_t i = 0; i < rows; ++i) if (p[i + 1] - p[i] > SPMV_LONG) lr.push_back((uint32_t)i);
    c.n_long = lr.size();
    if (c.n_long) { ZCHK(c.long_rows.alloc(lr.size() * 4)); RCHK(hipMemcpy(c.long_rows.p, lr.data(), lr.size() * 4, hipMemcpyHostToDevice)); }
    ZCHK(c.ptr.alloc(p.size() * 4)); ZCHK(c.idx.alloc(nnz * 4)); ZCHK(c.val.alloc(nnz * FRB));
    RCHK(hipMemcpyAsync(c.ptr.p, p.data(), p.size() * 4, hipMemcpyHostToDevice, s));
    if (nnz) {
      DBuf tmp; ZCHK(tmp.alloc(nnz * FRB));
      RCHK(hipMemcpyAsync(c.idx.p, idx, nnz * 4, hipMemcpyHostToDevice, s));
      RCHK(hipMemcpyAsync(tmp.p, val, nnz * FRB, hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(k_to_mont, dim3(nb(nnz)), dim3(256), 0, s, (const uint32_t*)tmp.w(), c.val.w(), nnz);
      RCHK(hipStreamSynchronize(s));
    }
    return ZKT_OK;
  };
  ZCHK(put(rowwise, rp, M->col, M->val, n));
  ZCHK(put(colwise, cp, ridx.data(), tval.data(), cols));
  return ZKT_OK;
}
}  // namespace

struct zkt_groth16_pk {
  size_t n = 0, l = 0, m = 0;
  // the quotient's blocked convolution (k_recip_blocks): this rank's cnt = hiC2 - loC2 values h(n+s), s = qs0 .. qs0+cnt-1, from Q input blocks of Bi, transforms of size M = 2 Bi
  size_t qcnt = 0, qs0 = 1, Bi = 1, M = 2, Q = 1; int logM = 1;
  Csr A, B, Cm;                                  // constraint rows (device), values in Montgomery form
  DBuf cinv, P, ghat, tw, twinv;                  // Fr tables
  zkt_g1_bases *setA = nullptr, *setC1 = nullptr, *setC2 = nullptr; zkt_g2_bases* setB = nullptr;   // the resident base sets (see the file header)
  size_t nA = 0, nC1 = 0, nC2 = 0;
  size_t loA = 0, hiA = 0, loC1 = 0, hiC1 = 0, loC2 = 0, hiC2 = 0;     // this shard's index ranges of the A/B sets and of the two C sets (whole sets when unsharded)
  DBuf cparts;                                   // the Jacobian partials of C1 and C2 of the proof being collected
  size_t shard = 0, nshards = 1;
  // per-proof work buffers.  The MSM scalar vectors (and r, s) are double-buffered: proof k+1's Fr stage may run while the MSMs of proof k
  // are still reading theirs (zkt_groth16_prove_r1cs_submit / _collect); everything else is consumed in stream order before it is rewritten.
  static constexpr int PSLOTS = 2;
  DBuf wires_c, wires_m, z_m[3], X, sA[PSLOTS], sB[PSLOTS], sC[PSLOTS], rs[PSLOTS];       // X[p][q][M]: the block spectra of a, b, c
  bool pending[PSLOTS] = {false, false};
  hipStream_t s = nullptr, sq = nullptr;        // the key's stream (head of the Fr stage, collection) and the quotient stage's own (three transform pairs; high priority)
  hipEvent_t e_head = nullptr, e_q = nullptr;   // (A w), (B w), (C w) and the scalar vectors are ready / the quotient stage has consumed them
  std::recursive_mutex mu;       // calls on one key are serialised (include/zkt.h, Threading)
  ~zkt_groth16_pk() {
    if (sq) { (void)hipStreamSynchronize(sq); hipStreamDestroy(sq); }
    if (e_head) hipEventDestroy(e_head); if (e_q) hipEventDestroy(e_q);
    if (setC1) zkt_g1_bases_free(setC1); if (setC2) zkt_g1_bases_free(setC2); if (setB) zkt_g2_bases_free(setB);
    if (setA) zkt_g1_bases_free(setA);            // last: it owns the streams the others work on
    if (s) hipStreamDestroy(s);
  }
};

extern int zkt_internal_ready();   // zkt_api.cpp
extern "C" int zkt_internal_bases_share_streams(void* dst, void* src, int share_acc, int tail_base, int tail_span);
extern void zkt_internal_set_error_index(size_t i);

extern "C" {

// Multi-GPU form (BASELINE config 4): rank `shard` of `nshards` keeps a contiguous index range of each of the three base
// sets resident and runs the three MSMs on them.  The Fr stage needs no exchange either: the mat-vecs are replicated (75 us), and of the quotient every rank
// evaluates only the h(n+s) whose bases it holds — W forward transforms of size 2n/W per polynomial and ONE inverse instead of a forward and an inverse of
// size 2n (k_recip_blocks).  The only exchange is an all_gather of the three Jacobian partials followed by zkt_g{1,2}_jac_sum_dev.
int zkt_groth16_setup_r1cs_sharded(size_t n, size_t l, size_t m, const zkt_sparse_rows* A, const zkt_sparse_rows* B, const zkt_sparse_rows* Cmat,
                                   const uint64_t* alpha, const uint64_t* beta, const uint64_t* gamma, const uint64_t* delta, const uint64_t* x,
                                   size_t shard, size_t nshards, zkt_groth16_crs* vk, zkt_groth16_pk** out) {
  if (zkt_internal_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (nshards == 0 || shard >= nshards || nshards > n) return ZKT_ERR_SHAPE;
  if (!A || !B || !Cmat || !alpha || !beta || !gamma || !delta || !x || !vk || !out || n == 0 || l > m || n >= (1ull << 30)) return ZKT_ERR_SHAPE;
  if (!A->rowptr || !B->rowptr || !Cmat->rowptr) return ZKT_ERR_SHAPE;
  uint64_t trap[20]; memcpy(trap, alpha, 32); memcpy(trap + 4, beta, 32); memcpy(trap + 8, gamma, 32); memcpy(trap + 12, delta, 32); memcpy(trap + 16, x, 32);
  for (int k = 0; k < 5; ++k) { bool z = true; for (int j = 0; j < 4; ++j) z = z && trap[4 * k + j] == 0; if (z) return ZKT_ERR_INV_ZERO; }   // rand_elem(true): non-zero (crs.rs:59-63)
  std::unique_ptr<zkt_groth16_pk> pk(new zkt_groth16_pk);
  pk->n = n; pk->l = l; pk->m = m;
  const size_t rows = m + 1;
  const size_t nw = m - l, nh = n >= 2 ? n - 1 : 0, nA = n + 2, nC1 = n + nw + 3, nC2 = nh, nC = nC1 + nC2;
  pk->nA = nA; pk->nC1 = nC1; pk->nC2 = nC2; pk->shard = shard; pk->nshards = nshards;
  auto range = [&](size_t tot, size_t& lo, size_t& hi) { size_t base = tot / nshards, extra = tot % nshards; lo = shard * base + (shard < extra ? shard : extra); hi = lo + base + (shard < extra ? 1 : 0); };
  range(nA, pk->loA, pk->hiA); range(nC1, pk->loC1, pk->hiC1); range(nC2, pk->loC2, pk->hiC2);
  pk->qcnt =pk->hiC2 - pk->loC2; pk->qs0 = pk->loC2 + 1;
  int logM = 1; while (((size_t)1 << (logM - 1)) < pk->qcnt || (logM <= 10 && ((size_t)1 << (logM - 1)) < n)) ++logM;      // Bi >= cnt; not below min(n, 1024) (many ranks on a small circuit)
  const size_t M = (size_t)1 << logM, Bi = M / 2, Q = pk->qcnt ? (n + Bi - 1) / Bi : 1, QM = Q * M;
  pk->logM = logM; pk->M = M; pk->Bi = Bi; pk->Q = Q;
  { int lo = 0, hi = 0; RCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    RCHK(hipStreamCreateWithFlags(&pk->s, hipStreamNonBlocking)); RCHK(hipStreamCreateWithPriority(&pk->sq, hipStreamNonBlocking, hi));
    RCHK(hipEventCreateWithFlags(&pk->e_head, hipEventDisableTiming)); RCHK(hipEventCreateWithFlags(&pk->e_q, hipEventDisableTiming)); }
  hipStream_t s = pk->s;
  Csr At, Bt, Ct;
  ZCHK(upload_csr(A, n, rows, pk->A, At, s)); ZCHK(upload_csr(B, n, rows, pk->B, Bt, s)); ZCHK(upload_csr(Cmat, n, rows, pk->Cm, Ct, s));

  // ---- Fr tables ----
  DBuf dtrap, consts, fact, invfact, xm, pre, xinv, Lm, Lc, hbc, derr;
  ZCHK(dtrap.alloc(160)); ZCHK(consts.alloc(K_COUNT * FRB)); ZCHK(fact.alloc((2 * n + 1) * FRB)); ZCHK(invfact.alloc((2 * n + 1) * FRB));
  ZCHK(xm.alloc((2 * n) * FRB)); ZCHK(pre.alloc((2 * n) * FRB)); ZCHK(xinv.alloc((2 * n) * FRB)); ZCHK(Lm.alloc(n * FRB)); ZCHK(Lc.alloc(n * FRB));
  ZCHK(hbc.alloc(n * FRB)); ZCHK(derr.alloc(8));
  ZCHK(pk->cinv.alloc(n * FRB)); ZCHK(pk->P.alloc(n * FRB)); ZCHK(pk->ghat.alloc(QM * FRB)); ZCHK(pk->tw.alloc(M / 2 * FRB)); ZCHK(pk->twinv.alloc(M / 2 * FRB));
  unsigned long long noerr = NO_ERR;
  RCHK(hipMemcpyAsync(derr.p, &noerr, 8, hipMemcpyHostToDevice, s));
  RCHK(hipMemcpyAsync(dtrap.p, trap, 160, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_setup_consts, dim3(1), dim3(64), 0, s, (const uint32_t*)dtrap.w(), consts.w(), logM);
  hipLaunchKernelGGL(k_iota, dim3(nb(2 * n + 1)), dim3(256), 0, s, fact.w(), 2 * n + 1);
  ZCHK(scan_mul(fact.w(), fact.w(), 2 * n + 1, s));
  hipLaunchKernelGGL(k_inv, dim3(nb(2 * n + 1)), dim3(256), 0, s, (const uint32_t*)fact.w(), invfact.w(), 2 * n + 1, (unsigned long long*)derr.p);
  const size_t nx = 2 * n - 1;
  hipLaunchKernelGGL(k_x_minus, dim3(nb(nx)), dim3(256), 0, s, (const uint32_t*)(consts.w() + K_X * FW), xm.w(), nx);
  ZCHK(scan_mul(xm.w(), pre.w(), nx, s));
  hipLaunchKernelGGL(k_inv, dim3(nb(nx)), dim3(256), 0, s, (const uint32_t*)xm.w(), xinv.w(), nx, (unsigned long long*)derr.p);
  unsigned long long e = NO_ERR;
  RCHK(hipMemcpyAsync(&e, derr.p, 8, hipMemcpyDeviceToHost, s)); RCHK(hipStreamSynchronize(s));
  if (e != NO_ERR) { zkt_internal_set_error_index((size_t)e); return ZKT_ERR_INV_ZERO; }      // x fell on the domain {1..2n-1}: t(x) = 0, no CRS
  hipLaunchKernelGGL(k_setup_consts2, dim3(1), dim3(64), 0, s, (const uint32_t*)pre.w(), n, consts.w());
  hipLaunchKernelGGL(k_lagrange, dim3(nb(n)), dim3(256), 0, s, (const uint32_t*)consts.w(), (const uint32_t*)invfact.w(), (const uint32_t*)xinv.w(), n, pk->cinv.w(), Lm.w(), Lc.w());
  if (n >= 2) hipLaunchKernelGGL(k_hbasis, dim3(nb(n - 1)), dim3(256), 0, s, (const uint32_t*)consts.w(), (const uint32_t*)fact.w(), (const uint32_t*)invfact.w(), (const uint32_t*)xinv.w(), n, hbc.w(), pk->P.w());
  // twiddles: w^k and w^-k, k < M/2, as prefix products
  hipLaunchKernelGGL(k_fill_pow, dim3(nb(M / 2)), dim3(256), 0, s, (const uint32_t*)(consts.w() + K_OMEGA * FW), pk->tw.w(), M / 2);
  ZCHK(scan_mul(pk->tw.w(), pk->tw.w(), M / 2, s));
  hipLaunchKernelGGL(k_fill_pow, dim3(nb(M / 2)), dim3(256), 0, s, (const uint32_t*)(consts.w() + K_OMEGA_INV * FW), pk->twinv.w(), M / 2);
  ZCHK(scan_mul(pk->twinv.w(), pk->twinv.w(), M / 2, s));
  // the Q kernel slices of this rank and their spectra (with the 1/M of the inverse transform)
  hipLaunchKernelGGL(k_recip_blocks, dim3(nb(QM)), dim3(256), 0, s, (const uint32_t*)fact.w(), (const uint32_t*)invfact.w(), n, pk->qs0, pk->qcnt, Bi, Q, pk->ghat.w());
  ZCHK(ntt_forward(pk->ghat.w(), logM, pk->tw.w(), nullptr, s, Q));
  hipLaunchKernelGGL(k_scale_all, dim3(nb(QM)), dim3(256), 0, s, pk->ghat.w(), (const uint32_t*)(consts.w() + K_NINV * FW), QM);

  // ---- per-wire evaluations u_i(x) = sum_j A[j][i] L_j(x)  and the scalars of crs.rs:66-84 ----
  DBuf ue, ve, we, y; ZCHK(ue.alloc(rows * FRB)); ZCHK(ve.alloc(rows * FRB)); ZCHK(we.alloc(rows * FRB)); ZCHK(y.alloc(rows * FRB));
  Csr* T[3] = {&At, &Bt, &Ct}; DBuf* ev[3] = {&ue, &ve, &we};
  for (int k = 0; k < 3; ++k) spmv(*T[k], Lm.w(), ev[k]->w(), s);
  hipLaunchKernelGGL(k_uvw, dim3(nb(rows)), dim3(256), 0, s, (const uint32_t*)consts.w(), (const uint32_t*)ue.w(), (const uint32_t*)ve.w(), (const uint32_t*)we.w(), l, rows, y.w());
  RCHK(hipGetLastError());

  // ---- group side: fixed-base multiplications of the generators (crs.rs:85-135), written straight into the three base sets ----
  DBuf gen1, gen2, pU, pA, pB, pC, small1, small2, gt;
  ZCHK(gen1.alloc(G1B)); ZCHK(gen2.alloc(G2B)); ZCHK(pU.alloc(rows * G1B)); ZCHK(pA.alloc(nA * G1B)); ZCHK(pB.alloc(nA * G2B)); ZCHK(pC.alloc(nC * G1B));
  ZCHK(small1.alloc(3 * G1B)); ZCHK(small2.alloc(3 * G2B)); ZCHK(gt.alloc(576));
  RCHK(hipMemcpyAsync(gen1.p, G1_GEN, G1B, hipMemcpyHostToDevice, s)); RCHK(hipMemcpyAsync(gen2.p, G2_GEN, G2B, hipMemcpyHostToDevice, s));
  RCHK(launch_generator_mul(G_G1, gen1.w(), dtrap.w(), small1.w(), 1, s));               // alpha
  RCHK(launch_generator_mul(G_G1, gen1.w(), dtrap.w() + 8, small1.w() + 26, 1, s));      // beta
  RCHK(launch_generator_mul(G_G1, gen1.w(), dtrap.w() + 24, small1.w() + 52, 1, s));     // delta
  RCHK(launch_generator_mul(G_G2, gen2.w(), dtrap.w() + 8, small2.w(), 1, s));           // beta
  RCHK(launch_generator_mul(G_G2, gen2.w(), dtrap.w() + 16, small2.w() + 50, 1, s));     // gamma
  RCHK(launch_generator_mul(G_G2, gen2.w(), dtrap.w() + 24, small2.w() + 100, 1, s));    // delta
  // uvw: statement part to the verifying key, witness part into the C1 set.  pC = [L (n) | uvw_wit (nw) | alpha, beta, delta || Lambda t/delta (nh)]
  RCHK(launch_generator_mul(G_G1, gen1.w(), y.w(), pU.w(), rows, s));
  RCHK(hipMemcpyAsync(vk->g1_uvw_stmt, pU.p, (l + 1) * G1B, hipMemcpyDeviceToHost, s));
  if (vk->g1_uvw_wit && nw) RCHK(hipMemcpyAsync(vk->g1_uvw_wit, pU.w() + (l + 1) * 26, nw * G1B, hipMemcpyDeviceToHost, s));
  if (nw) RCHK(hipMemcpyAsync(pC.w() + n * 26, pU.w() + (l + 1) * 26, nw * G1B, hipMemcpyDeviceToDevice, s));
  RCHK(launch_generator_mul(G_G1, gen1.w(), Lc.w(), pC.w(), n, s));                       // [L_j(x)]_1
  RCHK(hipMemcpyAsync(pA.p, pC.p, n * G1B, hipMemcpyDeviceToDevice, s));
  if (nh) RCHK(launch_generator_mul(G_G1, gen1.w(), hbc.w(), pC.w() + nC1 * 26, nh, s));        // [Lambda_s(x) t(x)/delta]_1
  RCHK(launch_generator_mul(G_G2, gen2.w(), Lc.w(), pB.w(), n, s));                       // [L_j(x)]_2
  RCHK(hipMemcpyAsync(pA.w() + n * 26, small1.p, G1B, hipMemcpyDeviceToDevice, s));            // A tail: alpha, delta
  RCHK(hipMemcpyAsync(pA.w() + (n + 1) * 26, small1.w() + 52, G1B, hipMemcpyDeviceToDevice, s));
  RCHK(hipMemcpyAsync(pB.w() + n * 50, small2.p, G2B, hipMemcpyDeviceToDevice, s));            // B tail: beta, delta
  RCHK(hipMemcpyAsync(pB.w() + (n + 1) * 50, small2.w() + 100, G2B, hipMemcpyDeviceToDevice, s));
  RCHK(hipMemcpyAsync(pC.w() + (n + nw) * 26, small1.p, 3 * G1B, hipMemcpyDeviceToDevice, s));         // C1 tail: alpha, beta, delta
  RCHK(hipStreamSynchronize(s));
  pU.release();
  ZCHK(zkt_g1_bases_from_device((const zkt_g1_affine*)(pA.w() + pk->loA * 26), pk->hiA - pk->loA, s, &pk->setA)); pA.release();
  ZCHK(zkt_g2_bases_from_device((const zkt_g2_affine*)(pB.w() + pk->loA * 50), pk->hiA - pk->loA, s, &pk->setB)); pB.release();
  ZCHK(zkt_g1_bases_from_device((const zkt_g1_affine*)(pC.w() + pk->loC1 * 26), pk->hiC1 - pk->loC1, s, &pk->setC1));
  ZCHK(zkt_g1_bases_from_device((const zkt_g1_affine*)(pC.w() + (nC1 + pk->loC2) * 26), pk->hiC2 - pk->loC2, s, &pk->setC2)); pC.release();
  // one set of streams for the key (owner: setA): a sort stream, the G1 accumulate stream (C1, A, C2 in turn), B's own accumulate stream, four reduce streams —
  // with the key's own stream that is eight, one per hardware queue
  // Sets below 2^19 terms (small circuits, and every shard of a proof spread over several GPUs) run their whole MSM on ONE stream each (zkt_api.cpp, msm_submit_locked): there
  // A, C1, C2 and B get a reduce stream each, so the four sums of a proof run side by side instead of one after the other (a rank's share of a 2^20-constraint proof
  // over 8 GPUs)
  const bool side_by_side = std::max(std::max(pk->hiA - pk->loA, pk->hiC1 - pk->loC1), pk->hiC2 - pk->loC2) < ((size_t)1 << 19);
  if (side_by_side) {
    // reduce stream of (set, proof slot) = (set index + slot) mod 4 with A = 0 (the owner: streams 0, 1), C1 = 1, B = 2, C2 = 3: the four sums of ONE proof are on four
    // streams, and with two proofs in flight no two slots of a set meet on one stream (a fixed stream per set cost 23 % of the pipelined rate at 2^16 constraints)
    ZCHK(zkt_internal_bases_share_streams(pk->setC1, pk->setA, 1, 1, 2)); ZCHK(zkt_internal_bases_share_streams(pk->setC2, pk->setA, 1, 3, 2));
    ZCHK(zkt_internal_bases_share_streams(pk->setB, pk->setA, 0, 2, 2));
  } else {
    ZCHK(zkt_internal_bases_share_streams(pk->setC1, pk->setA, 1, 0, 2)); ZCHK(zkt_internal_bases_share_streams(pk->setC2, pk->setA, 1, 0, 2));
    ZCHK(zkt_internal_bases_share_streams(pk->setB, pk->setA, 0, 2, 2));
  }
  RCHK(hipMemcpyAsync(derr.p, &noerr, 8, hipMemcpyHostToDevice, s));
  RCHK(launch_tate(small1.w(), small2.w(), gt.w(), 1, (unsigned long long*)derr.p, s));        // crs.rs:137-139
  RCHK(hipMemcpyAsync(vk->g1_alpha, small1.p, G1B, hipMemcpyDeviceToHost, s)); RCHK(hipMemcpyAsync(vk->g1_beta, small1.w() + 26, G1B, hipMemcpyDeviceToHost, s));
  RCHK(hipMemcpyAsync(vk->g1_delta, small1.w() + 52, G1B, hipMemcpyDeviceToHost, s)); RCHK(hipMemcpyAsync(vk->g2_beta, small2.p, G2B, hipMemcpyDeviceToHost, s));
  RCHK(hipMemcpyAsync(vk->g2_gamma, small2.w() + 50, G2B, hipMemcpyDeviceToHost, s)); RCHK(hipMemcpyAsync(vk->g2_delta, small2.w() + 100, G2B, hipMemcpyDeviceToHost, s));
  RCHK(hipMemcpyAsync(vk->gt_alpha_beta, gt.p, 576, hipMemcpyDeviceToHost, s));
  RCHK(hipStreamSynchronize(s));
  vk->n = n; vk->l = l; vk->m = m;

  // ---- per-proof work buffers ----
  ZCHK(pk->wires_c.alloc(rows * FRB)); ZCHK(pk->wires_m.alloc(rows * FRB));
  for (int k = 0; k < zkt_groth16_pk::PSLOTS; ++k) { ZCHK(pk->rs[k].alloc(2 * FRB)); ZCHK(pk->sA[k].alloc(nA * FRB)); ZCHK(pk->sB[k].alloc(nA * FRB)); ZCHK(pk->sC[k].alloc(nC * FRB)); }
  ZCHK(pk->cparts.alloc(2 * ZKT_G1_PARTIAL_WORDS * 4));
  for (int k = 0; k < 3; ++k) ZCHK(pk->z_m[k].alloc(n * FRB));
  ZCHK(pk->X.alloc(3 * QM * FRB));
  *out = pk.release();
  return ZKT_OK;
}

int zkt_groth16_setup_r1cs(size_t n, size_t l, size_t m, const zkt_sparse_rows* A, const zkt_sparse_rows* B, const zkt_sparse_rows* Cmat,
                           const uint64_t* alpha, const uint64_t* beta, const uint64_t* gamma, const uint64_t* delta, const uint64_t* x,
                           zkt_groth16_crs* vk, zkt_groth16_pk** out) {
  return zkt_groth16_setup_r1cs_sharded(n, l, m, A, B, Cmat, alpha, beta, gamma, delta, x, 0, 1, vk, out);
}
void zkt_groth16_pk_free(zkt_groth16_pk* pk) { delete pk; }

// Prover::prove (prover.rs:96-147) with r, s injected; wires = a_0..a_m canonical, on the host or (…_dev) already in HBM.
// enqueue one proof: the Fr stage on the key's stream, the three MSMs on their base sets' pipelines (MSM slot = proof slot)
static int prove_submit(zkt_groth16_pk* pk, int ps, const uint64_t* wires, bool wires_on_device, const uint64_t* r, const uint64_t* s_) {
  if (zkt_internal_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (!pk) return ZKT_ERR_SHAPE;
  std::lock_guard<std::recursive_mutex> lk(pk->mu);
  if (!wires || !r || !s_ || ps < 0 || ps >= zkt_groth16_pk::PSLOTS || pk->pending[ps]) return ZKT_ERR_SHAPE;
  const size_t n = pk->n, l = pk->l, m = pk->m, rows = m + 1, nw = m - l;
  hipStream_t s = pk->s;
  DBuf &sA = pk->sA[ps], &sB = pk->sB[ps], &sC = pk->sC[ps], &drs = pk->rs[ps];
  uint64_t rs[8]; memcpy(rs, r, 32); memcpy(rs + 4, s_, 32);
  RCHK(hipStreamWaitEvent(s, pk->e_q, 0));        // z_m and X are shared by the proof slots: the previous proof's quotient stage has to be through with them
  RCHK(hipMemcpyAsync(drs.p, rs, 64, hipMemcpyHostToDevice, s));
  RCHK(hipMemcpyAsync(pk->wires_c.p, wires, rows * FRB, wires_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_to_mont, dim3(nb(rows)), dim3(256), 0, s, (const uint32_t*)pk->wires_c.w(), pk->wires_m.w(), rows);
  Csr* M[3] = {&pk->A, &pk->B, &pk->Cm};
  for (int k = 0; k < 3; ++k) spmv(*M[k], pk->wires_m.w(), pk->z_m[k].w(), s);
  hipLaunchKernelGGL(k_prove_scalars, dim3(nb(n + nw + 1)), dim3(256), 0, s, (const uint32_t*)pk->z_m[0].w(), (const uint32_t*)pk->z_m[1].w(), (const uint32_t*)pk->wires_c.w(),
                     (const uint32_t*)drs.w(), n, l, m, sA.w(), sB.w(), sC.w());
  RCHK(hipGetLastError());
  // The quotient stage is enqueued FIRST, on a stream of its own: its launches cost the host ~0.1 ms, an MSM submission 0.5-0.7 ms, and the quotient's MSM is the end of the
  // critical path (on a shard of a proof the chain used to start 2.4 ms into the proof, behind three submissions; on one GPU it delayed the collection of the proof before).
  // A, B and the first part of C only need (A w), (B w) and the wires: their input event is recorded on `s`, which the chain is not on.
  const size_t nC1 = pk->nC1;
  hipStream_t q = s;
  if (pk->qcnt) {                       // (an empty range of the quotient's bases — n = 1, or more ranks than bases — has nothing to evaluate)
    q = pk->sq;
    const size_t M = pk->M, Q = pk->Q, QM = Q * M;
    RCHK(hipEventRecord(pk->e_head, s)); RCHK(hipStreamWaitEvent(q, pk->e_head, 0));
    // a, b, c side by side (grid.y): 2 + 2 k launches for transforms of k passes instead of 3 (1 + 2 k)
    hipLaunchKernelGGL(k_prep_blocks, dim3(nb(QM), 3), dim3(256), 0, q, (const uint32_t*)pk->z_m[0].w(), (const uint32_t*)pk->z_m[1].w(), (const uint32_t*)pk->z_m[2].w(),
                       (const uint32_t*)pk->cinv.w(), n, pk->Bi, Q, pk->X.w());
    ZCHK(ntt_forward(pk->X.w(), pk->logM, pk->tw.w(), pk->ghat.w(), q, Q, 3, QM));     // block spectrum * spectrum of its slice of 1/d (and 1/M)
    if (Q > 1) hipLaunchKernelGGL(k_sum_blocks, dim3(nb(M), 3), dim3(256), 0, q, pk->X.w(), M, Q);
    ZCHK(ntt_inverse(pk->X.w(), pk->logM, pk->twinv.w(), q, 1, 3, QM));
    hipLaunchKernelGGL(k_hvals, dim3(nb(pk->qcnt)), dim3(256), 0, q, (const uint32_t*)pk->X.w(), QM, (const uint32_t*)pk->P.w(), pk->qs0, pk->qcnt, sC.w() + nC1 * FW);
    RCHK(hipGetLastError());
    RCHK(hipEventRecord(pk->e_q, q));
  }
  // the sums, longest first; the quotient part of C is the only one that waits for the NTT chain (its input event is recorded on `s` behind k_hvals)
  hipStream_t in = s;          // (waiting for the chain instead — all four sums behind it — was measured on a shard of 8: 8.2 ms against 7.6)
  ZCHK(zkt_g2_msm_submit(pk->setB, (const uint64_t*)(sB.w() + pk->loA * FW), pk->hiA - pk->loA, in, ps));
  ZCHK(zkt_g1_msm_submit(pk->setC1, (const uint64_t*)(sC.w() + pk->loC1 * FW), pk->hiC1 - pk->loC1, in, ps));
  ZCHK(zkt_g1_msm_submit(pk->setC2, (const uint64_t*)(sC.w() + (nC1 + pk->loC2) * FW), pk->hiC2 - pk->loC2, q, ps));
  ZCHK(zkt_g1_msm_submit(pk->setA, (const uint64_t*)(sA.w() + pk->loA * FW), pk->hiA - pk->loA, in, ps));
  pk->pending[ps] = true;
  return ZKT_OK;
}
static int prove_collect(zkt_groth16_pk* pk, int ps, zkt_g1_affine* A, zkt_g2_affine* B, zkt_g1_affine* Cp, uint32_t* dev_partials) {
  if (!pk) return ZKT_ERR_SHAPE;
  std::lock_guard<std::recursive_mutex> lk(pk->mu);
  if (ps < 0 || ps >= zkt_groth16_pk::PSLOTS || !pk->pending[ps]) return ZKT_ERR_SHAPE;
  if (dev_partials ? false : (!A || !B || !Cp || pk->nshards != 1)) return ZKT_ERR_SHAPE;      // a shard can only produce partials
  pk->pending[ps] = false;
  // C = C1 + C2: both Jacobian partials side by side in cparts, one addition on the device
  uint32_t* cp = pk->cparts.w();
  ZCHK(zkt_g1_msm_collect(pk->setC1, ps, nullptr, cp)); ZCHK(zkt_g1_msm_collect(pk->setC2, ps, nullptr, cp + ZKT_G1_PARTIAL_WORDS));
  if (dev_partials) {            // [A: ZKT_G1_PARTIAL_WORDS | B: ZKT_G2_PARTIAL_WORDS | C: ZKT_G1_PARTIAL_WORDS]
    ZCHK(zkt_g1_msm_collect(pk->setA, ps, nullptr, dev_partials)); ZCHK(zkt_g2_msm_collect(pk->setB, ps, nullptr, dev_partials + ZKT_G1_PARTIAL_WORDS));
    RCHK(launch_msm_jac_add(G_G1, cp, cp + ZKT_G1_PARTIAL_WORDS, dev_partials + ZKT_G1_PARTIAL_WORDS + ZKT_G2_PARTIAL_WORDS, pk->s));
    RCHK(hipStreamSynchronize(pk->s));
    return ZKT_OK;
  }
  ZCHK(zkt_g1_msm_collect(pk->setA, ps, A, nullptr)); ZCHK(zkt_g2_msm_collect(pk->setB, ps, B, nullptr));
  ZCHK(zkt_g1_jac_sum_dev(cp, 2, pk->s, Cp));
  return ZKT_OK;
}
// Prover::prove (prover.rs:96-147) with r, s injected; wires = a_0..a_m canonical, on the host or (…_dev) already in HBM.
static int prove_impl(zkt_groth16_pk* pk, const uint64_t* wires, bool wires_on_device, const uint64_t* r, const uint64_t* s_, zkt_g1_affine* A, zkt_g2_affine* B, zkt_g1_affine* Cp,
                      uint32_t* dev_partials = nullptr) {
  if (!pk) return ZKT_ERR_SHAPE;
  std::lock_guard<std::recursive_mutex> lk(pk->mu);
  if (dev_partials ? false : (!A || !B || !Cp || pk->nshards != 1)) return ZKT_ERR_SHAPE;
  ZCHK(prove_submit(pk, 0, wires, wires_on_device, r, s_));
  return prove_collect(pk, 0, A, B, Cp, dev_partials);
}
int zkt_groth16_prove_r1cs(zkt_groth16_pk* pk, const uint64_t* wires, const uint64_t* r, const uint64_t* s_, zkt_g1_affine* A, zkt_g2_affine* B, zkt_g1_affine* Cp) {
  return prove_impl(pk, wires, false, r, s_, A, B, Cp);
}
// Pipelined form: two proofs in flight on one key — the Fr stage of proof k+1 runs under the MSMs of proof k.  slot in {0, 1}; a slot
// must be collected before it is submitted again.  (Single-GPU keys only; wires already in HBM.)
int zkt_groth16_prove_r1cs_submit(zkt_groth16_pk* pk, int slot, const uint64_t* dev_wires, const uint64_t* r, const uint64_t* s_) {
  if (pk && pk->nshards != 1) return ZKT_ERR_SHAPE;
  return prove_submit(pk, slot, dev_wires, true, r, s_);
}
int zkt_groth16_prove_r1cs_collect(zkt_groth16_pk* pk, int slot, zkt_g1_affine* A, zkt_g2_affine* B, zkt_g1_affine* Cp) {
  return prove_collect(pk, slot, A, B, Cp, nullptr);
}
// one shard's share of a proof: the three un-normalised Jacobian partial sums, on the device
int zkt_groth16_prove_r1cs_partials(zkt_groth16_pk* pk, const uint64_t* dev_wires, const uint64_t* r, const uint64_t* s_, uint32_t* dev_partials) {
  if (!dev_partials) return ZKT_ERR_SHAPE;
  return prove_impl(pk, dev_wires, true, r, s_, nullptr, nullptr, nullptr, dev_partials);
}
int zkt_groth16_prove_r1cs_dev(zkt_groth16_pk* pk, const uint64_t* dev_wires, const uint64_t* r, const uint64_t* s_, zkt_g1_affine* A, zkt_g2_affine* B, zkt_g1_affine* Cp) {
  return prove_impl(pk, dev_wires, true, r, s_, A, B, Cp);
}

}  // extern "C"
