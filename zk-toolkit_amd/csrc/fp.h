// Prime-field arithmetic for gfx950 (CDNA4): N x 32-bit limbs, Montgomery form
// internally, canonical residues in [0,p) at every function boundary.
//
// Replaces the reference's PrimeFieldElem (BigUint residue, `*` then `%`):
//   src/building_block/field/prime_field_elem.rs:263-457   (reference paths are
//   relative to /root/reference/).
//
// Measured on MI355X (build/exp/ubench, profiles/r01_valu_ubench.txt): v_mad_u64_u32,
// v_mul_lo/hi_u32 and v_addc_co_u32 all issue at the plain VALU rate (one wave64
// instruction per 4 clk per SIMD, ~33 T lane-ops/s chip-wide), so the cost of a
// field multiply is its *instruction count*.  The multiply is therefore a finely
// integrated product-scanning (Comba) Montgomery: every 32x32 product is one
// v_mad_u64_u32 into a 64-bit column accumulator plus one v_addc_co_u32 for the
// column's third word: 2*N*N + N MADs, no shifts of partial rows, the modulus
// limbs live in SGPRs.  Add/sub are carry chains (__builtin_addc/subc lower to
// v_add_co/v_addc_co on this compiler).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "zkt_constants.h"

namespace zkt {

// One field element = one ext-vector of N dwords: a single-member struct of a
// vector type is passed and returned in VGPRs by the AMDGPU calling convention
// (an array member would go through scratch), which is what lets fp_mul be a
// real (non-inlined) function without touching memory.
template <class C>
struct Fp {
  typedef uint32_t vec_t __attribute__((ext_vector_type(C::N)));
  vec_t v;
};

// acc(lo:64,hi:32) += a*b
ZKT_HD void mac(uint64_t& lo, uint32_t& hi, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
#else
  uint64_t p = (uint64_t)a * b; lo += p; hi += (lo < p);
#endif
}
// same, b is a compile-time constant kept in an SGPR
ZKT_HD void mac_k(uint64_t& lo, uint32_t& hi, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "s"(b) : "vcc");
#else
  mac(lo, hi, a, b);
#endif
}
ZKT_HD uint32_t addc(uint32_t a, uint32_t b, uint32_t& c) { unsigned co; uint32_t r = __builtin_addc(a, b, c, &co); c = co; return r; }
ZKT_HD uint32_t subb(uint32_t a, uint32_t b, uint32_t& c) { unsigned co; uint32_t r = __builtin_subc(a, b, c, &co); c = co; return r; }

template <class C> ZKT_HD Fp<C> fp_zero() { Fp<C> r;
#pragma unroll
  for (int i = 0; i < C::N; ++i) r.v[i] = 0; return r; }
template <class C> ZKT_HD Fp<C> fp_one() { Fp<C> r;   // Montgomery 1 = R mod p
#pragma unroll
  for (int i = 0; i < C::N; ++i) r.v[i] = C::one(i); return r; }
template <class C> ZKT_HD bool fp_is_zero(const Fp<C>& a) { uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < C::N; ++i) o |= a.v[i]; return o == 0; }
template <class C> ZKT_HD bool fp_eq(const Fp<C>& a, const Fp<C>& b) { uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < C::N; ++i) o |= a.v[i] ^ b.v[i]; return o == 0; }

// t (N limbs + carry word `top`) -> t - p if t >= p.  Requires t < 2p.
template <class C> ZKT_HD void fp_cond_sub(Fp<C>& r, uint32_t top) {
  uint32_t s[C::N]; uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < C::N; ++i) s[i] = subb(r.v[i], C::mod(i), bw);
  bool keep = (top == 0) && bw;   // t < p
#pragma unroll
  for (int i = 0; i < C::N; ++i) r.v[i] = keep ? r.v[i] : s[i];
}

// plus (prime_field_elem.rs:278-286)
template <class C> ZKT_HD Fp<C> fp_add(const Fp<C>& a, const Fp<C>& b) {
  Fp<C> r; uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < C::N; ++i) r.v[i] = addc(a.v[i], b.v[i], c);
  fp_cond_sub(r, c);
  return r;
}
// minus (prime_field_elem.rs:288-300): a<b -> p-(b-a)
template <class C> ZKT_HD Fp<C> fp_sub(const Fp<C>& a, const Fp<C>& b) {
  Fp<C> r; uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < C::N; ++i) r.v[i] = subb(a.v[i], b.v[i], bw);
  uint32_t mask = 0u - bw, c = 0;
#pragma unroll
  for (int i = 0; i < C::N; ++i) r.v[i] = addc(r.v[i], C::mod(i) & mask, c);
  return r;
}
// negate (prime_field_elem.rs:448-457): 0 stays 0
template <class C> ZKT_HD Fp<C> fp_neg(const Fp<C>& a) {
  Fp<C> r; uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < C::N; ++i) r.v[i] = subb(C::mod(i), a.v[i], bw);
  uint32_t mask = fp_is_zero(a) ? 0u : 0xffffffffu;
#pragma unroll
  for (int i = 0; i < C::N; ++i) r.v[i] &= mask;
  return r;
}
template <class C> ZKT_HD Fp<C> fp_dbl(const Fp<C>& a) { return fp_add(a, a); }

// Montgomery product a*b*R^-1 mod p, inputs and output in [0,p).
// times (prime_field_elem.rs:302-308) in the Montgomery domain.
template <class C> ZKT_HD Fp<C> fp_mul_impl(const Fp<C>& a, const Fp<C>& b) {
  constexpr int N = C::N;
  Fp<C> r; uint32_t m[N];
  uint64_t lo = 0; uint32_t hi = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) mac(lo, hi, a.v[i], b.v[k - i]);
#pragma unroll
    for (int j = 0; j < k; ++j) mac_k(lo, hi, m[j], C::mod(k - j));
    m[k] = (uint32_t)lo * C::INV;
    mac_k(lo, hi, m[k], C::mod(0));
    lo = (lo >> 32) | ((uint64_t)hi << 32); hi = 0;
  }
#pragma unroll
  for (int k = N; k < 2 * N; ++k) {
#pragma unroll
    for (int i = k - N + 1; i < N; ++i) mac(lo, hi, a.v[i], b.v[k - i]);
#pragma unroll
    for (int j = k - N + 1; j < N; ++j) mac_k(lo, hi, m[j], C::mod(k - j));
    r.v[k - N] = (uint32_t)lo;
    lo = (lo >> 32) | ((uint64_t)hi << 32); hi = 0;
  }
  fp_cond_sub(r, (uint32_t)lo);
  return r;
}

// Call policy.  One inlined multiply is ~1000 instructions (8 KB); curve and
// pairing kernels contain hundreds of them, far beyond the 64 KB instruction
// cache, so by default the multiply is ONE function per field and kernel image,
// called with both operands in VGPRs (IPRA keeps the caller's live values out of
// the callee's clobber set, so nothing is spilled around the call).  Small
// kernels define ZKT_INLINE_MUL before including this header.
#if !defined(ZKT_INLINE_MUL)
template <class C> ZKT_FN Fp<C> fp_mul(Fp<C> a, Fp<C> b) { return fp_mul_impl(a, b); }
#else
template <class C> ZKT_HD Fp<C> fp_mul(const Fp<C>& a, const Fp<C>& b) { return fp_mul_impl(a, b); }
#endif

// Montgomery square.  sq (prime_field_elem.rs:330-335).  A dedicated squaring
// (cross products once, doubled per column) saves 66 of the 300 MAD pairs but
// pays ~6 shift/add ops per column for the 96-bit doubling — no net win while
// v_mad_u64_u32 issues at the plain VALU rate, so it is the product.
template <class C> ZKT_HD Fp<C> fp_sqr(const Fp<C>& a) { return fp_mul(a, a); }

// canonical <-> Montgomery
template <class C> ZKT_HD Fp<C> fp_to_mont(const Fp<C>& a) {
  Fp<C> r2;
#pragma unroll
  for (int i = 0; i < C::N; ++i) r2.v[i] = C::r2(i);
  return fp_mul(a, r2);
}
template <class C> ZKT_HD Fp<C> fp_from_mont(const Fp<C>& a) {
  Fp<C> one = fp_zero<C>(); one.v[0] = 1;
  return fp_mul(a, one);
}
// is a canonical input really < p ?  (ABI contract check)
template <class C> ZKT_HD bool fp_is_canonical(const Fp<C>& a) {
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < C::N; ++i) (void)subb(a.v[i], C::mod(i), bw);
  return bw != 0;
}

// a^(p-2): inverse of a non-zero element (Montgomery domain in and out).
// safe_inv (prime_field_elem.rs:379-432) returns the unique inverse in [0,p);
// Fermat gives the same residue.  The exponent is a compile-time constant, so
// the branch is wave-uniform.
template <class C> ZKT_FN Fp<C> fp_inv_fermat(Fp<C> a) {
  Fp<C> r = fp_one<C>();
  bool started = false;
  for (int i = C::N * 32 - 1; i >= 0; --i) {
    uint32_t w = 0;
    // constant table lookup with a run-time index: select via unrolled compare
#pragma unroll
    for (int j = 0; j < C::N; ++j) w = (j == (i >> 5)) ? C::pm2(j) : w;
    bool bit = (w >> (i & 31)) & 1;
    if (started) r = fp_sqr(r);
    if (bit) { r = started ? fp_mul(r, a) : a; started = true; }
  }
  return r;
}

// Inverse by the binary extended Euclid on the plain integers (odd p): ~2*bits iterations of
// shifts and carry-chain adds instead of ~1.5*bits Montgomery products — about 4x cheaper than
// fp_inv_fermat on this machine, and it is the tail of every affine normalisation.
// In: a*R (Montgomery), non-zero.  Out: a^-1*R.  The integer inverse of a*R is a^-1*R^-1;
// one Montgomery product with R^3 lifts it back.
template <class C> ZKT_FN Fp<C> fp_inv(Fp<C> a) {
  constexpr int N = C::N;
  uint32_t u[N + 1], v[N + 1], x1[N + 1], x2[N + 1];
#pragma unroll
  for (int i = 0; i < N; ++i) { u[i] = a.v[i]; v[i] = C::mod(i); x1[i] = 0; x2[i] = 0; }
  u[N] = v[N] = x1[N] = x2[N] = 0; x1[0] = 1;
  auto is_one = [&](const uint32_t* t) { uint32_t o = t[0] ^ 1u;
#pragma unroll
    for (int i = 1; i <= N; ++i) o |= t[i]; return o == 0; };
  auto shr1 = [&](uint32_t* t) {
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = (t[i] >> 1) | (t[i + 1] << 31); t[N] >>= 1; };
  auto add_p = [&](uint32_t* t) { uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = addc(t[i], C::mod(i), c); t[N] += c; };
  auto sub = [&](uint32_t* t, const uint32_t* s) { uint32_t b = 0;
#pragma unroll
    for (int i = 0; i <= N; ++i) t[i] = subb(t[i], s[i], b); return b; };
  auto geq = [&](const uint32_t* t, const uint32_t* s) { uint32_t b = 0;
#pragma unroll
    for (int i = 0; i <= N; ++i) (void)subb(t[i], s[i], b); return b == 0; };
  auto halve_mod = [&](uint32_t* x) { if (x[0] & 1) add_p(x); shr1(x); };
  auto sub_mod = [&](uint32_t* x, const uint32_t* y) { if (sub(x, y)) add_p(x); };   // x,y in [0,p): borrow wraps mod 2^(32(N+1)), +p fixes it
  for (int guard = 0; guard < 4 * 32 * N + 8; ++guard) {
    if (is_one(u) || is_one(v)) break;
    while ((u[0] & 1) == 0) { shr1(u); halve_mod(x1); }
    while ((v[0] & 1) == 0) { shr1(v); halve_mod(x2); }
    if (geq(u, v)) { sub(u, v); sub_mod(x1, x2); } else { sub(v, u); sub_mod(x2, x1); }
  }
  const bool use1 = is_one(u);
  Fp<C> r, r3;
#pragma unroll
  for (int i = 0; i < N; ++i) { r.v[i] = use1 ? x1[i] : x2[i]; r3.v[i] = C::r3(i); }
  return fp_mul(r, r3);
}

// generic power with a run-time exponent of `nlimbs` 32-bit limbs (MSB-first
// square-and-multiply; pow, prime_field_elem.rs:311-328, computes the same residue)
template <class C> ZKT_FN Fp<C> fp_pow(Fp<C> a, const uint32_t* e, int nlimbs) {
  Fp<C> r = fp_one<C>();
  for (int i = nlimbs * 32 - 1; i >= 0; --i) {
    r = fp_sqr(r);
    if ((e[i >> 5] >> (i & 31)) & 1) r = fp_mul(r, a);
  }
  return r;
}

typedef Fp<FqC> Fq;
typedef Fp<FrC> FrE;
typedef Fp<SpC> SpE;
typedef Fp<SnC> SnE;

}  // namespace zkt
