// BLS12-381 extension tower Fq2 / Fq6 / Fq12 on top of fp.h.
//
// Same fields and the same element *values* as the reference's schoolbook tower
//   src/building_block/curves/bls12_381/fq2.rs:15-151   u^2 = -1
//   src/building_block/curves/bls12_381/fq6.rs:15-171   v^3 = 1+u
//   src/building_block/curves/bls12_381/fq12.rs:17-172  w^2 = v
// but with the usual fast formulas (Karatsuba 3/6/3, complex squaring, sparse line
// multiplication, Frobenius by constants).  Coefficients are stored low degree
// first (c0 + c1*g); the C ABI converts to the reference's struct order
// {u1,u0} / {v2,v1,v0} / {w1,w0} at the boundary.
#pragma once
#include "fp.h"

namespace zkt {

// Fq2 products are inlined into their Fq6/Fq12 callers by default (operands stay in VGPRs across the 6 products
// of an Fq6 multiply instead of being re-read from scratch); -DZKT_FQ2_CALLS makes them real functions.
#ifdef ZKT_FQ2_CALLS
#define ZKT_FQ2 ZKT_FN
#else
#define ZKT_FQ2 ZKT_HD
#endif
#if !defined(ZKT_FQ2_SPLIT)
struct Fq2 { Fq c0, c1; };
struct Fq6 { Fq2 c0, c1, c2; };
struct Fq12 { Fq6 c0, c1; };

// ---- Fq2 --------------------------------------------------------------------
ZKT_HD Fq2 fq2_zero() { return Fq2{fp_zero<FqC>(), fp_zero<FqC>()}; }
ZKT_HD Fq2 fq2_one() { return Fq2{fp_one<FqC>(), fp_zero<FqC>()}; }
ZKT_HD bool fq2_is_zero(const Fq2& a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1); }
ZKT_HD bool fq2_eq(const Fq2& a, const Fq2& b) { return fp_eq(a.c0, b.c0) && fp_eq(a.c1, b.c1); }
ZKT_HD Fq2 fq2_add(const Fq2& a, const Fq2& b) { return Fq2{fp_add(a.c0, b.c0), fp_add(a.c1, b.c1)}; }   // fq2.rs:96-110
ZKT_HD Fq2 fq2_sub(const Fq2& a, const Fq2& b) { return Fq2{fp_sub(a.c0, b.c0), fp_sub(a.c1, b.c1)}; }   // fq2.rs:115-129
ZKT_HD Fq2 fq2_neg(const Fq2& a) { return Fq2{fp_neg(a.c0), fp_neg(a.c1)}; }                               // fq2.rs:82-94
ZKT_HD Fq2 fq2_dbl(const Fq2& a) { return Fq2{fp_dbl(a.c0), fp_dbl(a.c1)}; }
ZKT_HD Fq2 fq2_conj(const Fq2& a) { return Fq2{a.c0, fp_neg(a.c1)}; }
// fq2.rs:134-146: the 4-product schoolbook, here with lazy reduction — each coordinate is a sum of two products under ONE
// Montgomery reduction (fp_mulsub / fp_muladd): 784 + 392 MADs like Karatsuba's three multiplications, but none of its five
// additions and subtractions.
// -DZKT_FQ2_KARATSUBA: three product scans instead of four (fp2_mul_kara, fp.h) — fewer multiply-adds, more additions: pays where one
// wave per SIMD runs (the pairing kernels), not where two waves share the issue port.
ZKT_FQ2 Fq2 fq2_mul(const Fq2& a, const Fq2& b) {
#if defined(ZKT_FQ2_KARATSUBA)
  Fq2 r; fp2_mul_kara(a.c0, a.c1, b.c0, b.c1, r.c0, r.c1); return r;
#else
  return Fq2{fp_mulsub(a.c0, b.c0, a.c1, b.c1), fp_muladd(a.c0, b.c1, a.c1, b.c0)};
#endif
}
ZKT_FQ2 Fq2 fq2_sqr(const Fq2& a) {                                      // fq2.rs:34-36
  Fq t = fp_mul(a.c0, a.c1);
  return Fq2{fp_mul(fp_add(a.c0, a.c1), fp_sub(a.c0, a.c1)), fp_dbl(t)};
}
ZKT_HD Fq2 fq2_mul_fq(const Fq2& a, const Fq& s) { return Fq2{fp_mul(a.c0, s), fp_mul(a.c1, s)}; }
ZKT_HD Fq2 fq2_sub2(const Fq2& a, const Fq2& b, const Fq2& c) { return Fq2{fp_sub2(a.c0, b.c0, c.c0), fp_sub2(a.c1, b.c1, c.c1)}; }   // a - b - 2c
ZKT_HD Fq2 fq2_mulsub(const Fq2& a, const Fq2& b, const Fq2& c, const Fq2& d) { return fq2_sub(fq2_mul(a, b), fq2_mul(c, d)); }          // a b - c d
ZKT_HD Fq2 fq2_subsub(const Fq2& a, const Fq2& b, const Fq2& c) { return Fq2{fp_subsub(a.c0, b.c0, c.c0), fp_subsub(a.c1, b.c1, c.c1)}; }   // a - b - c, one pass
ZKT_HD Fq2 fq2_add_mul_xi(const Fq2& v, const Fq2& t) { return Fq2{fp_addsub(v.c0, t.c0, t.c1), fp_add3(v.c1, t.c0, t.c1)}; }               // v + xi t, one pass
ZKT_HD Fq2 fq2_subsub_mul_xi(const Fq2& a, const Fq2& b, const Fq2& t) {                                                                        // a - b - xi t
  return Fq2{fp_addsub(a.c0, t.c1, fp_add(b.c0, t.c0)), fp_subsub(a.c1, b.c1, fp_add(t.c0, t.c1))}; }
ZKT_HD Fq2 fq2_from_fq(const Fq& a) { return Fq2{a, fp_zero<FqC>()}; }
// an Fq2 constant from its two limb tables
template <class F0, class F1> ZKT_HD Fq2 fq2_const(F0 c0_limb, F1 c1_limb) {
  Fq2 g;
#pragma unroll
  for (int i = 0; i < FqC::N; ++i) { g.c0.v[i] = c0_limb(i); g.c1.v[i] = c1_limb(i); }
  return g;
}
ZKT_HD Fq2 fq2_mul_xi(const Fq2& a) { return Fq2{fp_sub(a.c0, a.c1), fp_add(a.c0, a.c1)}; }   // Fq2::reduce, fq2.rs:52-58
ZKT_FN Fq2 fq2_inv(const Fq2& a) {                                      // fq2.rs:26-32
  Fq t = fp_inv(fp_add(fp_sqr(a.c0), fp_sqr(a.c1)));
  return Fq2{fp_mul(a.c0, t), fp_neg(fp_mul(a.c1, t))};
}

#else
// ---- Fq2 spread over a pair of adjacent lanes (device only; used by the G2 bucket accumulation, zkt_msm_g2pair.hip) -------------
// The even lane of a pair holds c0, the odd lane c1 of every Fq2 value; code written against the fq2_* functions does not change.
// Additions are lane-local; a product exchanges the operands with the partner lane (DPP quad_perm [1,0,3,2], one v_mov_dpp per limb)
// and is ONE two-product multiply per lane — even: a0 b0 - a1 b1, odd: a1 b0 + a0 b1.  Per-lane state halves (no spilling in the
// G2 bucket accumulation); the multiply count per Fq2 product is the same as on one lane.
struct Fq2 { Fq h; };
struct Fq6 { Fq2 c0, c1, c2; };
struct Fq12 { Fq6 c0, c1; };
__device__ inline bool fq2_odd() { return threadIdx.x & 1; }
__device__ inline Fq fq_partner(const Fq& a) { Fq r;
#pragma unroll
  for (int i = 0; i < FqC::N; ++i) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], 0xB1, 0xF, 0xF, true);
  return r; }
__device__ inline Fq fq_sel(bool c, const Fq& a, const Fq& b) { Fq r;      // c ? a : b
#pragma unroll
  for (int i = 0; i < FqC::N; ++i) r.v[i] = c ? a.v[i] : b.v[i];
  return r; }
__device__ inline Fq2 fq2_zero() { return Fq2{fp_zero<FqC>()}; }
__device__ inline Fq2 fq2_one() { return Fq2{fq_sel(fq2_odd(), fp_zero<FqC>(), fp_one<FqC>())}; }
__device__ inline bool fq2_is_zero(const Fq2& a) { int z = fp_is_zero(a.h); return z && __builtin_amdgcn_mov_dpp(z, 0xB1, 0xF, 0xF, true); }
__device__ inline Fq2 fq2_add(const Fq2& a, const Fq2& b) { return Fq2{fp_add(a.h, b.h)}; }
__device__ inline Fq2 fq2_sub(const Fq2& a, const Fq2& b) { return Fq2{fp_sub(a.h, b.h)}; }
__device__ inline bool fq2_eq(const Fq2& a, const Fq2& b) { return fq2_is_zero(fq2_sub(a, b)); }
__device__ inline Fq2 fq2_neg(const Fq2& a) { return Fq2{fp_neg(a.h)}; }
__device__ inline Fq2 fq2_dbl(const Fq2& a) { return Fq2{fp_dbl(a.h)}; }
__device__ inline Fq2 fq2_sub2(const Fq2& a, const Fq2& b, const Fq2& c) { return Fq2{fp_sub2(a.h, b.h, c.h)}; }
__device__ inline Fq2 fq2_conj(const Fq2& a) { return Fq2{fq_sel(fq2_odd(), fp_neg(a.h), a.h)}; }
__device__ inline Fq2 fq2_from_fq(const Fq& a) { return Fq2{fq_sel(fq2_odd(), fp_zero<FqC>(), a)}; }
template <class F0, class F1> __device__ inline Fq2 fq2_const(F0 c0_limb, F1 c1_limb) {
  Fq2 g; const bool odd = fq2_odd();
#pragma unroll
  for (int i = 0; i < FqC::N; ++i) g.h.v[i] = odd ? c1_limb(i) : c0_limb(i);
  return g;
}
// even: a0 b0 + a1 (8p - b1)      odd: a1 b0 + a0 b1      (own half = a.h, oa = the partner's)
__device__ inline Fq2 fq2_mul(const Fq2& a, const Fq2& b) {
  const bool odd = fq2_odd();
  const Fq oa = fq_partner(a.h), ob = fq_partner(b.h);
  Fq y1, nd;
#pragma unroll
  for (int i = 0; i < FqC::N; ++i) { y1.v[i] = odd ? ob.v[i] : b.h.v[i]; nd.v[i] = odd ? b.h.v[i] : FqC::subk(i) - ob.v[i]; }
  return Fq2{fp_muladd(a.h, y1, oa, nd)};                 // nd may carry 30-bit limbs: the second factor of fp_mul2 need not be normalised
}
__device__ inline Fq2 fq2_mulsub(const Fq2& a, const Fq2& b, const Fq2& c, const Fq2& d) { return fq2_sub(fq2_mul(a, b), fq2_mul(c, d)); }
__device__ inline Fq2 fq2_subsub(const Fq2& a, const Fq2& b, const Fq2& c) { return Fq2{fp_subsub(a.h, b.h, c.h)}; }
// even: (a0 + a1)(a0 - a1)        odd: (a1 + a1) a0
__device__ inline Fq2 fq2_sqr(const Fq2& a) {
  const bool odd = fq2_odd();
  const Fq oa = fq_partner(a.h);
  return Fq2{fp_mul(fp_add(a.h, fq_sel(odd, a.h, oa)), fq_sel(odd, oa, fp_sub(a.h, oa)))};
}
__device__ inline Fq2 fq2_mul_fq(const Fq2& a, const Fq& s) { return Fq2{fp_mul(a.h, s)}; }
// (c0 - c1, c0 + c1): even a0 - a1, odd a1 + a0 — one reduction pass either way
__device__ inline Fq2 fq2_mul_xi(const Fq2& a) {
  const bool odd = fq2_odd();
  const Fq oa = fq_partner(a.h);
  uint32_t v[FqC::N];
#pragma unroll
  for (int i = 0; i < FqC::N; ++i) v[i] = a.h.v[i] + (odd ? oa.v[i] : FqC::subk(i) - oa.v[i]);
  return Fq2{fp_lazy_reduce<FqC>(v)};
}
__device__ inline Fq2 fq2_add_mul_xi(const Fq2& v, const Fq2& t) { return fq2_add(v, fq2_mul_xi(t)); }
__device__ inline Fq2 fq2_subsub_mul_xi(const Fq2& a, const Fq2& b, const Fq2& t) { return fq2_subsub(a, b, fq2_mul_xi(t)); }
__device__ inline __attribute__((noinline)) Fq2 fq2_inv(const Fq2& a) {   // fq2.rs:26-32: conj / norm; both lanes invert the same norm
  const Fq sq = fp_sqr(a.h);
  const Fq t = fp_inv(fp_add(sq, fq_partner(sq)));
  const Fq m = fp_mul(a.h, t);
  return Fq2{fq_sel(fq2_odd(), fp_neg(m), m)};
}
#endif

// ---- Fq6 --------------------------------------------------------------------
ZKT_HD Fq6 fq6_zero() { return Fq6{fq2_zero(), fq2_zero(), fq2_zero()}; }
ZKT_HD Fq6 fq6_one() { return Fq6{fq2_one(), fq2_zero(), fq2_zero()}; }
ZKT_HD Fq6 fq6_add(const Fq6& a, const Fq6& b) { return Fq6{fq2_add(a.c0, b.c0), fq2_add(a.c1, b.c1), fq2_add(a.c2, b.c2)}; }
ZKT_HD Fq6 fq6_sub(const Fq6& a, const Fq6& b) { return Fq6{fq2_sub(a.c0, b.c0), fq2_sub(a.c1, b.c1), fq2_sub(a.c2, b.c2)}; }
ZKT_HD Fq6 fq6_neg(const Fq6& a) { return Fq6{fq2_neg(a.c0), fq2_neg(a.c1), fq2_neg(a.c2)}; }
ZKT_HD Fq6 fq6_mul_v(const Fq6& a) { return Fq6{fq2_mul_xi(a.c2), a.c0, a.c1}; }                 // Fq6::reduce, fq6.rs:54-62
// fq6.rs:148-166 is the 9-product schoolbook; Karatsuba (6 products) gives the same element
ZKT_HD Fq6 fq6_mul_inl(const Fq6& a, const Fq6& b) {
  Fq2 v0 = fq2_mul(a.c0, b.c0), v1 = fq2_mul(a.c1, b.c1), v2 = fq2_mul(a.c2, b.c2);
  Fq2 t0 = fq2_subsub(fq2_mul(fq2_add(a.c1, a.c2), fq2_add(b.c1, b.c2)), v1, v2);      // three-term recombinations: one reduction pass each
  Fq2 t1 = fq2_subsub(fq2_mul(fq2_add(a.c0, a.c1), fq2_add(b.c0, b.c1)), v0, v1);
  Fq2 t2 = fq2_subsub(fq2_mul(fq2_add(a.c0, a.c2), fq2_add(b.c0, b.c2)), v0, v2);
  return Fq6{fq2_add_mul_xi(v0, t0), fq2_add_mul_xi(t1, v2), fq2_add(t2, v1)};
}
ZKT_FN Fq6 fq6_mul(const Fq6& a, const Fq6& b) { return fq6_mul_inl(a, b); }
// inside the Fq12 square/product the Fq6 products are inlined (ZKT_FQ6_INLINE): operands and partial results stay in the
// 512-entry register file instead of round-tripping through scratch between calls
// the Fq12-level routines are real functions by default (operands and results travel through per-lane scratch: ~2 KB per call, the pairing kernels' memory traffic);
// -DZKT_FQ12_INLINE inlines them into their callers (A/B experiment of round 4: does the accumulator of a Miller loop stay in registers?)
#ifdef ZKT_FQ12_INLINE
#define ZKT_FQ12 ZKT_HD
#else
#define ZKT_FQ12 ZKT_FN
#endif
#ifdef ZKT_FQ6_CALLS
#define FQ6_MUL12 fq6_mul
#else
#define FQ6_MUL12 fq6_mul_inl
#endif
ZKT_FN Fq6 fq6_inv(const Fq6& a) {                                      // fq6.rs:23-37
  Fq2 t0 = fq2_sub(fq2_sqr(a.c0), fq2_mul_xi(fq2_mul(a.c1, a.c2)));
  Fq2 t1 = fq2_sub(fq2_mul_xi(fq2_sqr(a.c2)), fq2_mul(a.c0, a.c1));
  Fq2 t2 = fq2_sub(fq2_sqr(a.c1), fq2_mul(a.c0, a.c2));
  Fq2 f = fq2_inv(fq2_add(fq2_mul(a.c0, t0), fq2_add(fq2_mul_xi(fq2_mul(a.c2, t1)), fq2_mul_xi(fq2_mul(a.c1, t2)))));
  return Fq6{fq2_mul(t0, f), fq2_mul(t1, f), fq2_mul(t2, f)};
}

// ---- Fq12 -------------------------------------------------------------------
ZKT_HD Fq12 fq12_one() { return Fq12{fq6_one(), fq6_zero()}; }
ZKT_HD bool fq6_is_zero(const Fq6& a) { return fq2_is_zero(a.c0) && fq2_is_zero(a.c1) && fq2_is_zero(a.c2); }
ZKT_HD bool fq12_is_zero(const Fq12& a) { return fq6_is_zero(a.c0) && fq6_is_zero(a.c1); }
ZKT_HD Fq12 fq12_add(const Fq12& a, const Fq12& b) { return Fq12{fq6_add(a.c0, b.c0), fq6_add(a.c1, b.c1)}; }
ZKT_HD Fq12 fq12_sub(const Fq12& a, const Fq12& b) { return Fq12{fq6_sub(a.c0, b.c0), fq6_sub(a.c1, b.c1)}; }
ZKT_HD Fq12 fq12_neg(const Fq12& a) { return Fq12{fq6_neg(a.c0), fq6_neg(a.c1)}; }
ZKT_HD Fq12 fq12_conj(const Fq12& a) { return Fq12{a.c0, fq6_neg(a.c1)}; }
ZKT_HD void fq12_conj_ip(Fq12& a) { a.c1 = fq6_neg(a.c1); }      // in place: only the w-half moves
// fq12.rs:135-147 is 4 Fq6 products; Karatsuba (3) gives the same element
ZKT_HD Fq12 fq12_mul_body(const Fq12& a, const Fq12& b) {
  Fq6 v0 = FQ6_MUL12(a.c0, b.c0), v1 = FQ6_MUL12(a.c1, b.c1);
  Fq6 s = FQ6_MUL12(fq6_add(a.c0, a.c1), fq6_add(b.c0, b.c1));
  Fq12 r;
  r.c0.c0 = fq2_add_mul_xi(v0.c0, v1.c2); r.c0.c1 = fq2_add(v0.c1, v1.c0); r.c0.c2 = fq2_add(v0.c2, v1.c1);                   // v0 + v * v1
  r.c1.c0 = fq2_subsub(s.c0, v0.c0, v1.c0); r.c1.c1 = fq2_subsub(s.c1, v0.c1, v1.c1); r.c1.c2 = fq2_subsub(s.c2, v0.c2, v1.c2);
  return r;
}
ZKT_FQ12 Fq12 fq12_mul(const Fq12& a, const Fq12& b) { return fq12_mul_body(a, b); }
// complex squaring: (a0 + a1 w)^2 = (a0+a1)(a0+v a1) - v0 - v v0 + 2 v0 w,  v0 = a0 a1
ZKT_HD Fq12 fq12_sqr_body(const Fq12& a) {
  Fq6 v0 = FQ6_MUL12(a.c0, a.c1);
  Fq6 t = FQ6_MUL12(fq6_add(a.c0, a.c1), fq6_add(a.c0, fq6_mul_v(a.c1)));
  // (left as two-term passes: restructuring this return into three-term passes trips an AMDGPU backend error at -O1 —
  //  "Illegal instruction detected: Operand has incorrect register class ... $src_private_base" — in the verification kernels)
  return Fq12{fq6_sub(fq6_sub(t, v0), fq6_mul_v(v0)), fq6_add(v0, v0)};
}
ZKT_FQ12 Fq12 fq12_sqr(const Fq12& a) { return fq12_sqr_body(a); }
ZKT_FN Fq12 fq12_inv(const Fq12& a) {                                    // fq12.rs:31-40
  Fq6 t = fq6_inv(fq6_sub(fq6_mul(a.c0, a.c0), fq6_mul_v(fq6_mul(a.c1, a.c1))));
  return Fq12{fq6_mul(a.c0, t), fq6_neg(fq6_mul(a.c1, t))};
}

// Squaring in the cyclotomic subgroup G_{phi6}(q^2) (Granger-Scott): valid after the easy part of the final
// exponentiation, where a^(q^6+1) = 1.  Three Fq4 squarings = 9 Fq2 squarings instead of the 12 Fq2 products of the
// complex squaring; the result is the same field element as fq12_sqr on such inputs (tests/test_hostcheck.py).
ZKT_HD void fq4_sqr(const Fq2& a, const Fq2& b, Fq2& c0, Fq2& c1) {
  Fq2 t0 = fq2_sqr(a), t1 = fq2_sqr(b);
  c0 = fq2_add(fq2_mul_xi(t1), t0);
  c1 = fq2_sub(fq2_sub(fq2_sqr(fq2_add(a, b)), t0), t1);
}
ZKT_FQ12 Fq12 fq12_cyclotomic_sqr(const Fq12& f) {
  Fq2 z0 = f.c0.c0, z4 = f.c0.c1, z3 = f.c0.c2, z2 = f.c1.c0, z1 = f.c1.c1, z5 = f.c1.c2;
  Fq2 t0, t1, t2, t3;
  auto three_minus_two = [](const Fq2& t, const Fq2& z) { Fq2 d = fq2_sub(t, z); return fq2_add(fq2_dbl(d), t); };   // 3t - 2z
  auto three_plus_two = [](const Fq2& t, const Fq2& z) { Fq2 d = fq2_add(t, z); return fq2_add(fq2_dbl(d), t); };     // 3t + 2z
  fq4_sqr(z0, z1, t0, t1);
  z0 = three_minus_two(t0, z0); z1 = three_plus_two(t1, z1);
  fq4_sqr(z2, z3, t0, t1);
  fq4_sqr(z4, z5, t2, t3);
  z4 = three_minus_two(t0, z4); z5 = three_plus_two(t1, z5);
  t0 = fq2_mul_xi(t3);
  z2 = three_plus_two(t0, z2); z3 = three_minus_two(t2, z3);
  Fq12 r; r.c0 = Fq6{z0, z4, z3}; r.c1 = Fq6{z2, z1, z5};
  return r;
}

// n consecutive cyclotomic squarings IN PLACE (round 4).  The hard part of the final exponentiation is runs of squarings between a few products (|x| = 0xd201000000010000:
// runs of 1, 2, 3, 9, 32 and 16; the signed digits of e1: runs of ~4).  One call per squaring read the value from per-lane scratch and wrote it back (1.3 KB per call, 315 calls
// per pairing = 28 % of the pairing kernel's memory traffic); here the six Fq2 coefficients stay in registers for the whole run.  Same arithmetic, same element.
// m (optional): the run's value is multiplied by *m before it goes back to memory — a square-and-multiply step "n squarings, one product" without the round trip between them.
ZKT_FN void fq12_cyclotomic_sqr_n_mul(Fq12& f, int n, const Fq12* m) {
  ZKT_FORCE_FRAME();
  Fq2 z0 = f.c0.c0, z4 = f.c0.c1, z3 = f.c0.c2, z2 = f.c1.c0, z1 = f.c1.c1, z5 = f.c1.c2;
  auto three_minus_two = [](const Fq2& t, const Fq2& z) { Fq2 d = fq2_sub(t, z); return fq2_add(fq2_dbl(d), t); };   // 3t - 2z
  auto three_plus_two = [](const Fq2& t, const Fq2& z) { Fq2 d = fq2_add(t, z); return fq2_add(fq2_dbl(d), t); };     // 3t + 2z
#pragma unroll 1
  for (int k = 0; k < n; ++k) {
    Fq2 t0, t1, t2, t3;
    fq4_sqr(z0, z1, t0, t1);
    z0 = three_minus_two(t0, z0); z1 = three_plus_two(t1, z1);
    fq4_sqr(z2, z3, t0, t1);
    fq4_sqr(z4, z5, t2, t3);
    z4 = three_minus_two(t0, z4); z5 = three_plus_two(t1, z5);
    t0 = fq2_mul_xi(t3);
    z2 = three_plus_two(t0, z2); z3 = three_minus_two(t2, z3);
  }
  Fq12 r; r.c0 = Fq6{z0, z4, z3}; r.c1 = Fq6{z2, z1, z5};
  if (m) f = fq12_mul_body(r, *m); else f = r;
}
ZKT_HD void fq12_cyclotomic_sqr_n(Fq12& f, int n) { fq12_cyclotomic_sqr_n_mul(f, n, nullptr); }

// Frobenius pi^K, K in {1,2}: conj^K on every Fq2 coefficient of w^i times gamma_i^(K)
template <int K> ZKT_HD Fq2 frob_const(int idx) {
  return fq2_const([&](int i) { return K == 1 ? frob1_limb(idx, 0, i) : frob2_limb(idx, 0, i); },
                   [&](int i) { return K == 1 ? frob1_limb(idx, 1, i) : frob2_limb(idx, 1, i); });
}
ZKT_HD Fq frob2_fq(int idx) { Fq g;      // gamma^(2) lies in Fq
#pragma unroll
  for (int i = 0; i < FqC::N; ++i) g.v[i] = frob2_limb(idx, 0, i);
  return g; }
template <int K> ZKT_HD Fq2 frob_coeff(const Fq2& a, int idx) {
  Fq2 c = (K & 1) ? fq2_conj(a) : a;
  if (idx == 0) return c;
  if (K == 2) return fq2_mul_fq(c, frob2_fq(idx));
  return fq2_mul(c, frob_const<1>(idx));
}
// (the inlined form is for callers that keep a base pointer — final_exponentiation_t: as a called function pi^1 parks literals in s34, which such a caller needs
//  kept and this toolchain does not keep; pairing.h "ZKT_ATE_STEP", tools/check_base_pointer.py)
template <int K> ZKT_HD Fq12 fq12_frob_inl(const Fq12& a) {
  // basis w^i: 0->c0.c0, 1->c1.c0, 2->c0.c1, 3->c1.c1, 4->c0.c2, 5->c1.c2
  Fq12 r;
  r.c0.c0 = frob_coeff<K>(a.c0.c0, 0); r.c1.c0 = frob_coeff<K>(a.c1.c0, 1);
  r.c0.c1 = frob_coeff<K>(a.c0.c1, 2); r.c1.c1 = frob_coeff<K>(a.c1.c1, 3);
  r.c0.c2 = frob_coeff<K>(a.c0.c2, 4); r.c1.c2 = frob_coeff<K>(a.c1.c2, 5);
  return r;
}
template <int K> ZKT_FN Fq12 fq12_frob(const Fq12& a) { return fq12_frob_inl<K>(a); }

// f * (a + b v^2 + c v w) with a in Fq, b,c in Fq2: the value of a Miller line at
// an untwisted G2 point has exactly these slots (g12_point.rs:47-68).
ZKT_HD Fq12 fq12_mul_line_body(const Fq12& f, const Fq& a, const Fq2& b, const Fq2& c) {
  // (x + y w)(a + b v^2 + c v w), w^2 = v, v^3 = xi:
  //   c0 = x a + x b v^2 + y c v^2         c1 = y a + x c v + y b v^2
  // The six products x_i b and y_i c serve c0; the cross terms of c1 pair up slot by slot — x2 c + y1 b, x1 c + y0 b and
  // x0 c + xi y2 b — so each pair is one Karatsuba product with (b + c): 9 Fq2 products instead of 12.
  const Fq6 &x = f.c0, &y = f.c1;
  const Fq2 xb0 = fq2_mul(x.c0, b), xb1 = fq2_mul(x.c1, b), xb2 = fq2_mul(x.c2, b);
  const Fq2 yc0 = fq2_mul(y.c0, c), yc1 = fq2_mul(y.c1, c), yc2 = fq2_mul(y.c2, c);
  const Fq2 bc = fq2_add(b, c);
  const Fq2 k0 = fq2_mul(fq2_add(x.c2, y.c1), bc);                       // x2 b + x2 c + y1 b + y1 c
  const Fq2 k1 = fq2_mul(fq2_add(x.c0, fq2_mul_xi(y.c2)), bc);            // x0 b + x0 c + xi y2 b + xi y2 c
  const Fq2 k2 = fq2_mul(fq2_add(x.c1, y.c0), bc);                       // x1 b + x1 c + y0 b + y0 c
  Fq12 r;
  r.c0.c0 = fq2_add_mul_xi(fq2_mul_fq(x.c0, a), fq2_add(xb1, yc1));      // x0 a + xi (x1 b + y1 c)
  r.c0.c1 = fq2_add_mul_xi(fq2_mul_fq(x.c1, a), fq2_add(xb2, yc2));      // x1 a + xi (x2 b + y2 c)
  r.c0.c2 = fq2_add(fq2_mul_fq(x.c2, a), fq2_add(xb0, yc0));             // x2 a + x0 b + y0 c
  r.c1.c0 = fq2_add_mul_xi(fq2_mul_fq(y.c0, a), fq2_subsub(k0, xb2, yc1));             // y0 a + xi (x2 c + y1 b)
  r.c1.c1 = fq2_add(fq2_mul_fq(y.c1, a), fq2_subsub(k1, xb0, fq2_mul_xi(yc2)));        // y1 a + x0 c + xi y2 b
  r.c1.c2 = fq2_add(fq2_mul_fq(y.c2, a), fq2_subsub(k2, xb1, yc0));                    // y2 a + x1 c + y0 b
  return r;
}
ZKT_FQ12 Fq12 fq12_mul_line(const Fq12& f, const Fq& a, const Fq2& b, const Fq2& c) { return fq12_mul_line_body(f, a, b, c); }
// f <- f * line IN PLACE (the lines of a step after the first, and the addition steps): "ft = f * line; f = ft" wrote the product, read it back and wrote it again
// (2.3 KB per line more than this form)
ZKT_FN void fq12_mul_line_ip(Fq12& f, const Fq& a, const Fq2& b, const Fq2& c) {
  ZKT_FORCE_FRAME();
  const Fq12 t = fq12_mul_line_body(f, a, b, c);
  f = t;
}
// One Miller doubling step on the accumulator, IN PLACE: f <- f^2 * line.  As two calls the square travelled through per-lane scratch between them (672 B out, 672 B back,
// 127 times per pairing); as one function it is the compiler's to keep (round 4, after fq12_cyclotomic_sqr_n showed what those round trips cost).
ZKT_FN void fq12_sqr_mul_line(Fq12& f, const Fq& a, const Fq2& b, const Fq2& c) {
  ZKT_FORCE_FRAME();
  const Fq12 t = fq12_sqr_body(f);
  f = fq12_mul_line_body(t, a, b, c);
}

// square-and-multiply by a run-time exponent (u32 limbs, little endian), MSB first.
// Fq12::pow (fq12.rs:42-57) is LSB-first; the power is the same element.
ZKT_FN Fq12 fq12_pow(const Fq12& a, const uint32_t* e, int nlimbs) {
  Fq12 r = fq12_one(); bool started = false;
  for (int i = nlimbs * 32 - 1; i >= 0; --i) {
    if (started) r = fq12_sqr(r);
    if ((e[i >> 5] >> (i & 31)) & 1) { r = started ? fq12_mul(r, a) : a; started = true; }
  }
  return r;
}

// same, for elements of the cyclotomic subgroup (final exponentiation hard part)
ZKT_FN Fq12 fq12_cyclotomic_pow(const Fq12& a, const uint32_t* e, int nlimbs) {
  Fq12 r = fq12_one(), t; bool started = false;
  for (int i = nlimbs * 32 - 1; i >= 0; --i) {
    if (started) { t = fq12_cyclotomic_sqr(r); r = t; }
    if ((e[i >> 5] >> (i & 31)) & 1) { if (started) { t = fq12_mul(r, a); r = t; } else { r = a; started = true; } }
  }
  return r;
}

}  // namespace zkt
