// Short-Weierstrass a=0 group arithmetic, generic over the coordinate field
// (Fq for G1, Fq2 for G2, secp256k1's Fp), in Jacobian and XYZZ coordinates.
//
// The reference adds affine points with one field inversion per addition
//   impl_affine_add!        src/building_block/curves/macros.rs:34-163
//   impl_scalar_mul_point!  src/building_block/curves/macros.rs:1-32
// Any correct group law yields the same group element, and every result leaves
// this engine as a canonical affine point (or the infinity flag), so projective
// coordinates are bit-compatible.  The reference's case split (inf+P, P+(-P),
// P+P, y=0) is reproduced by the exceptional-case branches below.
#pragma once
#include "tower.h"

namespace zkt {

// ---- field policies -----------------------------------------------------------
template <class C> struct PrimeOps {
  typedef Fp<C> E;
  ZKT_HD static E zero() { return fp_zero<C>(); }
  ZKT_HD static E one() { return fp_one<C>(); }
  ZKT_HD static E add(const E& a, const E& b) { return fp_add(a, b); }
  ZKT_HD static E sub(const E& a, const E& b) { return fp_sub(a, b); }
  ZKT_HD static E mul(const E& a, const E& b) { return fp_mul(a, b); }
  ZKT_HD static E sqr(const E& a) { return fp_sqr(a); }
  ZKT_HD static E dbl(const E& a) { return fp_dbl(a); }
  ZKT_HD static E sub2(const E& a, const E& b, const E& c) { return fp_sub2(a, b, c); }                    // a - b - 2c
  ZKT_HD static E mulsub(const E& a, const E& b, const E& c, const E& d) { return fp_mulsub(a, b, c, d); }   // a b - c d
  ZKT_HD static E neg(const E& a) { return fp_neg(a); }
  ZKT_HD static E inv(const E& a) { return fp_inv(a); }
  ZKT_HD static bool is_zero(const E& a) { return fp_is_zero(a); }
  ZKT_HD static bool eq(const E& a, const E& b) { return fp_eq(a, b); }
};
typedef PrimeOps<FqC> FqOps;
typedef PrimeOps<SpC> SpOps;
struct Fq2Ops {
  typedef Fq2 E;
  ZKT_HD static E zero() { return fq2_zero(); }
  ZKT_HD static E one() { return fq2_one(); }
  ZKT_HD static E add(const E& a, const E& b) { return fq2_add(a, b); }
  ZKT_HD static E sub(const E& a, const E& b) { return fq2_sub(a, b); }
  ZKT_HD static E mul(const E& a, const E& b) { return fq2_mul(a, b); }
  ZKT_HD static E sqr(const E& a) { return fq2_sqr(a); }
  ZKT_HD static E dbl(const E& a) { return fq2_dbl(a); }
  ZKT_HD static E sub2(const E& a, const E& b, const E& c) { return fq2_sub2(a, b, c); }
  ZKT_HD static E mulsub(const E& a, const E& b, const E& c, const E& d) { return fq2_mulsub(a, b, c, d); }
  ZKT_HD static E neg(const E& a) { return fq2_neg(a); }
  ZKT_HD static E inv(const E& a) { return fq2_inv(a); }
  ZKT_HD static bool is_zero(const E& a) { return fq2_is_zero(a); }
  ZKT_HD static bool eq(const E& a, const E& b) { return fq2_eq(a, b); }
};

// ---- point types (coordinates in the Montgomery domain) -----------------------
template <class F> struct Aff { typename F::E x, y; bool inf; };
template <class F> struct Jac { typename F::E X, Y, Z; };            // infinity <=> Z == 0
template <class F> struct Xyzz { typename F::E X, Y, ZZ, ZZZ; };     // infinity <=> ZZ == 0

template <class F> ZKT_HD Jac<F> jac_inf() { return Jac<F>{F::one(), F::one(), F::zero()}; }
template <class F> ZKT_HD bool jac_is_inf(const Jac<F>& p) { return F::is_zero(p.Z); }
template <class F> ZKT_HD Jac<F> jac_from_aff(const Aff<F>& a) {
  if (a.inf) return jac_inf<F>();
  return Jac<F>{a.x, a.y, F::one()};
}

// dbl-2009-l (a = 0): 2M + 5S.  y = 0 => Z3 = 0 => infinity (macros.rs:61-63).
template <class F> ZKT_HD Jac<F> jac_dbl(const Jac<F>& p) {
  typedef typename F::E E;
  E A = F::sqr(p.X), B = F::sqr(p.Y), C = F::sqr(B);
  E t = F::sqr(F::add(p.X, B));
  E D = F::dbl(F::sub(F::sub(t, A), C));
  E Ee = F::add(F::dbl(A), A);
  E Ff = F::sqr(Ee);
  Jac<F> r;
  r.X = F::sub2(Ff, F::zero(), D);
  E C8 = F::dbl(F::dbl(F::dbl(C)));
  r.Y = F::sub(F::mul(Ee, F::sub(D, r.X)), C8);
  r.Z = F::dbl(F::mul(p.Y, p.Z));
  return r;
}

// mixed addition Jacobian + affine, complete: handles P=inf, Q=inf, P==Q, P==-Q
// (case order of macros.rs:43-63).
template <class F> ZKT_HD Jac<F> jac_add_aff(const Jac<F>& p, const Aff<F>& q) {
  typedef typename F::E E;
  if (q.inf) return p;
  if (jac_is_inf(p)) return Jac<F>{q.x, q.y, F::one()};
  E ZZ = F::sqr(p.Z);
  E U2 = F::mul(q.x, ZZ);
  E S2 = F::mul(F::mul(q.y, p.Z), ZZ);
  E H = F::sub(U2, p.X);
  E Rr = F::sub(S2, p.Y);
  if (F::is_zero(H)) {
    if (F::is_zero(Rr)) return jac_dbl(Jac<F>{q.x, q.y, F::one()});   // same point
    return jac_inf<F>();                                               // vertical line
  }
  E HH = F::sqr(H), HHH = F::mul(H, HH), V = F::mul(p.X, HH);
  Jac<F> r;
  r.X = F::sub2(F::sqr(Rr), HHH, V);
  r.Y = F::mulsub(Rr, F::sub(V, r.X), p.Y, HHH);
  r.Z = F::mul(p.Z, H);
  return r;
}

// full Jacobian addition (add-2007-bl without the doubling trick), complete.
template <class F> ZKT_HD Jac<F> jac_add(const Jac<F>& p, const Jac<F>& q) {
  typedef typename F::E E;
  if (jac_is_inf(q)) return p;
  if (jac_is_inf(p)) return q;
  E Z1Z1 = F::sqr(p.Z), Z2Z2 = F::sqr(q.Z);
  E U1 = F::mul(p.X, Z2Z2), U2 = F::mul(q.X, Z1Z1);
  E S1 = F::mul(F::mul(p.Y, q.Z), Z2Z2), S2 = F::mul(F::mul(q.Y, p.Z), Z1Z1);
  E H = F::sub(U2, U1), Rr = F::sub(S2, S1);
  if (F::is_zero(H)) {
    if (F::is_zero(Rr)) return jac_dbl(p);
    return jac_inf<F>();
  }
  E HH = F::sqr(H), HHH = F::mul(H, HH), V = F::mul(U1, HH);
  Jac<F> r;
  r.X = F::sub2(F::sqr(Rr), HHH, V);
  r.Y = F::mulsub(Rr, F::sub(V, r.X), S1, HHH);
  r.Z = F::mul(F::mul(p.Z, q.Z), H);
  return r;
}

// Jacobian -> affine: the one inversion per output point.
template <class F> ZKT_HD Aff<F> jac_to_aff(const Jac<F>& p) {
  typedef typename F::E E;
  Aff<F> a;
  if (jac_is_inf(p)) { a.x = F::zero(); a.y = F::zero(); a.inf = true; return a; }
  E zi = F::inv(p.Z), zi2 = F::sqr(zi);
  a.x = F::mul(p.X, zi2);
  a.y = F::mul(p.Y, F::mul(zi2, zi));
  a.inf = false;
  return a;
}

// ---- XYZZ (bucket accumulators): x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2 --------------
template <class F> ZKT_HD Xyzz<F> xyzz_inf() { return Xyzz<F>{F::one(), F::one(), F::zero(), F::zero()}; }
template <class F> ZKT_HD bool xyzz_is_inf(const Xyzz<F>& p) { return F::is_zero(p.ZZ); }
// dbl-2008-s-1 from an affine point (ZZ = ZZZ = 1)
template <class F> ZKT_HD Xyzz<F> xyzz_dbl_aff(const typename F::E& x, const typename F::E& y) {
  typedef typename F::E E;
  if (F::is_zero(y)) return xyzz_inf<F>();
  E U = F::dbl(y), V = F::sqr(U), W = F::mul(U, V), S = F::mul(x, V);
  E xx = F::sqr(x), M = F::add(F::dbl(xx), xx);
  Xyzz<F> r;
  r.X = F::sub2(F::sqr(M), F::zero(), S);
  r.Y = F::mulsub(M, F::sub(S, r.X), W, y);
  r.ZZ = V; r.ZZZ = W;
  return r;
}
template <class F> ZKT_HD Xyzz<F> xyzz_dbl(const Xyzz<F>& p) {
  typedef typename F::E E;
  if (xyzz_is_inf(p) || F::is_zero(p.Y)) return xyzz_inf<F>();
  E U = F::dbl(p.Y), V = F::sqr(U), W = F::mul(U, V), S = F::mul(p.X, V);
  E xx = F::sqr(p.X), M = F::add(F::dbl(xx), xx);
  Xyzz<F> r;
  r.X = F::sub2(F::sqr(M), F::zero(), S);
  r.Y = F::mulsub(M, F::sub(S, r.X), W, p.Y);
  r.ZZ = F::mul(V, p.ZZ); r.ZZZ = F::mul(W, p.ZZZ);
  return r;
}
// madd-2008-s: XYZZ += affine (x2, y2), 8M + 2S, complete
template <class F> ZKT_HD Xyzz<F> xyzz_add_aff(const Xyzz<F>& p, const typename F::E& x2, const typename F::E& y2) {
  typedef typename F::E E;
  if (xyzz_is_inf(p)) return Xyzz<F>{x2, y2, F::one(), F::one()};
  E U2 = F::mul(x2, p.ZZ), S2 = F::mul(y2, p.ZZZ);
  E Pp = F::sub(U2, p.X), Rr = F::sub(S2, p.Y);
  if (F::is_zero(Pp)) {
    if (F::is_zero(Rr)) return xyzz_dbl_aff<F>(x2, y2);
    return xyzz_inf<F>();
  }
  E PP = F::sqr(Pp), PPP = F::mul(Pp, PP), Qq = F::mul(p.X, PP);
  Xyzz<F> r;
  r.X = F::sub2(F::sqr(Rr), PPP, Qq);
  r.Y = F::mulsub(Rr, F::sub(Qq, r.X), p.Y, PPP);
  r.ZZ = F::mul(p.ZZ, PP); r.ZZZ = F::mul(p.ZZZ, PPP);
  return r;
}
// add-2008-s: XYZZ + XYZZ, 12M + 2S, complete
template <class F> ZKT_HD Xyzz<F> xyzz_add(const Xyzz<F>& p, const Xyzz<F>& q) {
  typedef typename F::E E;
  if (xyzz_is_inf(q)) return p;
  if (xyzz_is_inf(p)) return q;
  E U1 = F::mul(p.X, q.ZZ), U2 = F::mul(q.X, p.ZZ);
  E S1 = F::mul(p.Y, q.ZZZ), S2 = F::mul(q.Y, p.ZZZ);
  E Pp = F::sub(U2, U1), Rr = F::sub(S2, S1);
  if (F::is_zero(Pp)) {
    if (F::is_zero(Rr)) return xyzz_dbl(p);
    return xyzz_inf<F>();
  }
  E PP = F::sqr(Pp), PPP = F::mul(Pp, PP), Qq = F::mul(U1, PP);
  Xyzz<F> r;
  r.X = F::sub2(F::sqr(Rr), PPP, Qq);
  r.Y = F::mulsub(Rr, F::sub(Qq, r.X), S1, PPP);
  r.ZZ = F::mul(F::mul(p.ZZ, q.ZZ), PP); r.ZZZ = F::mul(F::mul(p.ZZZ, q.ZZZ), PPP);
  return r;
}
template <class F> ZKT_HD Aff<F> xyzz_to_aff(const Xyzz<F>& p) {
  typedef typename F::E E;
  Aff<F> a;
  if (xyzz_is_inf(p)) { a.x = F::zero(); a.y = F::zero(); a.inf = true; return a; }
  // one inversion: 1/ZZZ; 1/ZZ = ZZ^2/ZZ^3 = ZZ^2/ZZZ^2 = (ZZ/ZZZ)^2
  E zi3 = F::inv(p.ZZZ);
  E zi2 = F::sqr(F::mul(zi3, p.ZZ));
  a.x = F::mul(p.X, zi2);
  a.y = F::mul(p.Y, zi3);
  a.inf = false;
  return a;
}
template <class F> ZKT_HD Jac<F> xyzz_to_jac(const Xyzz<F>& p) {
  // x = X/ZZ = (X*ZZ)/ZZ^2, y = Y/ZZZ = (Y*ZZZ)/ZZZ^2 = (Y*ZZZ)/ZZ^3  =>  (X*ZZ, Y*ZZZ, Z=ZZ)
  if (xyzz_is_inf(p)) return jac_inf<F>();
  return Jac<F>{F::mul(p.X, p.ZZ), F::mul(p.Y, p.ZZZ), p.ZZ};
}

// ---- scalar multiplication ---------------------------------------------------
// k*P for an affine P and an nbits-wide scalar (little-endian u32 limbs, used as-is,
// no reduction mod r: macros.rs:10-21).  MSB-first double-and-add with mixed
// additions; the reference's LSB-first loop computes the same group element.
// k * P with k given as `nlimbs` 32-bit words, used as-is (macros.rs:10-21 runs LSB-first double-and-add on the stored integer; the
// group element is the same).  The scalar is recoded on the fly into its non-adjacent form — one third instead of one half of the
// digits are non-zero, a digit -1 adds -P — and scanned from the top: nlimbs*32 doublings at most, ~nlimbs*32/3 mixed additions.
static constexpr int SCALAR_MAX_LIMBS = 12;          // the ABI allows up to 384-bit scalars
template <class F> ZKT_HD Jac<F> scalar_mul_aff(const Aff<F>& p, const uint32_t* k, int nlimbs) {
  Jac<F> acc = jac_inf<F>();
  if (p.inf) return acc;
  // NAF by the carry recurrence: with x = k + carry, digit = 0 if x even, else 2 - (x mod 4); carry' = (x - digit) / 2 ... done on bits
  uint32_t nz[SCALAR_MAX_LIMBS + 1], ng[SCALAR_MAX_LIMBS + 1];
  uint32_t carry = 0;
  for (int w = 0; w <= nlimbs; ++w) {
    const uint32_t cur = w < nlimbs ? k[w] : 0u, nxt = w + 1 < nlimbs ? k[w + 1] : 0u;
    uint32_t znz = 0, zng = 0;
    for (int b = 0; b < 32; ++b) {
      const uint32_t b0 = (cur >> b) & 1, b1 = b < 31 ? (cur >> (b + 1)) & 1 : nxt & 1;
      const uint32_t x0 = b0 ^ carry;                       // low bit of (k >> pos) + carry
      // x odd: digit = +1 if the next bit of the sum is 0, else -1 (and a carry is generated)
      const uint32_t sum1 = b1 ^ (b0 & carry);              // second bit of (k >> pos) + carry
      const uint32_t neg = x0 & sum1;
      znz |= x0 << b; zng |= neg << b;
      carry = (b0 & carry) | (neg);                         // carry into the next position: from the addition, or from rounding up
    }
    nz[w] = znz; ng[w] = zng;
  }
  Aff<F> q = p; q.y = F::neg(p.y);
  bool started = false;
  for (int i = nlimbs * 32; i >= 0; --i) {
    if (started) acc = jac_dbl(acc);
    if ((nz[i >> 5] >> (i & 31)) & 1) { acc = jac_add_aff(acc, ((ng[i >> 5] >> (i & 31)) & 1) ? q : p); started = true; }
  }
  return acc;
}

}  // namespace zkt
