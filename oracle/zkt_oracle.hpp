// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// CPU oracle: a plain C++ restatement of the reference's algorithm for the hot
// path (exfinen/zk-toolkit, paths relative to /root/reference/).  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
// there only as the checker.  Nothing under zk-toolkit_amd/ links, includes or
// calls this file.
//
// It follows the reference *algorithm* (canonical residues, schoolbook tower,
// affine add with one inversion, LSB-first double-and-add, textbook Miller loop
// with untwist + vertical lines, 4314-bit square-and-multiply); big-integer
// arithmetic that the reference gets from num-bigint 0.4.3 is restated here on
// fixed 64-bit limb arrays (Barrett reduction, binary extended Euclid) — those
// are exact integer operations whose results are unique, so results are
// identical residue for residue.
//
// Parity pin: the reference cannot be built here (Rust, no toolchain), so this
// oracle is pinned by the reference's own known-answer tests, transcribed in
// tests/golden/ref_kats.json with file:line provenance and checked by
// tests/test_oracle_kats.py.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>
#include <stdexcept>

namespace zkto {

typedef unsigned __int128 u128;
static const int MAXL = 6;  // max 64-bit limbs of a modulus (Fq = 381 bits)

// ---------------------------------------------------------------------------
// raw limb helpers (little-endian 64-bit limbs)
// ---------------------------------------------------------------------------
static inline int limb_cmp(const uint64_t* a, const uint64_t* b, int n) {
  for (int i = n - 1; i >= 0; --i) {
    if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  }
  return 0;
}
static inline uint64_t limb_add(uint64_t* r, const uint64_t* a, const uint64_t* b, int n) {
  u128 c = 0;
  for (int i = 0; i < n; ++i) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
  return (uint64_t)c;
}
static inline uint64_t limb_sub(uint64_t* r, const uint64_t* a, const uint64_t* b, int n) {
  uint64_t borrow = 0;
  for (int i = 0; i < n; ++i) {
    u128 d = (u128)a[i] - b[i] - borrow;
    r[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1;
  }
  return borrow;
}
static inline bool limb_is_zero(const uint64_t* a, int n) {
  uint64_t o = 0; for (int i = 0; i < n; ++i) o |= a[i]; return o == 0;
}
static inline void limb_mul(uint64_t* r /*na+nb*/, const uint64_t* a, int na, const uint64_t* b, int nb) {
  for (int i = 0; i < na + nb; ++i) r[i] = 0;
  for (int i = 0; i < na; ++i) {
    u128 c = 0;
    for (int j = 0; j < nb; ++j) {
      c += (u128)a[i] * b[j] + r[i + j];
      r[i + j] = (uint64_t)c; c >>= 64;
    }
    r[i + nb] = (uint64_t)c;
  }
}

// ---------------------------------------------------------------------------
// Prime field parameters (reference: PrimeField{order},
// src/building_block/field/prime_field.rs:15-26)
// ---------------------------------------------------------------------------
struct FieldParams {
  int k;                    // limbs actually used by the modulus (top limb != 0)
  uint64_t p[MAXL];         // modulus
  uint64_t mu[MAXL + 1];    // floor(2^(128k) / p), Barrett constant (k+1 limbs)

  void init_from_limbs(const uint64_t* m, int nlimbs) {
    k = nlimbs;
    while (k > 1 && m[k - 1] == 0) --k;
    for (int i = 0; i < MAXL; ++i) p[i] = i < k ? m[i] : 0;
    // mu = floor(b^(2k)/p) by bit-serial long division (run once)
    uint64_t rem[MAXL + 1] = {0};
    uint64_t quo[2 * MAXL + 1] = {0};
    int nbits = 128 * k + 1;  // dividend = 1 followed by 128k zero bits
    for (int bit = nbits - 1; bit >= 0; --bit) {
      // rem = rem*2 + dividend_bit
      uint64_t carry = (bit == nbits - 1) ? 1 : 0;
      for (int i = 0; i <= k; ++i) { uint64_t nc = rem[i] >> 63; rem[i] = (rem[i] << 1) | carry; carry = nc; }
      uint64_t pp[MAXL + 1]; for (int i = 0; i <= k; ++i) pp[i] = i < k ? p[i] : 0;
      if (limb_cmp(rem, pp, k + 1) >= 0) { limb_sub(rem, rem, pp, k + 1); quo[bit / 64] |= (uint64_t)1 << (bit % 64); }
    }
    for (int i = 0; i <= MAXL; ++i) mu[i] = i <= k ? quo[i] : 0;
  }

  // x (2k limbs, x < p*b^k is NOT required; any 2k-limb value) -> x mod p.  HAC 14.42.
  void reduce_wide(uint64_t* r /*k*/, const uint64_t* x /*2k*/) const {
    uint64_t q2[2 * MAXL + 2];
    limb_mul(q2, x + (k - 1), k + 1, mu, k + 1);          // q1*mu
    const uint64_t* q3 = q2 + (k + 1);                      // floor(q2 / b^(k+1)), k+1 limbs
    uint64_t qp[2 * MAXL + 2];
    limb_mul(qp, q3, k + 1, p, k);                          // q3*p (only low k+1 limbs used)
    uint64_t t[MAXL + 1];
    limb_sub(t, x, qp, k + 1);                              // mod b^(k+1): wraparound is the HAC "+b^(k+1)" step
    uint64_t pp[MAXL + 1]; for (int i = 0; i <= k; ++i) pp[i] = i < k ? p[i] : 0;
    while (limb_cmp(t, pp, k + 1) >= 0) limb_sub(t, t, pp, k + 1);
    for (int i = 0; i < k; ++i) r[i] = t[i];
  }
};

// The four fixed fields of the hot path + one dynamic slot for the reference's
// small-modulus PrimeFieldElem tests (prime_field_elem.rs:465-964).
struct FqTag  { static FieldParams P; };   // BLS12-381 base field, params.rs:8-11
struct FrTag  { static FieldParams P; };   // BLS12-381 subgroup order, params.rs:13-16
struct SpTag  { static FieldParams P; };   // secp256k1 base field, secp256k1/affine_point.rs:30-47
struct SnTag  { static FieldParams P; };   // secp256k1 group order
struct DynTag { static FieldParams P; };   // set by tests

// ---------------------------------------------------------------------------
// PrimeFieldElem restated: canonical residue in [0, order)
// (src/building_block/field/prime_field_elem.rs:263-457)
// ---------------------------------------------------------------------------
template <class Tag>
struct Fp {
  uint64_t l[MAXL];
  static const FieldParams& F() { return Tag::P; }

  Fp() { for (int i = 0; i < MAXL; ++i) l[i] = 0; }
  explicit Fp(uint64_t v) { for (int i = 0; i < MAXL; ++i) l[i] = 0; l[0] = v; reduce_self(); }

  // PrimeFieldElem::new — reduces e mod order on construction (:263-272)
  static Fp from_limbs(const uint64_t* v, int n) {
    // accept up to 2k limbs
    const FieldParams& f = F();
    uint64_t wide[2 * MAXL] = {0};
    for (int i = 0; i < n && i < 2 * f.k; ++i) wide[i] = v[i];
    Fp r; f.reduce_wide(r.l, wide); return r;
  }
  void reduce_self() { *this = from_limbs(l, F().k); }

  bool is_zero() const { return limb_is_zero(l, MAXL); }
  bool operator==(const Fp& o) const { return limb_cmp(l, o.l, MAXL) == 0; }
  bool operator!=(const Fp& o) const { return !(*this == o); }

  // plus (:278-286)
  Fp operator+(const Fp& o) const {
    const FieldParams& f = F(); Fp r;
    uint64_t c = limb_add(r.l, l, o.l, f.k);
    if (c || limb_cmp(r.l, f.p, f.k) >= 0) limb_sub(r.l, r.l, f.p, f.k);
    return r;
  }
  // minus (:288-300): a<b -> order-(b-a)
  Fp operator-(const Fp& o) const {
    const FieldParams& f = F(); Fp r;
    if (limb_cmp(l, o.l, f.k) < 0) {
      uint64_t d[MAXL]; limb_sub(d, o.l, l, f.k); limb_sub(r.l, f.p, d, f.k);
    } else {
      limb_sub(r.l, l, o.l, f.k);
    }
    return r;
  }
  // times (:302-308): BigUint multiply then %
  Fp operator*(const Fp& o) const {
    const FieldParams& f = F(); Fp r;
    uint64_t w[2 * MAXL]; limb_mul(w, l, f.k, o.l, f.k);
    f.reduce_wide(r.l, w); return r;
  }
  // sq (:330-335)
  Fp sq() const { return (*this) * (*this); }
  // negate (:448-457): 0 stays 0
  Fp negate() const {
    const FieldParams& f = F();
    if (is_zero()) return *this;
    Fp r; limb_sub(r.l, f.p, l, f.k); return r;
  }
  Fp operator-() const { return negate(); }

  // pow (:311-328) LSB-first square-and-multiply over the exponent's limbs
  Fp pow_limbs(const uint64_t* e, int n) const {
    Fp sum(1), bit_value = *this;
    int top = n * 64;
    while (top > 0 && !((e[(top - 1) / 64] >> ((top - 1) % 64)) & 1)) --top;
    // the reference iterates over whole little-endian *bytes* of the exponent,
    // i.e. it also squares through the zero bits of the top byte; squaring the
    // running base does not change `sum`, so the value is identical.
    for (int i = 0; i < top; ++i) {
      if ((e[i / 64] >> (i % 64)) & 1) sum = sum * bit_value;
      bit_value = bit_value * bit_value;
    }
    return sum;
  }

  // cube (:337-344): e*e % p, then *e % p
  Fp cube() const { return ((*this) * (*this)) * (*this); }

  // safe_inv (:379-432): extended Euclid; Err on zero.  Restated as the binary
  // extended Euclid for an odd modulus — the inverse in [0,order) is unique.
  bool safe_inv(Fp& out) const {
    const FieldParams& f = F();
    if (is_zero()) return false;
    const int k = f.k;
    uint64_t u[MAXL + 1] = {0}, v[MAXL + 1] = {0}, x1[MAXL + 1] = {0}, x2[MAXL + 1] = {0}, pp[MAXL + 1] = {0};
    for (int i = 0; i < k; ++i) { u[i] = l[i]; v[i] = f.p[i]; pp[i] = f.p[i]; }
    x1[0] = 1;
    auto is_one = [&](const uint64_t* a) { if (a[0] != 1) return false; for (int i = 1; i <= k; ++i) if (a[i]) return false; return true; };
    auto shr1 = [&](uint64_t* a) { for (int i = 0; i < k; ++i) a[i] = (a[i] >> 1) | (a[i + 1] << 63); a[k] >>= 1; };
    auto halve_mod = [&](uint64_t* x) { if (x[0] & 1) limb_add(x, x, pp, k + 1); shr1(x); };
    auto sub_mod = [&](uint64_t* a, const uint64_t* b) {  // a = a-b mod p, a,b in [0,p)
      if (limb_cmp(a, b, k + 1) < 0) limb_add(a, a, pp, k + 1);
      limb_sub(a, a, b, k + 1);
    };
    if ((f.p[0] & 1) == 0) throw std::runtime_error("oracle: even modulus not supported");
    while (!is_one(u) && !is_one(v)) {
      if (limb_is_zero(u, k + 1) || limb_is_zero(v, k + 1)) return false;  // not coprime
      while ((u[0] & 1) == 0) { shr1(u); halve_mod(x1); }
      while ((v[0] & 1) == 0) { shr1(v); halve_mod(x2); }
      if (limb_cmp(u, v, k + 1) >= 0) { limb_sub(u, u, v, k + 1); sub_mod(x1, x2); }
      else { limb_sub(v, v, u, k + 1); sub_mod(x2, x1); }
    }
    const uint64_t* r = is_one(u) ? x1 : x2;
    for (int i = 0; i < MAXL; ++i) out.l[i] = i < k ? r[i] : 0;
    return true;
  }
  // inv (:434-436) panics on zero
  Fp inv() const {
    Fp r; if (!safe_inv(r)) throw std::domain_error("Cannot find inverse of zero"); return r;
  }
};

typedef Fp<FqTag> Fq1;   // fq1.rs:13
typedef Fp<FrTag> Fr;

// ---------------------------------------------------------------------------
// Fq2 = Fq[u]/(u^2+1), struct order {u1,u0} (fq2.rs:15-19)
// ---------------------------------------------------------------------------
struct Fq2 {
  Fq1 u1, u0;
  Fq2() {}
  Fq2(const Fq1& a1, const Fq1& a0) : u1(a1), u0(a0) {}
  static Fq2 zero() { return Fq2(); }
  static Fq2 from_u64(uint64_t n) { return Fq2(Fq1(), Fq1(n)); }       // fq2.rs:68-73
  bool is_zero() const { return u0.is_zero() && u1.is_zero(); }       // fq2.rs:39-42
  bool operator==(const Fq2& o) const { return u1 == o.u1 && u0 == o.u0; }
  bool operator!=(const Fq2& o) const { return !(*this == o); }
  Fq2 operator+(const Fq2& o) const { return Fq2(u1 + o.u1, u0 + o.u0); }   // :96-110
  Fq2 operator-(const Fq2& o) const { return Fq2(u1 - o.u1, u0 - o.u0); }   // :115-129
  Fq2 operator*(const Fq2& o) const {                                        // :134-146 (schoolbook, 4 Fq mul)
    return Fq2(u1 * o.u0 + u0 * o.u1, u0 * o.u0 - u1 * o.u1);
  }
  Fq2 operator-() const { return Fq2::zero() - *this; }                      // :82-94
  Fq2 inv() const {                                                          // :26-32
    Fq1 factor = (u1 * u1 + u0 * u0).inv();
    return Fq2(u1.negate() * factor, u0 * factor);
  }
  Fq2 sq() const { return (*this) * (*this); }                               // :34-36
  Fq2 reduce() const { return Fq2(u1 + u0, u0 - u1); }                       // x(1+u), :52-58
};

// ---------------------------------------------------------------------------
// Fq6 = Fq2[v]/(v^3-(1+u)), struct order {v2,v1,v0} (fq6.rs:15-20)
// ---------------------------------------------------------------------------
struct Fq6 {
  Fq2 v2, v1, v0;
  Fq6() {}
  Fq6(const Fq2& a2, const Fq2& a1, const Fq2& a0) : v2(a2), v1(a1), v0(a0) {}
  static Fq6 zero() { return Fq6(); }
  static Fq6 from_u64(uint64_t n) { return Fq6(Fq2::zero(), Fq2::zero(), Fq2::from_u64(n)); }  // fq6.rs:82-90
  bool operator==(const Fq6& o) const { return v2 == o.v2 && v1 == o.v1 && v0 == o.v0; }
  Fq6 operator+(const Fq6& o) const { return Fq6(v2 + o.v2, v1 + o.v1, v0 + o.v0); }
  Fq6 operator-(const Fq6& o) const { return Fq6(v2 - o.v2, v1 - o.v1, v0 - o.v0); }
  Fq6 operator-() const { return Fq6::zero() - *this; }                      // fq6.rs:74-80
  Fq6 operator*(const Fq6& o) const {                                        // fq6.rs:148-166 (9 Fq2 mul)
    Fq2 t0 = v0 * o.v0;
    Fq2 t1 = v0 * o.v1 + v1 * o.v0;
    Fq2 t2 = v0 * o.v2 + v1 * o.v1 + v2 * o.v0;
    Fq2 t3 = (v1 * o.v2 + v2 * o.v1).reduce();
    Fq2 t4 = (v2 * o.v2).reduce();
    return Fq6(t2, t1 + t4, t0 + t3);
  }
  Fq6 inv() const {                                                          // fq6.rs:23-37
    Fq2 t0 = v0 * v0 - (v1 * v2).reduce();
    Fq2 t1 = (v2 * v2).reduce() - v0 * v1;
    Fq2 t2 = v1 * v1 - v0 * v2;
    Fq2 factor = (v0 * t0 + (v2 * t1).reduce() + (v1 * t2).reduce()).inv();
    return Fq6(t2 * factor, t1 * factor, t0 * factor);
  }
  Fq6 reduce() const { return Fq6(v1, v0, v2.reduce()); }                    // x v, fq6.rs:54-62
};

// ---------------------------------------------------------------------------
// Fq12 = Fq6[w]/(w^2-v), struct order {w1,w0} (fq12.rs:17-21)
// ---------------------------------------------------------------------------
struct Fq12 {
  Fq6 w1, w0;
  Fq12() {}
  Fq12(const Fq6& a1, const Fq6& a0) : w1(a1), w0(a0) {}
  static Fq12 zero() { return Fq12(); }
  static Fq12 from_u64(uint64_t n) { return Fq12(Fq6::zero(), Fq6::from_u64(n)); }   // fq12.rs:60-67
  static Fq12 from_fq(const Fq1& x) {            // From<&dyn ToBigUint> with an Fq1 argument
    return Fq12(Fq6::zero(), Fq6(Fq2::zero(), Fq2::zero(), Fq2(Fq1(), x)));
  }
  bool operator==(const Fq12& o) const { return w1 == o.w1 && w0 == o.w0; }
  Fq12 operator+(const Fq12& o) const { return Fq12(w1 + o.w1, w0 + o.w0); }
  Fq12 operator-(const Fq12& o) const { return Fq12(w1 - o.w1, w0 - o.w0); }
  Fq12 operator-() const { return Fq12::zero() - *this; }
  Fq12 operator*(const Fq12& o) const {                                      // fq12.rs:135-147 (4 Fq6 mul)
    return Fq12(w1 * o.w0 + w0 * o.w1, w0 * o.w0 + (w1 * o.w1).reduce());
  }
  Fq12 inv() const {                                                         // fq12.rs:31-40
    Fq6 factor = (w0 * w0 - (w1 * w1).reduce()).inv();
    return Fq12((-w1) * factor, w0 * factor);
  }
  // pow (fq12.rs:42-57): LSB-first square-and-multiply, base squared every step
  Fq12 pow_bits(const std::vector<uint32_t>& e /*LE u32 limbs*/) const {
    Fq12 base = *this, acc = Fq12::from_u64(1);
    int top = (int)e.size() * 32;
    while (top > 0 && !((e[(top - 1) / 32] >> ((top - 1) % 32)) & 1)) --top;
    for (int i = 0; i < top; ++i) {
      if ((e[i / 32] >> (i % 32)) & 1) acc = acc * base;
      base = base * base;
    }
    return acc;
  }
};

// ---------------------------------------------------------------------------
// Affine points and the group law (curves/macros.rs:1-163), instantiated for
// G1 over Fq1 (g1_point.rs:154-155) and G2 over Fq2 (g2_point.rs:139-140).
// ---------------------------------------------------------------------------
template <class F>
struct Affine {
  F x, y; bool inf;
  Affine() : inf(true) {}
  Affine(const F& ax, const F& ay) : x(ax), y(ay), inf(false) {}
  static Affine infinity() { return Affine(); }
  bool operator==(const Affine& o) const {               // g1_point.rs:163-175
    if (inf || o.inf) return inf && o.inf;
    return x == o.x && y == o.y;
  }
  Affine neg() const { return inf ? *this : Affine(x, -y); }   // g1_point.rs:177-195
};

// impl_affine_add! (macros.rs:34-163) — case order preserved
template <class F>
Affine<F> affine_add(const Affine<F>& a, const Affine<F>& b) {
  if (a.inf && b.inf) return Affine<F>::infinity();
  if (a.inf) return b;
  if (b.inf) return a;
  if (a.x == b.x && a.y != b.y) return Affine<F>::infinity();
  if (a.x == b.x && a.y == b.y) {
    if (a.y.is_zero()) return Affine<F>::infinity();
    F x1_sq = a.x.sq();
    F m1 = x1_sq + x1_sq + x1_sq;
    F m2 = a.y + a.y;
    F m = m1 * m2.inv();
    F p3x = m.sq() - (a.x + a.x);
    F p3y_neg = m * (a.x - p3x) - a.y;
    return Affine<F>(p3x, p3y_neg);
  }
  F m = (b.y - a.y) * (b.x - a.x).inv();
  F p3x = m.sq() - a.x - b.x;
  F p3y = m * (p3x - a.x) + a.y;
  return Affine<F>(p3x, -p3y);
}

// impl_scalar_mul_point! (macros.rs:1-32): LSB-first double-and-add; the
// scalar's stored integer is used as-is (no reduction mod r).
template <class F>
Affine<F> scalar_mul(const Affine<F>& pt, const uint64_t* n, int nlimbs) {
  Affine<F> res = Affine<F>::infinity();
  Affine<F> pt_pow_n = pt;
  int top = nlimbs * 64;
  while (top > 0 && !((n[(top - 1) / 64] >> ((top - 1) % 64)) & 1)) --top;
  for (int i = 0; i < top; ++i) {
    if ((n[i / 64] >> (i % 64)) & 1) res = affine_add(res, pt_pow_n);
    pt_pow_n = affine_add(pt_pow_n, pt_pow_n);
  }
  return res;
}

typedef Affine<Fq1> G1Point;
typedef Affine<Fq2> G2Point;
typedef Affine<Fp<SpTag>> SecpPoint;   // secp256k1/affine_point.rs:146-147

// Polynomial::eval_with_g{1,2}_hidings (field/polynomial.rs:271-293): the
// crate's MSM, strictly sequential.
template <class F>
Affine<F> msm_naive(const Affine<F>* powers, const uint64_t* coeffs, int coeff_limbs, size_t n) {
  Affine<F> sum = Affine<F>::infinity();
  for (size_t i = 0; i < n; ++i)
    sum = affine_add(sum, scalar_mul(powers[i], coeffs + i * coeff_limbs, coeff_limbs));
  return sum;
}

G1Point g1_generator();   // g1_point.rs:38-47
G2Point g2_generator();   // g2_point.rs:36-46
SecpPoint secp_generator();

// ---------------------------------------------------------------------------
// G12Point, RationalFunction, Pairing (g12_point.rs, rational_function.rs,
// pairing.rs:15-100)
// ---------------------------------------------------------------------------
struct G12Point { Fq12 x, y; bool inf; };

G12Point g12_from_g1(const G1Point& p);          // g12_point.rs:29-44
G12Point g12_from_g2(const G2Point& p);          // untwist, g12_point.rs:47-68

struct RationalFunction {                         // rational_function.rs:11-18
  bool vertical; Fq12 x, y, slope;
  static RationalFunction new_g1(const G1Point& p, const G1Point& q);
  static RationalFunction new_g2(const G2Point& p, const G2Point& q);
  Fq12 eval_with_g1(const G1Point& q) const;
  Fq12 eval_with_g2(const G2Point& q) const;
};

struct Pairing {
  std::vector<bool> l_bits;                       // pairing.rs:58-73
  std::vector<uint32_t> final_exp;                // (q^12-1)/r, pairing.rs:94-97
  Pairing();
  Fq12 calc_g1_g2(const G1Point& p, const G2Point& q) const;   // pairing.rs:54
  Fq12 calc_g2_g1(const G2Point& p, const G1Point& q) const;   // pairing.rs:55
  Fq12 weil(const G1Point& p, const G2Point& q) const;         // pairing.rs:75-84
  Fq12 tate(const G1Point& p, const G2Point& q) const;         // pairing.rs:86-100
};

void init_fields();   // idempotent

}  // namespace zkto
