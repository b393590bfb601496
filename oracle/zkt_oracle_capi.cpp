// TEST INFRASTRUCTURE — NOT PRODUCT CODE.  C entry points of the CPU oracle
// for ctypes (tests/, __graft_entry__.smoke(), bench.py cpu_baseline only).
// Buffer layouts are those of include/zkt.h so the same numpy arrays can be
// handed to the oracle and to the HIP library.
#include "zkt_oracle.hpp"
#include <thread>
#include <atomic>

using namespace zkto;

namespace {
const int FQ = 6, FR = 4;                 // u64 limbs
const int G1W = 13, G2W = 25;             // u64 words per point incl. infinity word
const int FQ2 = 12, FQ6 = 36, FQ12 = 72;

enum { OK = 0, ERR_INV_ZERO = 1, ERR_INFINITY = 2, ERR_SHAPE = 3 };

template <class T> T ld(const uint64_t* p, int n) { return T::from_limbs(p, n); }
template <class T> void st(uint64_t* p, const T& v, int n) { for (int i = 0; i < n; ++i) p[i] = v.l[i]; }

Fq2 ld2(const uint64_t* p) { return Fq2(ld<Fq1>(p, FQ), ld<Fq1>(p + FQ, FQ)); }              // {u1,u0}
void st2(uint64_t* p, const Fq2& v) { st(p, v.u1, FQ); st(p + FQ, v.u0, FQ); }
Fq6 ld6(const uint64_t* p) { return Fq6(ld2(p), ld2(p + FQ2), ld2(p + 2 * FQ2)); }            // {v2,v1,v0}
void st6(uint64_t* p, const Fq6& v) { st2(p, v.v2); st2(p + FQ2, v.v1); st2(p + 2 * FQ2, v.v0); }
Fq12 ld12(const uint64_t* p) { return Fq12(ld6(p), ld6(p + FQ6)); }                            // {w1,w0}
void st12(uint64_t* p, const Fq12& v) { st6(p, v.w1); st6(p + FQ6, v.w0); }

G1Point ldg1(const uint64_t* p) { if (p[12] & 0xffffffffu) return G1Point::infinity(); return G1Point(ld<Fq1>(p, FQ), ld<Fq1>(p + FQ, FQ)); }
void stg1(uint64_t* p, const G1Point& v) {
  if (v.inf) { for (int i = 0; i < 12; ++i) p[i] = 0; p[12] = 1; return; }
  st(p, v.x, FQ); st(p + FQ, v.y, FQ); p[12] = 0;
}
G2Point ldg2(const uint64_t* p) { if (p[24] & 0xffffffffu) return G2Point::infinity(); return G2Point(ld2(p), ld2(p + FQ2)); }
void stg2(uint64_t* p, const G2Point& v) {
  if (v.inf) { for (int i = 0; i < 24; ++i) p[i] = 0; p[24] = 1; return; }
  st2(p, v.x); st2(p + FQ2, v.y); p[24] = 0;
}
typedef Fp<SpTag> Sp;
SecpPoint ldsp(const uint64_t* p) { if (p[8] & 0xffffffffu) return SecpPoint::infinity(); return SecpPoint(ld<Sp>(p, 4), ld<Sp>(p + 4, 4)); }
void stsp(uint64_t* p, const SecpPoint& v) {
  if (v.inf) { for (int i = 0; i < 8; ++i) p[i] = 0; p[8] = 1; return; }
  st(p, v.x, 4); st(p + 4, v.y, 4); p[8] = 0;
}

// run f(i) for i in [0,n) on `threads` std::threads (independent items)
template <class Fn> void par_for(size_t n, int threads, Fn f) {
  if (threads <= 1 || n < 2) { for (size_t i = 0; i < n; ++i) f(i); return; }
  std::atomic<size_t> next(0);
  std::vector<std::thread> ts;
  for (int t = 0; t < threads; ++t) ts.emplace_back([&] { for (;;) { size_t i = next++; if (i >= n) break; f(i); } });
  for (auto& t : ts) t.join();
}

template <class T> int field_binop(int op, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n, int w) {
  init_fields();
  for (size_t i = 0; i < n; ++i) {
    T x = ld<T>(a + i * w, w), y = ld<T>(b + i * w, w), r;
    switch (op) { case 0: r = x + y; break; case 1: r = x - y; break; default: r = x * y; }
    st(o + i * w, r, w);
  }
  return OK;
}
template <class T> int field_unop(int op, const uint64_t* a, uint64_t* o, size_t n, int w, size_t* err_index) {
  init_fields();
  for (size_t i = 0; i < n; ++i) {
    T x = ld<T>(a + i * w, w), r;
    switch (op) {
      case 0: r = x.sq(); break;
      case 1: r = x.negate(); break;
      default: if (!x.safe_inv(r)) { if (err_index) *err_index = i; return ERR_INV_ZERO; }
    }
    st(o + i * w, r, w);
  }
  return OK;
}
// a1-a3 over any of the four fields the reference instantiates (field: 0 Fq, 1 Fr, 2 secp256k1 p, 3 secp256k1 n).
// op: 0 plus 1 minus 2 times 3 sq 4 negate 5 inv 6 cube (prime_field_elem.rs:278-344,379-457)
template <class T> int field_op_t(int op, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n, int w, size_t* err_index) {
  for (size_t i = 0; i < n; ++i) {
    T x = ld<T>(a + i * w, w), y = b ? ld<T>(b + i * w, w) : T(), r;
    switch (op) {
      case 0: r = x + y; break; case 1: r = x - y; break; case 2: r = x * y; break; case 3: r = x.sq(); break; case 4: r = x.negate(); break;
      case 5: if (!x.safe_inv(r)) { if (err_index) *err_index = i; return ERR_INV_ZERO; } break;
      case 6: r = x.cube(); break;
      default: return ERR_SHAPE;
    }
    st(o + i * w, r, w);
  }
  return OK;
}
// pow (prime_field_elem.rs:311-328): out[i] = a[i] ^ e_i, exponents of e_limbs u64 limbs each (exp_shared: one exponent for all)
template <class T> int field_pow_t(const uint64_t* a, const uint64_t* e, int e_limbs, int shared, uint64_t* o, size_t n, int w) {
  for (size_t i = 0; i < n; ++i) st(o + i * w, ld<T>(a + i * w, w).pow_limbs(e + (shared ? 0 : i * e_limbs), e_limbs), w);
  return OK;
}
// pow_seq (prime_field_elem.rs:346-361): 1, x, x^2, ... by the running product x = x * self.e;  repeat (:363-376): n clones
template <class T> int field_pow_seq_t(const uint64_t* a, size_t n, uint64_t* o, int w, bool repeat) {
  T base = ld<T>(a, w), x(1);
  for (size_t i = 0; i < n; ++i) { st(o + i * w, repeat ? base : x, w); x = x * base; }
  return OK;
}
}  // namespace

extern "C" {

int zkto_version() { return 1; }

// --- a1/a2/a3: prime-field ops on canonical residues ------------------------
int zkto_fq_add_batch(const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) { return field_binop<Fq1>(0, a, b, o, n, FQ); }
int zkto_fq_sub_batch(const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) { return field_binop<Fq1>(1, a, b, o, n, FQ); }
int zkto_fq_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) { return field_binop<Fq1>(2, a, b, o, n, FQ); }
int zkto_fq_sqr_batch(const uint64_t* a, uint64_t* o, size_t n) { return field_unop<Fq1>(0, a, o, n, FQ, nullptr); }
int zkto_fq_neg_batch(const uint64_t* a, uint64_t* o, size_t n) { return field_unop<Fq1>(1, a, o, n, FQ, nullptr); }
int zkto_fq_inv_batch(const uint64_t* a, uint64_t* o, size_t n, size_t* err_index) { return field_unop<Fq1>(2, a, o, n, FQ, err_index); }

int zkto_fr_add_batch(const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) { return field_binop<Fr>(0, a, b, o, n, FR); }
int zkto_fr_sub_batch(const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) { return field_binop<Fr>(1, a, b, o, n, FR); }
int zkto_fr_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) { return field_binop<Fr>(2, a, b, o, n, FR); }
int zkto_fr_sqr_batch(const uint64_t* a, uint64_t* o, size_t n) { return field_unop<Fr>(0, a, o, n, FR, nullptr); }
int zkto_fr_neg_batch(const uint64_t* a, uint64_t* o, size_t n) { return field_unop<Fr>(1, a, o, n, FR, nullptr); }
int zkto_fr_inv_batch(const uint64_t* a, uint64_t* o, size_t n, size_t* err_index) { return field_unop<Fr>(2, a, o, n, FR, err_index); }

// generic-modulus slot for the small-modulus reference KATs (prime_field_elem.rs:465-964)
int zkto_dyn_set_modulus(const uint64_t* m, int nlimbs) { init_fields(); if (nlimbs < 1 || nlimbs > MAXL) return ERR_SHAPE; DynTag::P.init_from_limbs(m, nlimbs); return OK; }
int zkto_dyn_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* o) {
  // operands are MAXL limbs; op: 0 add 1 sub 2 mul 3 inv 4 neg 5 new(reduce) 6 pow(b = exponent)
  typedef Fp<DynTag> D;
  init_fields();
  D x = D::from_limbs(a, MAXL), y = b ? D::from_limbs(b, MAXL) : D(), r;
  switch (op) {
    case 0: r = x + y; break; case 1: r = x - y; break; case 2: r = x * y; break;
    case 3: if (!x.safe_inv(r)) return ERR_INV_ZERO; break;
    case 4: r = x.negate(); break; case 5: r = x; break;
    case 6: r = x.pow_limbs(b, MAXL); break;
    default: return ERR_SHAPE;
  }
  for (int i = 0; i < MAXL; ++i) o[i] = r.l[i];
  return OK;
}
int zkto_fq_pow(const uint64_t* a, const uint64_t* e, int e_limbs, uint64_t* o) {
  init_fields(); Fq1 r = ld<Fq1>(a, FQ).pow_limbs(e, e_limbs); st(o, r, FQ); return OK;
}

int zkto_field_op(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n, size_t* err_index) {
  init_fields();
  switch (field) {
    case 0: return field_op_t<Fq1>(op, a, b, o, n, FQ, err_index);
    case 1: return field_op_t<Fr>(op, a, b, o, n, FR, err_index);
    case 2: return field_op_t<Sp>(op, a, b, o, n, 4, err_index);
    case 3: return field_op_t<Fp<SnTag>>(op, a, b, o, n, 4, err_index);
  }
  return ERR_SHAPE;
}
int zkto_field_pow_batch(int field, const uint64_t* a, const uint64_t* e, int e_limbs, int shared, uint64_t* o, size_t n) {
  init_fields();
  switch (field) {
    case 0: return field_pow_t<Fq1>(a, e, e_limbs, shared, o, n, FQ);
    case 1: return field_pow_t<Fr>(a, e, e_limbs, shared, o, n, FR);
    case 2: return field_pow_t<Sp>(a, e, e_limbs, shared, o, n, 4);
    case 3: return field_pow_t<Fp<SnTag>>(a, e, e_limbs, shared, o, n, 4);
  }
  return ERR_SHAPE;
}
int zkto_field_pow_seq(int field, const uint64_t* a, size_t n, uint64_t* o, int repeat) {
  init_fields();
  switch (field) {
    case 0: return field_pow_seq_t<Fq1>(a, n, o, FQ, repeat != 0);
    case 1: return field_pow_seq_t<Fr>(a, n, o, FR, repeat != 0);
    case 2: return field_pow_seq_t<Sp>(a, n, o, 4, repeat != 0);
    case 3: return field_pow_seq_t<Fp<SnTag>>(a, n, o, 4, repeat != 0);
  }
  return ERR_SHAPE;
}

// --- a4-a6: tower -------------------------------------------------------------
// op: 0 add 1 sub 2 mul 3 inv 4 neg 5 reduce 6 sq
int zkto_fq2_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) {
  init_fields();
  for (size_t i = 0; i < n; ++i) {
    Fq2 x = ld2(a + i * FQ2), y = b ? ld2(b + i * FQ2) : Fq2(), r;
    switch (op) {
      case 0: r = x + y; break; case 1: r = x - y; break; case 2: r = x * y; break;
      case 3: if ((x.u1 * x.u1 + x.u0 * x.u0).is_zero()) return ERR_INV_ZERO; r = x.inv(); break;
      case 4: r = -x; break; case 5: r = x.reduce(); break; case 6: r = x.sq(); break;
      default: return ERR_SHAPE;
    }
    st2(o + i * FQ2, r);
  }
  return OK;
}
int zkto_fq6_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) {
  init_fields();
  try {
    for (size_t i = 0; i < n; ++i) {
      Fq6 x = ld6(a + i * FQ6), y = b ? ld6(b + i * FQ6) : Fq6(), r;
      switch (op) {
        case 0: r = x + y; break; case 1: r = x - y; break; case 2: r = x * y; break;
        case 3: r = x.inv(); break; case 4: r = -x; break; case 5: r = x.reduce(); break;
        default: return ERR_SHAPE;
      }
      st6(o + i * FQ6, r);
    }
  } catch (const std::domain_error&) { return ERR_INV_ZERO; }
  return OK;
}
int zkto_fq12_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) {
  init_fields();
  try {
    for (size_t i = 0; i < n; ++i) {
      Fq12 x = ld12(a + i * FQ12), y = b ? ld12(b + i * FQ12) : Fq12(), r;
      switch (op) {
        case 0: r = x + y; break; case 1: r = x - y; break; case 2: r = x * y; break;
        case 3: r = x.inv(); break; case 4: r = -x; break;
        default: return ERR_SHAPE;
      }
      st12(o + i * FQ12, r);
    }
  } catch (const std::domain_error&) { return ERR_INV_ZERO; }
  return OK;
}
// Fq12::pow (fq12.rs:42-57); exponent as little-endian u32 limbs
int zkto_fq12_pow(const uint64_t* a, const uint32_t* e, size_t e_nlimbs, uint64_t* o) {
  init_fields();
  std::vector<uint32_t> ev(e, e + e_nlimbs);
  st12(o, ld12(a).pow_bits(ev)); return OK;
}

// --- a7-a9, a16: groups -------------------------------------------------------
void zkto_g1_generator(uint64_t* o) { init_fields(); stg1(o, g1_generator()); }
void zkto_g2_generator(uint64_t* o) { init_fields(); stg2(o, g2_generator()); }
void zkto_secp_generator(uint64_t* o) { init_fields(); stsp(o, secp_generator()); }

int zkto_g1_add_batch(const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) {
  init_fields(); for (size_t i = 0; i < n; ++i) stg1(o + i * G1W, affine_add(ldg1(a + i * G1W), ldg1(b + i * G1W))); return OK;
}
int zkto_g1_neg_batch(const uint64_t* a, uint64_t* o, size_t n) {
  init_fields(); for (size_t i = 0; i < n; ++i) stg1(o + i * G1W, ldg1(a + i * G1W).neg()); return OK;
}
int zkto_g1_mul_batch(const uint64_t* pts, const uint64_t* scalars, int scalar_limbs, uint64_t* o, size_t n, int threads) {
  init_fields();
  par_for(n, threads, [&](size_t i) { stg1(o + i * G1W, scalar_mul(ldg1(pts + i * G1W), scalars + i * scalar_limbs, scalar_limbs)); });
  return OK;
}
// Polynomial::eval_with_g1_hidings (polynomial.rs:271-281)
int zkto_g1_msm(const uint64_t* bases, const uint64_t* scalars, int scalar_limbs, size_t n, uint64_t* o) {
  init_fields();
  std::vector<G1Point> ps(n); for (size_t i = 0; i < n; ++i) ps[i] = ldg1(bases + i * G1W);
  stg1(o, msm_naive(ps.data(), scalars, scalar_limbs, n)); return OK;
}
int zkto_g2_add_batch(const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) {
  init_fields(); for (size_t i = 0; i < n; ++i) stg2(o + i * G2W, affine_add(ldg2(a + i * G2W), ldg2(b + i * G2W))); return OK;
}
int zkto_g2_neg_batch(const uint64_t* a, uint64_t* o, size_t n) {
  init_fields(); for (size_t i = 0; i < n; ++i) stg2(o + i * G2W, ldg2(a + i * G2W).neg()); return OK;
}
int zkto_g2_mul_batch(const uint64_t* pts, const uint64_t* scalars, int scalar_limbs, uint64_t* o, size_t n, int threads) {
  init_fields();
  par_for(n, threads, [&](size_t i) { stg2(o + i * G2W, scalar_mul(ldg2(pts + i * G2W), scalars + i * scalar_limbs, scalar_limbs)); });
  return OK;
}
int zkto_g2_msm(const uint64_t* bases, const uint64_t* scalars, int scalar_limbs, size_t n, uint64_t* o) {
  init_fields();
  std::vector<G2Point> ps(n); for (size_t i = 0; i < n; ++i) ps[i] = ldg2(bases + i * G2W);
  stg2(o, msm_naive(ps.data(), scalars, scalar_limbs, n)); return OK;
}
int zkto_g2_is_on_curve(const uint64_t* p) {   // g2_point.rs:76-81
  init_fields(); G2Point q = ldg2(p); if (q.inf) return 0;
  Fq2 lhs = q.y * q.y; Fq2 rhs = q.x * q.x * q.x + Fq2::from_u64(4).reduce(); return lhs == rhs;
}
int zkto_g1_is_on_curve(const uint64_t* p) {   // g1_point.rs:95-110 (y^2 = x^3 + 4)
  init_fields(); G1Point q = ldg1(p); if (q.inf) return 0;
  return q.y * q.y == q.x * q.x * q.x + Fq1(4);
}
// secp256k1 (affine_point.rs:146-147; affine_points.rs)
int zkto_secp_add_batch(const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) {
  init_fields(); for (size_t i = 0; i < n; ++i) stsp(o + i * 9, affine_add(ldsp(a + i * 9), ldsp(b + i * 9))); return OK;
}
int zkto_secp_mul_batch(const uint64_t* pts, const uint64_t* scalars, int scalar_limbs, uint64_t* o, size_t n, int threads) {
  init_fields();
  par_for(n, threads, [&](size_t i) { stsp(o + i * 9, scalar_mul(ldsp(pts + i * 9), scalars + i * scalar_limbs, scalar_limbs)); });
  return OK;
}

// --- a10-a15: pairing ------------------------------------------------------------
static const Pairing& pairing() { static Pairing p; return p; }

int zkto_pairing_params(uint32_t* l_bits /*>=254*/, size_t* n_l_bits, uint32_t* final_exp /*>=135*/, size_t* n_final_exp) {
  const Pairing& p = pairing();
  for (size_t i = 0; i < p.l_bits.size(); ++i) l_bits[i] = p.l_bits[i];
  *n_l_bits = p.l_bits.size();
  for (size_t i = 0; i < p.final_exp.size(); ++i) final_exp[i] = p.final_exp[i];
  *n_final_exp = p.final_exp.size();
  return OK;
}
// which: 0 calc_g1_g2, 1 calc_g2_g1, 2 weil, 3 tate
int zkto_pairing_batch(int which, const uint64_t* g1, const uint64_t* g2, uint64_t* o, size_t n, int threads, size_t* err_index) {
  const Pairing& pr = pairing();
  std::atomic<long> bad(-1);
  par_for(n, threads, [&](size_t i) {
    try {
      G1Point p = ldg1(g1 + i * G1W); G2Point q = ldg2(g2 + i * G2W); Fq12 r;
      switch (which) {
        case 0: r = pr.calc_g1_g2(p, q); break;
        case 1: r = pr.calc_g2_g1(q, p); break;
        case 2: r = pr.weil(p, q); break;
        default: r = pr.tate(p, q);
      }
      st12(o + i * FQ12, r);
    } catch (const std::domain_error&) { long e = -1; bad.compare_exchange_strong(e, (long)i); }
  });
  if (bad >= 0) { if (err_index) *err_index = (size_t)bad.load(); return ERR_INFINITY; }
  return OK;
}
int zkto_untwist(const uint64_t* g2, uint64_t* x12, uint64_t* y12) {   // g12_point.rs:47-68
  init_fields(); G12Point p = g12_from_g2(ldg2(g2)); if (p.inf) return ERR_INFINITY; st12(x12, p.x); st12(y12, p.y); return OK;
}

}  // extern "C"
