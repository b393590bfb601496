// TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Oracle restatement of the two callers of the hot path:
//   Groth16  src/zk/w_trusted_setup/groth16/zktoolkit_based/{crs.rs:49-146, prover.rs:96-147, verifier.rs:30-54}
//   Bulletproofs inner-product argument  src/zk/wo_trusted_setup/bulletproofs.rs:19-55 (secp256k1)
// The reference samples every random value from OS entropy (prime_field.rs:73-85, random_number.rs:8-13);
// here they are INJECTED (alpha,beta,gamma,delta,x,r,s and the IPA challenges) so that results are comparable.
// QAP polynomials arrive as dense coefficient arrays (low degree first), n coefficients each; the reference's
// Polynomial trims trailing zeros, which only removes terms that contribute the point at infinity.
#include "zkt_oracle.hpp"
using namespace zkto;

namespace {
const int FQ = 6, FR = 4, G1W = 13, G2W = 25, FQ12 = 72;
Fr ldr(const uint64_t* p) { return Fr::from_limbs(p, FR); }
Fq2 ld2(const uint64_t* p) { return Fq2(Fq1::from_limbs(p, FQ), Fq1::from_limbs(p + FQ, FQ)); }
void st2(uint64_t* p, const Fq2& v) { for (int i = 0; i < FQ; ++i) { p[i] = v.u1.l[i]; p[FQ + i] = v.u0.l[i]; } }
G1Point ldg1(const uint64_t* p) { if (p[12] & 0xffffffffu) return G1Point::infinity(); return G1Point(Fq1::from_limbs(p, FQ), Fq1::from_limbs(p + FQ, FQ)); }
void stg1(uint64_t* p, const G1Point& v) {
  if (v.inf) { for (int i = 0; i < 12; ++i) p[i] = 0; p[12] = 1; return; }
  for (int i = 0; i < FQ; ++i) { p[i] = v.x.l[i]; p[FQ + i] = v.y.l[i]; } p[12] = 0;
}
G2Point ldg2(const uint64_t* p) { if (p[24] & 0xffffffffu) return G2Point::infinity(); return G2Point(ld2(p), ld2(p + 12)); }
void stg2(uint64_t* p, const G2Point& v) {
  if (v.inf) { for (int i = 0; i < 24; ++i) p[i] = 0; p[24] = 1; return; }
  st2(p, v.x); st2(p + 12, v.y); p[24] = 0;
}
void st12(uint64_t* p, const Fq12& v) {
  const Fq6* six[2] = {&v.w1, &v.w0};
  for (int s = 0; s < 2; ++s) { const Fq2* c[3] = {&six[s]->v2, &six[s]->v1, &six[s]->v0}; for (int k = 0; k < 3; ++k) st2(p + s * 36 + k * 12, *c[k]); }
}
Fq12 ld12(const uint64_t* p) {
  Fq6 s[2]; for (int k = 0; k < 2; ++k) s[k] = Fq6(ld2(p + k * 36), ld2(p + k * 36 + 12), ld2(p + k * 36 + 24));
  return Fq12(s[0], s[1]);
}
template <class P> P mul_fr(const P& p, const Fr& k) { return scalar_mul(p, k.l, FR); }   // G * Fq1-like scalar (macros.rs:1-32)

// Polynomial::eval_at (field/polynomial.rs:240-249)
Fr poly_eval(const uint64_t* coeffs, size_t n, const Fr& x) {
  Fr acc(0), xp(1);
  for (size_t i = 0; i < n; ++i) { acc = acc + ldr(coeffs + i * FR) * xp; xp = xp * x; }
  return acc;
}
// QAP::build_t(f, n).eval_at(x): t = prod_{i=1..n} (x - i)   (qap/qap.rs:115-135)
Fr t_eval(size_t n, const Fr& x) { Fr t(1); for (size_t i = 1; i <= n; ++i) t = t * (x - Fr((uint64_t)i)); return t; }
}  // namespace

struct zkto_groth16_crs {   // same field order as zkt_groth16_crs in include/zkt.h
  size_t n, l, m;
  uint64_t *g1_alpha, *g1_beta, *g1_delta, *g1_xi, *g1_uvw_stmt, *g1_uvw_wit, *g1_xt_by_delta;
  uint64_t *g2_beta, *g2_gamma, *g2_delta, *g2_xi;
  uint64_t* gt_alpha_beta;
};

extern "C" {

// CRS::new (crs.rs:49-146) with the trapdoors injected
int zkto_groth16_setup(zkto_groth16_crs* c, const uint64_t* ui, const uint64_t* vi, const uint64_t* wi,
                       const uint64_t* alpha_, const uint64_t* beta_, const uint64_t* gamma_, const uint64_t* delta_, const uint64_t* x_) {
  init_fields();
  const size_t n = c->n, l = c->l, m = c->m;
  G1Point g = g1_generator(); G2Point h = g2_generator();
  Fr alpha = ldr(alpha_), beta = ldr(beta_), gamma = ldr(gamma_), delta = ldr(delta_), x = ldr(x_);
  auto uvw_div = [&](size_t from, size_t to, const Fr& div, uint64_t* out) {     // calc_uvw_div! crs.rs:65-83
    for (size_t i = from; i <= to; ++i) {
      Fr u = beta * poly_eval(ui + i * n * FR, n, x);
      Fr v = alpha * poly_eval(vi + i * n * FR, n, x);
      Fr w = poly_eval(wi + i * n * FR, n, x);
      Fr y = (u + v + w) * div;
      stg1(out + (i - from) * G1W, mul_fr(g, y));
    }
  };
  uvw_div(0, l, gamma.inv(), c->g1_uvw_stmt);
  uvw_div(l + 1, m, delta.inv(), c->g1_uvw_wit);
  Fr xp(1);
  for (size_t k = 0; k < n; ++k) { stg1(c->g1_xi + k * G1W, mul_fr(g, xp)); xp = xp * x; }          // calc_n_pows! crs.rs:88-104
  Fr t = t_eval(n, x); xp = Fr(1);
  for (size_t k = 0; k < n; ++k) { stg1(c->g1_xt_by_delta + k * G1W, mul_fr(g, xp * t * delta.inv())); xp = xp * x; }   // crs.rs:106-116
  stg1(c->g1_alpha, mul_fr(g, alpha)); stg1(c->g1_beta, mul_fr(g, beta)); stg1(c->g1_delta, mul_fr(g, delta));
  xp = Fr(1);
  for (size_t k = 0; k < n; ++k) { stg2(c->g2_xi + k * G2W, mul_fr(h, xp)); xp = xp * x; }
  stg2(c->g2_beta, mul_fr(h, beta)); stg2(c->g2_gamma, mul_fr(h, gamma)); stg2(c->g2_delta, mul_fr(h, delta));
  static Pairing pr;
  st12(c->gt_alpha_beta, pr.tate(ldg1(c->g1_alpha), ldg2(c->g2_beta)));                               // crs.rs:137-139
  return 0;
}

// Prover::prove (prover.rs:96-147) with r, s injected.  literal != 0: the reference's loop — per wire three MSMs, each
// followed by a scalar multiplication by a_i; literal == 0: the same group elements via one MSM per output
// (A = alpha + (sum a_i u_i)(x) G + r delta), used for sizes where the literal loop would take hours.
int zkto_groth16_prove(const zkto_groth16_crs* c, const uint64_t* ui, const uint64_t* vi, const uint64_t* wires,
                       const uint64_t* hcoef, size_t h_len, const uint64_t* r_, const uint64_t* s_, int literal,
                       uint64_t* A_, uint64_t* B_, uint64_t* C_) {
  init_fields();
  const size_t n = c->n, l = c->l, m = c->m;
  Fr r = ldr(r_), s = ldr(s_);
  std::vector<G1Point> xi1(n), xtd(n); std::vector<G2Point> xi2(n);
  for (size_t k = 0; k < n; ++k) { xi1[k] = ldg1(c->g1_xi + k * G1W); xi2[k] = ldg2(c->g2_xi + k * G2W); xtd[k] = ldg1(c->g1_xt_by_delta + k * G1W); }
  G1Point sumA = G1Point::infinity(), sumB1 = G1Point::infinity(); G2Point sumB = G2Point::infinity();
  if (literal) {
    for (size_t i = 0; i <= m; ++i) {                                          // prover.rs:107-117
      Fr ai = ldr(wires + i * FR);
      G1Point up = mul_fr(msm_naive(xi1.data(), ui + i * n * FR, FR, n), ai);
      G2Point vp = mul_fr(msm_naive(xi2.data(), vi + i * n * FR, FR, n), ai);
      G1Point vp1 = mul_fr(msm_naive(xi1.data(), vi + i * n * FR, FR, n), ai);
      sumA = affine_add(sumA, up); sumB = affine_add(sumB, vp); sumB1 = affine_add(sumB1, vp1);
    }
  } else {
    std::vector<uint64_t> U(n * FR), V(n * FR);
    for (size_t k = 0; k < n; ++k) {
      Fr u(0), v(0);
      for (size_t i = 0; i <= m; ++i) { Fr ai = ldr(wires + i * FR); u = u + ai * ldr(ui + (i * n + k) * FR); v = v + ai * ldr(vi + (i * n + k) * FR); }
      for (int j = 0; j < FR; ++j) { U[k * FR + j] = u.l[j]; V[k * FR + j] = v.l[j]; }
    }
    sumA = msm_naive(xi1.data(), U.data(), FR, n); sumB = msm_naive(xi2.data(), V.data(), FR, n); sumB1 = msm_naive(xi1.data(), V.data(), FR, n);
  }
  G1Point d1 = ldg1(c->g1_delta);
  G1Point A = affine_add(affine_add(ldg1(c->g1_alpha), sumA), mul_fr(d1, r));                 // prover.rs:118
  G2Point B = affine_add(affine_add(ldg2(c->g2_beta), sumB), mul_fr(ldg2(c->g2_delta), s));    // :119
  G1Point B1 = affine_add(affine_add(ldg1(c->g1_beta), sumB1), mul_fr(d1, s));                 // :120
  G1Point sum = G1Point::infinity();
  for (size_t i = l + 1; i <= m; ++i) sum = affine_add(sum, mul_fr(ldg1(c->g1_uvw_wit + (i - l - 1) * G1W), ldr(wires + i * FR)));   // :127-131
  G1Point ht = msm_naive(xtd.data(), hcoef, FR, h_len);                                          // :133
  G1Point C = affine_add(affine_add(affine_add(affine_add(sum, ht), mul_fr(A, s)), mul_fr(B1, r)), mul_fr(mul_fr(d1, r), s).neg());   // :135-140
  stg1(A_, A); stg2(B_, B); stg1(C_, C);
  return 0;
}

// Verifier::verify (verifier.rs:30-54); returns 1/0, -2 if a pairing argument is at infinity (the reference panics)
int zkto_groth16_verify(const zkto_groth16_crs* c, const uint64_t* A_, const uint64_t* B_, const uint64_t* C_,
                        const uint64_t* stmt_wires, size_t n_stmt) {
  init_fields();
  static Pairing pr;
  try {
    Fq12 lhs = pr.tate(ldg1(A_), ldg2(B_));
    G1Point sum = G1Point::infinity();
    for (size_t i = 0; i < n_stmt; ++i) sum = affine_add(sum, mul_fr(ldg1(c->g1_uvw_stmt + i * G1W), ldr(stmt_wires + i * FR)));
    Fq12 rhs = ld12(c->gt_alpha_beta) * pr.tate(sum, ldg2(c->g2_gamma)) * pr.tate(ldg1(C_), ldg2(c->g2_delta));
    return lhs == rhs ? 1 : 0;
  } catch (const std::domain_error&) { return -2; }
}

// ---- Pinocchio (protocol 2 of eprint 2013/279): src/zk/w_trusted_setup/pinocchio/{crs.rs:49-161, prover.rs:98-170, verifier.rs:31-85}
// Random values injected: rnd = r_v, r_w, alpha_v, alpha_w, alpha_y, beta, gamma, s (crs.rs:58-64,82); prover: delta_v, delta_y (prover.rs:104-105).
// vi/wi/yi: (n_io + n_mid) dense polynomials of n coefficients (Prover.vi/wi/yi, prover.rs:43-45); wires 0..n_io-1 are the io part
// (witness.rs:20-23), the rest the mid part (witness.rs:25-27).
struct zkto_pinocchio_crs {    // same field order as zkt_pinocchio_crs in include/zkt.h
  size_t n, n_io, n_mid, max_degree;
  uint64_t *vk_mid, *g1_wk_mid, *g2_wk_mid, *yk_mid, *alpha_vk_mid, *alpha_wk_mid, *alpha_yk_mid, *si, *beta_vwy_k_mid;       // EvaluationKeys crs.rs:12-22
  uint64_t *one_g1, *one_g2, *alpha_v, *alpha_w, *alpha_y, *gamma, *beta_gamma, *t, *vk_io, *wk_io, *yk_io, *alpha_v_t, *alpha_y_t, *beta_t;   // VerificationKeys crs.rs:24-39
};
struct zkto_pinocchio_proof { uint64_t *v_mid_s, *g1_w_mid_s, *g2_w_mid_s, *y_mid_s, *h_s, *alpha_v_mid_s, *alpha_w_mid_s, *alpha_y_mid_s, *beta_vwy_mid_s; };   // proof.rs:6-17

int zkto_pinocchio_setup(zkto_pinocchio_crs* c, const uint64_t* vi, const uint64_t* wi, const uint64_t* yi, const uint64_t* rnd) {
  init_fields();
  const size_t n = c->n, nio = c->n_io, nmid = c->n_mid;
  G1Point g1 = g1_generator(); G2Point g2 = g2_generator();
  Fr r_v = ldr(rnd), r_w = ldr(rnd + 4), alpha_v = ldr(rnd + 8), alpha_w = ldr(rnd + 12), alpha_y = ldr(rnd + 16), beta = ldr(rnd + 20), gamma = ldr(rnd + 24), s = ldr(rnd + 28);
  Fr r_y = r_v * r_w;                                                                        // crs.rs:67
  G1Point g1_v = mul_fr(g1, r_v), g1_w = mul_fr(g1, r_w), g1_y = mul_fr(g1, r_y); G2Point g2_w = mul_fr(g2, r_w);   // :68-71
  for (size_t k = 0; k < nmid; ++k) {                                                        // :86-108, mid = mid_beg..=end
    const size_t i = nio + k;
    Fr v = poly_eval(vi + i * n * FR, n, s), w = poly_eval(wi + i * n * FR, n, s), y = poly_eval(yi + i * n * FR, n, s);
    stg1(c->vk_mid + k * G1W, mul_fr(g1_v, v)); stg1(c->g1_wk_mid + k * G1W, mul_fr(g1_w, w)); stg2(c->g2_wk_mid + k * G2W, mul_fr(g2_w, w));
    stg1(c->yk_mid + k * G1W, mul_fr(g1_y, y));
    stg1(c->alpha_vk_mid + k * G1W, mul_fr(mul_fr(g1_v, alpha_v), v)); stg1(c->alpha_wk_mid + k * G1W, mul_fr(mul_fr(g1_w, alpha_w), w));
    stg1(c->alpha_yk_mid + k * G1W, mul_fr(mul_fr(g1_y, alpha_y), y));
    stg1(c->beta_vwy_k_mid + k * G1W, affine_add(affine_add(mul_fr(mul_fr(g1_v, beta), v), mul_fr(mul_fr(g1_w, beta), w)), mul_fr(mul_fr(g1_y, beta), y)));
  }
  Fr sp(1);
  for (size_t k = 0; k < c->max_degree; ++k) { stg2(c->si + k * G2W, mul_fr(g2, sp)); sp = sp * s; }   // :98-99 (pow_seq)
  stg1(c->one_g1, mul_fr(g1, Fr(1))); stg2(c->one_g2, mul_fr(g2, Fr(1)));                    // :112-113
  stg2(c->alpha_v, mul_fr(g2, alpha_v)); stg1(c->alpha_w, mul_fr(g1, alpha_w)); stg2(c->alpha_y, mul_fr(g2, alpha_y));
  stg2(c->gamma, mul_fr(g2, gamma)); stg2(c->beta_gamma, mul_fr(mul_fr(g2, gamma), beta));
  G1Point t = mul_fr(g1_y, t_eval(n, s));                                                    // :120
  stg1(c->t, t);
  for (size_t i = 0; i < nio; ++i) {                                                         // :122-124
    stg1(c->vk_io + i * G1W, mul_fr(g1_v, poly_eval(vi + i * n * FR, n, s)));
    stg2(c->wk_io + i * G2W, mul_fr(g2_w, poly_eval(wi + i * n * FR, n, s)));
    stg1(c->yk_io + i * G1W, mul_fr(g1_y, poly_eval(yi + i * n * FR, n, s)));
  }
  stg1(c->alpha_v_t, mul_fr(t, alpha_v)); stg1(c->alpha_y_t, mul_fr(t, alpha_y)); stg1(c->beta_t, mul_fr(t, beta));   // :138-140
  return 0;
}

// Prover::prove (prover.rs:98-170); h = p / t coefficients (prover.rs:143-146), delta_v / delta_y injected
int zkto_pinocchio_prove(const zkto_pinocchio_crs* c, const uint64_t* wires, const uint64_t* hcoef, size_t h_len,
                         const uint64_t* delta_v_, const uint64_t* delta_y_, zkto_pinocchio_proof* pf) {
  init_fields();
  const size_t nio = c->n_io, nmid = c->n_mid;
  Fr dv = ldr(delta_v_), dy = ldr(delta_y_);
  G1Point t = ldg1(c->t), bt = ldg1(c->beta_t);
  G1Point v_mid = mul_fr(t, dv), g1_w_mid = G1Point::infinity(), y_mid = mul_fr(t, dy);      // :124-128
  G2Point g2_w_mid = G2Point::infinity();
  G1Point a_v = mul_fr(ldg1(c->alpha_v_t), dv), a_w = G1Point::infinity(), a_y = mul_fr(ldg1(c->alpha_y_t), dy);
  G1Point b_vwy = affine_add(mul_fr(bt, dv), mul_fr(bt, dy));                                // :131
  for (size_t k = 0; k < nmid; ++k) {                                                        // :133-146
    Fr w = ldr(wires + (nio + k) * FR);
    v_mid = affine_add(v_mid, mul_fr(ldg1(c->vk_mid + k * G1W), w));
    g1_w_mid = affine_add(g1_w_mid, mul_fr(ldg1(c->g1_wk_mid + k * G1W), w));
    g2_w_mid = affine_add(g2_w_mid, mul_fr(ldg2(c->g2_wk_mid + k * G2W), w));
    y_mid = affine_add(y_mid, mul_fr(ldg1(c->yk_mid + k * G1W), w));
    a_v = affine_add(a_v, mul_fr(ldg1(c->alpha_vk_mid + k * G1W), w));
    a_w = affine_add(a_w, mul_fr(ldg1(c->alpha_wk_mid + k * G1W), w));
    a_y = affine_add(a_y, mul_fr(ldg1(c->alpha_yk_mid + k * G1W), w));
    b_vwy = affine_add(b_vwy, mul_fr(ldg1(c->beta_vwy_k_mid + k * G1W), w));
  }
  std::vector<G2Point> si(h_len);
  for (size_t k = 0; k < h_len; ++k) si[k] = ldg2(c->si + k * G2W);
  G2Point h_s = msm_naive(si.data(), hcoef, FR, h_len);                                      // :153 eval_with_g2_hidings
  G2Point w_s = g2_w_mid;
  for (size_t i = 0; i < nio; ++i) w_s = affine_add(w_s, mul_fr(ldg2(c->wk_io + i * G2W), ldr(wires + i * FR)));   // :155-159
  G2Point adj = affine_add(affine_add(h_s, mul_fr(w_s, dv)), mul_fr(ldg2(c->one_g2), dy).neg());    // :160
  stg1(pf->v_mid_s, v_mid); stg1(pf->g1_w_mid_s, g1_w_mid); stg2(pf->g2_w_mid_s, g2_w_mid); stg1(pf->y_mid_s, y_mid); stg2(pf->h_s, adj);
  stg1(pf->alpha_v_mid_s, a_v); stg1(pf->alpha_w_mid_s, a_w); stg1(pf->alpha_y_mid_s, a_y); stg1(pf->beta_vwy_mid_s, b_vwy);
  return 0;
}

// Verifier::verify (verifier.rs:31-85): 1 accept, 0 reject, -2 if a tate() argument is the point at infinity (panic)
int zkto_pinocchio_verify(const zkto_pinocchio_crs* c, const zkto_pinocchio_proof* pf, const uint64_t* io_wires) {
  init_fields();
  static Pairing pr;
  auto e = [&](const G1Point& a, const G2Point& b) { return pr.tate(a, b); };
  try {
    G1Point v_mid = ldg1(pf->v_mid_s), g1_w = ldg1(pf->g1_w_mid_s), y_mid = ldg1(pf->y_mid_s);
    G2Point g2_w = ldg2(pf->g2_w_mid_s), one2 = ldg2(c->one_g2);
    {                                                                                        // :43-49
      G1Point vwy = affine_add(affine_add(v_mid, g1_w), y_mid);
      if (!(e(ldg1(pf->beta_vwy_mid_s), ldg2(c->gamma)) == e(vwy, ldg2(c->beta_gamma)))) return 0;
    }
    if (!(e(ldg1(pf->alpha_v_mid_s), one2) == e(v_mid, ldg2(c->alpha_v)))) return 0;         // :52-56
    if (!(e(ldg1(pf->alpha_w_mid_s), one2) == e(ldg1(c->alpha_w), g2_w))) return 0;          // :57-61
    if (!(e(ldg1(pf->alpha_y_mid_s), one2) == e(y_mid, ldg2(c->alpha_y)))) return 0;         // :62-66
    G1Point v_s = v_mid, y_s = y_mid; G2Point w_s = g2_w;                                    // :69-79
    for (size_t i = 0; i < c->n_io; ++i) {
      Fr w = ldr(io_wires + i * FR);
      v_s = affine_add(v_s, mul_fr(ldg1(c->vk_io + i * G1W), w));
      w_s = affine_add(w_s, mul_fr(ldg2(c->wk_io + i * G2W), w));
      y_s = affine_add(y_s, mul_fr(ldg1(c->yk_io + i * G1W), w));
    }
    Fq12 lhs = e(v_s, w_s), rhs = e(ldg1(c->t), ldg2(pf->h_s)) * e(y_s, one2);               // :81-84
    return lhs == rhs ? 1 : 0;
  } catch (const std::domain_error&) { return -2; }
}

// ---- Bulletproofs::inner_product_argument (bulletproofs.rs:19-55), challenges injected (xs[level]) -----------
typedef Fp<SpTag> Sp; typedef Fp<SnTag> Sn;
static SecpPoint ldsp(const uint64_t* p) { if (p[8] & 0xffffffffu) return SecpPoint::infinity(); return SecpPoint(Sp::from_limbs(p, 4), Sp::from_limbs(p + 4, 4)); }
static void stsp(uint64_t* p, const SecpPoint& v) {
  if (v.inf) { for (int i = 0; i < 8; ++i) p[i] = 0; p[8] = 1; return; }
  for (int i = 0; i < 4; ++i) { p[i] = v.x.l[i]; p[4 + i] = v.y.l[i]; } p[8] = 0;
}
static SecpPoint smul(const SecpPoint& p, const Sn& k) { return scalar_mul(p, k.l, 4); }
static SecpPoint vec_msm(const std::vector<SecpPoint>& g, size_t g0, const std::vector<Sn>& a, size_t a0, size_t cnt) {   // (gg * a).sum(), affine_points.rs:123-144,25-31
  SecpPoint s = SecpPoint::infinity();
  for (size_t i = 0; i < cnt; ++i) s = affine_add(s, smul(g[g0 + i], a[a0 + i]));
  return s;
}
// out_trace (optional): per level L,R (2 points) then the folded P' — lets the GPU path be compared level by level.
int zkto_bp_ipa(size_t n, const uint64_t* gg_, const uint64_t* hh_, const uint64_t* u_, const uint64_t* P_,
                const uint64_t* a_, const uint64_t* b_, const uint64_t* xs, uint64_t* out_trace) {
  init_fields();
  std::vector<SecpPoint> gg(n), hh(n); std::vector<Sn> a(n), b(n);
  for (size_t i = 0; i < n; ++i) { gg[i] = ldsp(gg_ + i * 9); hh[i] = ldsp(hh_ + i * 9); a[i] = Sn::from_limbs(a_ + i * 4, 4); b[i] = Sn::from_limbs(b_ + i * 4, 4); }
  SecpPoint u = ldsp(u_), P = ldsp(P_);
  size_t level = 0;
  while (n > 1) {
    size_t np = n / 2;
    Sn cL(0), cR(0);
    for (size_t i = 0; i < np; ++i) { cL = cL + a[i] * b[np + i]; cR = cR + a[np + i] * b[i]; }                                  // :36-37
    SecpPoint L = affine_add(affine_add(vec_msm(gg, np, a, 0, np), vec_msm(hh, 0, b, np, np)), smul(u, cL));                      // :39
    SecpPoint R = affine_add(affine_add(vec_msm(gg, 0, a, np, np), vec_msm(hh, np, b, 0, np)), smul(u, cR));                      // :40
    Sn x = Sn::from_limbs(xs + level * 4, 4), xi = x.inv();                                                                        // :42 (injected)
    std::vector<SecpPoint> g2(np), h2(np); std::vector<Sn> a2(np), b2(np);
    for (size_t i = 0; i < np; ++i) {
      g2[i] = affine_add(smul(gg[i], xi), smul(gg[np + i], x));                                                                    // :44
      h2[i] = affine_add(smul(hh[i], x), smul(hh[np + i], xi));                                                                    // :45
      a2[i] = a[i] * x + a[np + i] * xi; b2[i] = b[i] * xi + b[np + i] * x;                                                        // :49-50
    }
    P = affine_add(affine_add(smul(L, x.sq()), P), smul(R, x.sq().inv()));                                                         // :47
    if (out_trace) { stsp(out_trace + level * 27, L); stsp(out_trace + level * 27 + 9, R); stsp(out_trace + level * 27 + 18, P); }
    gg.swap(g2); hh.swap(h2); a.swap(a2); b.swap(b2); n = np; ++level;
  }
  Sn c = a[0] * b[0];                                                                                                              // :28-32
  SecpPoint rhs = affine_add(affine_add(smul(gg[0], a[0]), smul(hh[0], b[0])), smul(u, c));
  return P == rhs ? 1 : 0;
}
// helper for the tests: P = g^a h^b u^<a,b> (bulletproofs.rs:17)
int zkto_bp_commit(size_t n, const uint64_t* gg_, const uint64_t* hh_, const uint64_t* u_, const uint64_t* a_, const uint64_t* b_, uint64_t* P_) {
  init_fields();
  std::vector<SecpPoint> gg(n), hh(n); std::vector<Sn> a(n), b(n);
  Sn c(0);
  for (size_t i = 0; i < n; ++i) { gg[i] = ldsp(gg_ + i * 9); hh[i] = ldsp(hh_ + i * 9); a[i] = Sn::from_limbs(a_ + i * 4, 4); b[i] = Sn::from_limbs(b_ + i * 4, 4); c = c + a[i] * b[i]; }
  stsp(P_, affine_add(affine_add(vec_msm(gg, 0, a, 0, n), vec_msm(hh, 0, b, 0, n)), smul(ldsp(u_), c)));
  return 0;
}
// Fr / secp-n helpers the python harness uses to build QAPs and challenges (field only)
int zkto_sn_inv(const uint64_t* a, uint64_t* o) { init_fields(); Sn r; if (!Sn::from_limbs(a, 4).safe_inv(r)) return 1; for (int i = 0; i < 4; ++i) o[i] = r.l[i]; return 0; }

}  // extern "C"

// ---- Bulletproofs::range_proof (bulletproofs.rs:58-147), every random draw injected ------------------------
// rnd layout (4-limb residues mod the secp256k1 group order): alpha, rho, y, z, tau1, tau2, x, then sL[n], sR[n].
// u = the random point of :137 (used only with the inner-product argument), xs = the IPA challenges.
// out_pts (optional): A, S, T1, T2, P  (5 points) for parity with the GPU path.
extern "C" int zkto_bp_range_proof(size_t n, const uint64_t* V_, const uint64_t* aL_, const uint64_t* gamma_, const uint64_t* g_, const uint64_t* h_,
                                   const uint64_t* gg_, const uint64_t* hh_, int use_ipa, const uint64_t* rnd, const uint64_t* u_, const uint64_t* xs,
                                   uint64_t* out_pts) {
  init_fields();
  typedef std::vector<Sn> Vec;
  auto ldv = [&](const uint64_t* p, size_t cnt) { Vec v(cnt); for (size_t i = 0; i < cnt; ++i) v[i] = Sn::from_limbs(p + i * 4, 4); return v; };
  auto had = [](const Vec& a, const Vec& b) { Vec r(a.size()); for (size_t i = 0; i < a.size(); ++i) r[i] = a[i] * b[i]; return r; };
  auto add = [](const Vec& a, const Vec& b) { Vec r(a.size()); for (size_t i = 0; i < a.size(); ++i) r[i] = a[i] + b[i]; return r; };
  auto sub = [](const Vec& a, const Vec& b) { Vec r(a.size()); for (size_t i = 0; i < a.size(); ++i) r[i] = a[i] - b[i]; return r; };
  auto scl = [](const Vec& a, const Sn& s) { Vec r(a.size()); for (size_t i = 0; i < a.size(); ++i) r[i] = a[i] * s; return r; };
  auto sum = [](const Vec& a) { Sn s(0); for (const Sn& x : a) s = s + x; return s; };
  auto powseq = [&](const Sn& b) { Vec r(n); Sn x(1); for (size_t i = 0; i < n; ++i) { r[i] = x; x = x * b; } return r; };   // pow_seq, prime_field_elem.rs:346-361
  std::vector<SecpPoint> gg(n), hh(n);
  for (size_t i = 0; i < n; ++i) { gg[i] = ldsp(gg_ + i * 9); hh[i] = ldsp(hh_ + i * 9); }
  auto msm = [&](const std::vector<SecpPoint>& pts, const Vec& k) { return vec_msm(pts, 0, k, 0, n); };
  SecpPoint V = ldsp(V_), g = ldsp(g_), h = ldsp(h_);
  Vec aL = ldv(aL_, n);
  Sn gamma = Sn::from_limbs(gamma_, 4);
  Sn alpha = Sn::from_limbs(rnd, 4), rho = Sn::from_limbs(rnd + 4, 4), y = Sn::from_limbs(rnd + 8, 4), z = Sn::from_limbs(rnd + 12, 4),
     tau1 = Sn::from_limbs(rnd + 16, 4), tau2 = Sn::from_limbs(rnd + 20, 4), x = Sn::from_limbs(rnd + 24, 4);
  Vec sL = ldv(rnd + 28, n), sR = ldv(rnd + 28 + 4 * n, n);

  Vec one_n = powseq(Sn(1)), two_n = powseq(Sn(2));                                   // :72-73
  Vec aR = sub(aL, one_n);                                                             // :75
  SecpPoint A = affine_add(affine_add(smul(h, alpha), msm(gg, aL)), msm(hh, aR));      // :77
  SecpPoint S = affine_add(affine_add(smul(h, rho), msm(gg, sL)), msm(hh, sR));        // :82
  Vec y_n = powseq(y);                                                                 // :87
  Vec l0 = sub(aL, scl(one_n, z)), l1 = sL;                                            // :88-89
  Vec r0 = add(had(y_n, add(aR, scl(one_n, z))), scl(two_n, z.sq())), r1 = had(y_n, sR);   // :90-91
  Sn t0 = sum(had(l0, r0)), t1 = sum(had(l1, r0)) + sum(had(l0, r1)), t2 = sum(had(l1, r1));   // :93-95
  SecpPoint T1 = affine_add(smul(g, t1), smul(h, tau1)), T2 = affine_add(smul(g, t2), smul(h, tau2));   // :99-100
  Sn t_hat = t0 + t1 * x + t2 * x.sq();                                                // :104
  Sn tau_x = tau2 * x.sq() + tau1 * x + z.sq() * gamma;                                // :105
  Sn mu = alpha + rho * x;                                                             // :106
  Vec yinv_n = powseq(y.inv());
  std::vector<SecpPoint> hhp(n);
  for (size_t i = 0; i < n; ++i) hhp[i] = smul(hh[i], yinv_n[i]);                      // :109
  Sn delta_yz = (z - z.sq()) * sum(had(one_n, y_n)) - (z.sq() * z) * sum(had(one_n, two_n));   // :112
  SecpPoint lhs65 = affine_add(smul(g, t_hat), smul(h, tau_x));                        // :114
  SecpPoint rhs65 = affine_add(affine_add(affine_add(smul(V, z.sq()), smul(g, delta_yz)), smul(T1, x)), smul(T2, x.sq()));   // :115
  Vec l = add(sub(aL, scl(one_n, z)), scl(sL, x));                                     // :121
  Vec r = add(had(y_n, add(add(aR, scl(one_n, z)), scl(sR, x))), scl(two_n, z.sq()));  // :122
  SecpPoint P = affine_add(affine_add(affine_add(A, smul(S, x)), msm(gg, scl(one_n, z.negate()))),
                           msm(hhp, add(scl(y_n, z), scl(two_n, z.sq()))));            // :124-128
  if (out_pts) { stsp(out_pts, A); stsp(out_pts + 9, S); stsp(out_pts + 18, T1); stsp(out_pts + 27, T2); stsp(out_pts + 36, P); }
  if (!(lhs65 == rhs65)) return 0;                                                     // :116-118
  if (use_ipa) {
    SecpPoint u = ldsp(u_);
    SecpPoint Pp = affine_add(affine_add(P, smul(h, mu.negate())), smul(u, sum(had(l, r))));   // :138
    std::vector<uint64_t> ggb(n * 9), hhb(n * 9), lb(n * 4), rb(n * 4); uint64_t ub[9], pb[9];
    for (size_t i = 0; i < n; ++i) { stsp(ggb.data() + i * 9, gg[i]); stsp(hhb.data() + i * 9, hhp[i]); for (int j = 0; j < 4; ++j) { lb[i * 4 + j] = l[i].l[j]; rb[i * 4 + j] = r[i].l[j]; } }
    stsp(ub, u); stsp(pb, Pp);
    return zkto_bp_ipa(n, ggb.data(), hhb.data(), ub, pb, lb.data(), rb.data(), xs, nullptr);   // :139
  }
  SecpPoint rhs6667 = affine_add(affine_add(smul(h, mu), msm(gg, l)), msm(hhp, r));    // :142
  if (!(P == rhs6667)) return 0;
  return t_hat == sum(had(l, r)) ? 1 : 0;                                              // :147-149
}
