// TEST-ONLY: the kernel math headers (fp.h … pairing.h) compiled for the HOST, so the
// `-m "not gpu"` suite can check the exact functions the HIP kernels inline against the
// oracle without a GPU.  Not part of the product: libzkt_hip.so neither links nor calls
// this, and the C ABI in include/zkt.h has no CPU path.  Built by __graft_entry__.build()
// as zk-toolkit_amd/libzkt_hostcheck.so (hipcc --cuda-host-only).
#include "abi.h"
#include "fq_program.h"
using namespace zkt;

extern "C" {
unsigned long zkt_hostcheck_bgcd_fallbacks() { return bgcd_fallbacks(); }      // 0 expected: the word-step GCD always ends with b = 1
int zkt_hostcheck_fq_program(uint64_t seed, int steps, const uint32_t* in4, uint32_t* out4) { return fq_program(seed, steps, in4, out4); }
// op: 0 add 1 sub 2 mul 3 sqr 4 neg 5 inv (fp_inv: the word-step binary GCD) 6 inv as x^(p-2) through fp_pow 7 cube 8 pow (b = exponent, ABI_N words per element)
//     9 the classic bit-step binary Euclid on canonical words (bgcd_inverse_classic), 10 the word-step GCD on canonical words (bgcd_inverse)
int zkt_hostcheck_fp(int field, int op, const uint32_t* a, const uint32_t* b, uint32_t* o, size_t n) {
  auto run = [&](auto tag) {
    typedef decltype(tag) C;
    for (size_t i = 0; i < n; ++i) {
      if (op == 9 || op == 10) { uint32_t w[C::ABI_N]; for (int j = 0; j < C::ABI_N; ++j) w[j] = a[i * C::ABI_N + j]; if (op == 9) bgcd_inverse_classic<C>(w); else bgcd_inverse<C>(w); for (int j = 0; j < C::ABI_N; ++j) o[i * C::ABI_N + j] = w[j]; continue; }
      Fp<C> x = ld_fp<C>(a + i * C::ABI_N), y = b ? ld_fp<C>(b + i * C::ABI_N) : fp_zero<C>(), r;
      switch (op) {
        case 0: r = fp_add(x, y); break; case 1: r = fp_sub(x, y); break; case 2: r = fp_mul(x, y); break;
        case 3: r = fp_sqr(x); break; case 4: r = fp_neg(x); break; case 6: { uint32_t e[C::ABI_N]; for (int j = 0; j < C::ABI_N; ++j) e[j] = C::pm2(j); r = fp_pow(x, e, C::ABI_N); } break;
        case 7: r = fp_mul(fp_sqr(x), x); break; case 8: r = fp_pow(x, b + i * C::ABI_N, C::ABI_N); break; default: r = fp_inv(x);
      }
      st_fp<C>(o + i * C::ABI_N, r);
    }
  };
  switch (field) { case 0: run(FqC{}); break; case 1: run(FrC{}); break; case 2: run(SpC{}); break; default: run(SnC{}); }
  return 0;
}
// Both Fq2 products (four-scan lazy form and three-scan Karatsuba form, tower.h / fp.h) on non-canonical representatives: every operand
// coordinate is moved to canonical + j*p (j < 3, selected by two bits of `lifts` each).  Writes the canonical product; returns the number
// of disagreements between the two forms or with the lazy-limb invariants (0 expected).
int zkt_hostcheck_fq2_mul_lifted(const uint32_t* a, const uint32_t* b, unsigned lifts, uint32_t* o) {
  Fq2 x = ld_fq2(a), y = ld_fq2(b);
  x.c0 = fq_lift(x.c0, (lifts & 3) % 3); x.c1 = fq_lift(x.c1, ((lifts >> 2) & 3) % 3);
  y.c0 = fq_lift(y.c0, ((lifts >> 4) & 3) % 3); y.c1 = fq_lift(y.c1, ((lifts >> 6) & 3) % 3);
  const Fq2 four{fp_mulsub(x.c0, y.c0, x.c1, y.c1), fp_muladd(x.c0, y.c1, x.c1, y.c0)};
  Fq2 three; fp2_mul_kara(x.c0, x.c1, y.c0, y.c1, three.c0, three.c1);
  int bad = 0;
  if (!fq_lazy_ok(three.c0) || !fq_lazy_ok(three.c1)) ++bad;
  if (!fp_eq(four.c0, three.c0) || !fp_eq(four.c1, three.c1)) ++bad;
  st_fq2(o, three);
  return bad;
}
// op: 0 add 1 sub 2 mul 3 inv 4 neg 5 mul_xi/mul_v 6 sqr 7 frob1 8 frob2 9 conj
int zkt_hostcheck_tower(int deg, int op, const uint32_t* a, const uint32_t* b, uint32_t* o) {
  if (deg == 2) {
    Fq2 x = ld_fq2(a), y = b ? ld_fq2(b) : fq2_zero(), r;
    switch (op) { case 0: r = fq2_add(x, y); break; case 1: r = fq2_sub(x, y); break; case 2: r = fq2_mul(x, y); break;
      case 3: r = fq2_inv(x); break; case 4: r = fq2_neg(x); break; case 5: r = fq2_mul_xi(x); break; default: r = fq2_sqr(x); }
    st_fq2(o, r);
  } else if (deg == 6) {
    Fq6 x = ld_fq6(a), y = b ? ld_fq6(b) : fq6_zero(), r;
    switch (op) { case 0: r = fq6_add(x, y); break; case 1: r = fq6_sub(x, y); break; case 2: r = fq6_mul(x, y); break;
      case 3: r = fq6_inv(x); break; case 4: r = fq6_neg(x); break; default: r = fq6_mul_v(x); }
    st_fq6(o, r);
  } else {
    Fq12 x = ld_fq12(a), y = b ? ld_fq12(b) : fq12_one(), r;
    switch (op) { case 0: r = fq12_add(x, y); break; case 1: r = fq12_sub(x, y); break; case 2: r = fq12_mul(x, y); break;
      case 3: r = fq12_inv(x); break; case 4: r = fq12_neg(x); break; case 6: r = fq12_sqr(x); break;
      case 7: r = fq12_frob<1>(x); break; case 8: r = fq12_frob<2>(x); break; case 10: r = fq12_cyclotomic_sqr(x); break; default: r = fq12_conj(x); }
    st_fq12(o, r);
  }
  return 0;
}
}  // extern "C"
template <class F> static void pt_add(const uint32_t* a, const uint32_t* b, uint32_t* o) {
  Aff<F> p = PtIO<F>::ld(a), q = PtIO<F>::ld(b);
  PtIO<F>::st(o, jac_to_aff(jac_add_aff(jac_from_aff(p), q)));
}
template <class F> static void pt_add_full(const uint32_t* a, const uint32_t* b, uint32_t* o) {
  // exercise jac_add and the XYZZ formulas too: ((2a) + b) - a computed two ways must agree with a + b
  Aff<F> p = PtIO<F>::ld(a), q = PtIO<F>::ld(b);
  Jac<F> j = jac_add(jac_dbl(jac_from_aff(p)), jac_from_aff(q));
  Aff<F> np = p; if (!np.inf) np.y = F::neg(np.y);
  j = jac_add_aff(j, np);
  Xyzz<F> z = xyzz_inf<F>();
  if (!p.inf) z = xyzz_add_aff(z, p.x, p.y);
  if (!q.inf) z = xyzz_add_aff(z, q.x, q.y);
  Xyzz<F> z2 = xyzz_add(z, xyzz_inf<F>());
  Aff<F> r1 = jac_to_aff(j), r2 = xyzz_to_aff(z2), r3 = jac_to_aff(xyzz_to_jac(z));
  bool same = (r1.inf == r2.inf) && (r1.inf || (F::eq(r1.x, r2.x) && F::eq(r1.y, r2.y)));
  same = same && (r1.inf == r3.inf) && (r1.inf || (F::eq(r1.x, r3.x) && F::eq(r1.y, r3.y)));
  if (!same) { r1.inf = true; }   // poison: a mismatch between formula sets shows up as a wrong answer
  PtIO<F>::st(o, r1);
}
template <class F> static void pt_mul(const uint32_t* a, const uint32_t* k, int klimbs, uint32_t* o) {
  PtIO<F>::st(o, jac_to_aff(scalar_mul_aff(PtIO<F>::ld(a), k, klimbs)));
}
extern "C" {
// grp: 0 G1, 1 G2, 2 secp256k1.  op: 0 add (mixed), 1 add via the other formula sets, 2 scalar mul (b = scalar, u32 limbs)
int zkt_hostcheck_group(int grp, int op, const uint32_t* a, const uint32_t* b, int klimbs, uint32_t* o) {
  if (grp == 0) { if (op == 0) pt_add<FqOps>(a, b, o); else if (op == 1) pt_add_full<FqOps>(a, b, o); else pt_mul<FqOps>(a, b, klimbs, o); }
  else if (grp == 1) { if (op == 0) pt_add<Fq2Ops>(a, b, o); else if (op == 1) pt_add_full<Fq2Ops>(a, b, o); else pt_mul<Fq2Ops>(a, b, klimbs, o); }
  else { if (op == 0) pt_add<SpOps>(a, b, o); else if (op == 1) pt_add_full<SpOps>(a, b, o); else pt_mul<SpOps>(a, b, klimbs, o); }
  return 0;
}
// which: 0 calc_g1_g2, 1 calc_g2_g1, 2 weil
int zkt_hostcheck_miller_exact(int which, const uint32_t* g1, const uint32_t* g2, uint32_t* o) {
  Aff<FqOps> p = PtIO<FqOps>::ld(g1); Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(g2);
  if (p.inf || q.inf) return 2;
  Fq12 r; bool bad = false;
  if (which == 0) r = miller_g1_g2_exact(p.x, p.y, q.x, q.y, bad);
  else if (which == 1) r = miller_g2_g1_exact(q.x, q.y, p.x, p.y);
  else r = fq12_mul(miller_g1_g2_exact(p.x, p.y, q.x, q.y, bad), fq12_inv(miller_g2_g1_exact(q.x, q.y, p.x, p.y)));
  if (bad) return 2;
  st_fq12(o, r);
  return 0;
}
int zkt_hostcheck_tate(const uint32_t* g1, const uint32_t* g2, uint32_t* o) {
  Aff<FqOps> p = PtIO<FqOps>::ld(g1); Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(g2);
  if (p.inf || q.inf) return 2;
  // the three passes of launch_tate (zkt_tate.hip): 127-step loop for P in G1 and Q in G2, 255-step loop for Q on E' outside G2, the reference's own chain otherwise
  Fq12 f; bool in_g1 = false, bad;
  const int route = tate_short(p.x, p.y, q.x, q.y, f);
  if (route == TATE_ROUTE_SHORT) { st_fq12(o, f); return 0; }
  if (route == TATE_ROUTE_LONG) {
    f = miller_g1_g2(p.x, p.y, q.x, q.y, in_g1);
    if (in_g1) { st_fq12(o, final_exponentiation(f)); return 50; }          // 50: value produced by the 255-step loop
  }
  f = miller_g1_g2_exact(p.x, p.y, q.x, q.y, bad);
  if (bad) return 2;
  if (fq12_is_zero(f)) { for (int k = 0; k < 144; ++k) o[k] = 0; return 100; }
  st_fq12(o, final_exponentiation(f));
  return 100;                             // 100: value produced by the exact path
}
// the membership tests that guard the 127-step loop: bit 0 = P on E, bit 1 = r P == infinity (from the loop), bit 2 = Q in G2, bit 3 = Q on E'
int zkt_hostcheck_short_loop_guards(const uint32_t* g1, const uint32_t* g2) {
  Aff<FqOps> p = PtIO<FqOps>::ld(g1); Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(g2);
  if (p.inf || q.inf) return -1;
  bool ok; (void)miller_g1_g2_short(p.x, p.y, q.x, q.y, ok);
  return (g1_on_curve(p.x, p.y) ? 1 : 0) | (ok ? 2 : 0) | (g2_on_curve(q.x, q.y) && g2_in_subgroup(q.x, q.y) ? 4 : 0) | (g2_on_curve(q.x, q.y) ? 8 : 0);
}
// The 63-step (optimal-ate) product of the deciding entry points (pairing.h): kv pairs whose Q runs its own chain, then kf pairs whose Q comes as a line table built here.
// Returns -1 when a guard refuses an argument (the kernels then leave the element to the older routes), else writes final_exponentiation(prod f_{|x|,Q_k}(P_k)).
// bit 8 of the return value: the table builder's G2 verdict for the tabulated points disagreed with g2_in_subgroup (0 expected).
int zkt_hostcheck_ate_product(int kv, int kf, const uint32_t* g1, const uint32_t* g2, uint32_t* o) {
  const int K = kv + kf;
  if (K < 1 || K > 3) return -2;
  Fq xp[3], yp[3]; Fq2 xq[3], yq[3];
  static uint32_t tabs[3][ATE_LINES * ATE_LINE_WORDS];
  const uint32_t* tp[3] = {tabs[0], tabs[1], tabs[2]};
  int odd = 0;
  for (int k = 0; k < K; ++k) {
    Aff<FqOps> p = PtIO<FqOps>::ld(g1 + k * ABI_G1_WORDS); Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(g2 + k * ABI_G2_WORDS);
    if (p.inf || q.inf) return -3;
    if (!g1_on_curve(p.x, p.y) || !g1_in_subgroup(p.x, p.y) || !g2_on_curve(q.x, q.y)) return -1;
    xp[k] = p.x; yp[k] = p.y; xq[k] = q.x; yq[k] = q.y;
    if (k >= kv) {
      const bool in = ate_line_table(q.x, q.y, tabs[k - kv]);
      if (in != g2_in_subgroup(q.x, q.y)) odd = 256;
      if (!in) return -1 - odd;
    }
  }
  bool q_ok = false; Fq12 f;
  if (kv == 1 && kf == 0) f = miller_ate_multi<1, 0>(xp, yp, xq, yq, tp, q_ok);
  else if (kv == 2 && kf == 0) f = miller_ate_multi<2, 0>(xp, yp, xq, yq, tp, q_ok);
  else if (kv == 3 && kf == 0) f = miller_ate_multi<3, 0>(xp, yp, xq, yq, tp, q_ok);
  else if (kv == 1 && kf == 2) f = miller_ate_multi<1, 2>(xp, yp, xq, yq, tp, q_ok);
  else if (kv == 0 && kf == 1) f = miller_ate_multi<0, 1>(xp, yp, xq, yq, tp, q_ok);
  else return -2;
  if (!q_ok) return -1;
  st_fq12(o, final_exponentiation_3h(f));       // the exponent the deciding kernels use (three times the exact one)
  return odd;
}
}
