// Batched Tate pairing for BLS12-381: one pairing per lane.
//
// Replaces, value for value,
//   Pairing::tate     src/building_block/curves/bls12_381/pairing.rs:86-100
// = calc_g1_g2(P,Q)^((q^12-1)/r)  (textbook Miller loop over the bits of r-1 with
// explicit untwist, vertical lines and one Fq12 inversion per step, pairing.rs:20-73;
// then a 4314-bit square-and-multiply, fq12.rs:42-57).
//
// What is computed here instead (proved bit-identical on the CPU by
// tests/test_fast_model.py against the faithful oracle, and on the GPU by
// tests/test_gpu_parity.py):
//  * Miller loop over the same bits of r-1, with V in Jacobian coordinates (no
//    inversion), tangent/chord lines scaled by an Fq factor, vertical lines
//    dropped.  Every dropped or extra factor lies in Fq6 and is annihilated by
//    the (q^6-1) part of the final exponent, so tate() is unchanged.
//  * the untwisted Q has x' = (x/xi) v^2 (slot w0.v2) and y' = (y/xi) v w (slot
//    w1.v1)  (g12_point.rs:47-68), so a line value is a + b v^2 + c v w with
//    a in Fq: fq12_mul_line.
//  * final exponent split exactly: (q^12-1)/r = (q^6-1)(q^2+1) * [e1 (x+q)(x^2+q^2-1) + 1],
//    e1 = (x-1)^2/3, x = -0xd201000000010000 — the exact exponent, not the usual
//    3x multiple.
#pragma once
#include "curve.h"

namespace zkt {

ZKT_HD Fq2 xi_inv_const() { return fq2_const([](int i) { return xi_inv_limb(0, i); }, [](int i) { return xi_inv_limb(1, i); }); }

// One Miller step on the G1 side as its own function: its Fq temporaries live in a transient frame instead of the
// Miller loop's, which must stay small (<4 KB per lane keeps the whole grid's Fq12 working set inside the 256 MB MALL).
struct MillerLine { Fq a; Fq2 b, c; };     // a + b v^2 + c v w
struct MillerPt { Fq X, Y, Z; };
// tangent at V scaled by 2YZ^3: (3X^3 - 2Y^2) - 3X^2 Z^2 * X' + Z3 Z^2 * Y';  V <- 2V
ZKT_FN void miller_dbl_step(MillerPt& V, const Fq2& Xq, const Fq2& Yq, MillerLine& l) {
  const Fq X = V.X, Y = V.Y, Z = V.Z;
  Fq A = fp_sqr(X), B = fp_sqr(Y), C = fp_sqr(B), ZZ = fp_sqr(Z);
  Fq t = fp_sqr(fp_add(X, B));
  Fq D = fp_dbl(fp_sub(fp_sub(t, A), C));
  Fq E = fp_add(fp_dbl(A), A);
  Fq X3 = fp_sub2(fp_sqr(E), fp_zero<FqC>(), D);
  Fq Y3 = fp_sub(fp_mul(E, fp_sub(D, X3)), fp_dbl(fp_dbl(fp_dbl(C))));
  Fq Z3 = fp_dbl(fp_mul(Y, Z));
  l.a = fp_sub(fp_mul(E, X), fp_dbl(B));
  l.b = fq2_mul_fq(Xq, fp_neg(fp_mul(E, ZZ)));
  l.c = fq2_mul_fq(Yq, fp_mul(Z3, ZZ));
  V.X = X3; V.Y = Y3; V.Z = Z3;
}
// chord through V and P scaled by Z*H: (R xp - Z3 yp) - R X' + Z3 Y';  V <- V + P
ZKT_FN void miller_add_step(MillerPt& V, const Fq& xp, const Fq& yp, const Fq2& Xq, const Fq2& Yq, MillerLine& l) {
  const Fq X = V.X, Y = V.Y, Z = V.Z;
  Fq ZZ = fp_sqr(Z), H = fp_sub(fp_mul(xp, ZZ), X), Rr = fp_sub(fp_mul(fp_mul(yp, ZZ), Z), Y);
  Fq HH = fp_sqr(H), HHH = fp_mul(H, HH), Vv = fp_mul(X, HH);
  Fq X3 = fp_sub2(fp_sqr(Rr), HHH, Vv);
  Fq Y3 = fp_mulsub(Rr, fp_sub(Vv, X3), Y, HHH);
  Fq Z3 = fp_mul(Z, H);
  l.a = fp_mulsub(Rr, xp, Z3, yp);
  l.b = fq2_mul_fq(Xq, fp_neg(Rr));
  l.c = fq2_mul_fq(Yq, Z3);
  V.X = X3; V.Y = Y3; V.Z = Z3;
}

// V == -P for a Jacobian V and an affine P?  After the signed-digit chain V = (r-1) P, so this is r P == infinity: P lies in the
// order-r subgroup G1 — four multiplications, once per pairing.
ZKT_FN bool miller_pt_is_neg(const MillerPt& V, const Fq& xp, const Fq& yp) {
  if (fp_is_zero(V.Z)) return false;
  const Fq ZZ = fp_sqr(V.Z);
  return fp_eq(fp_mul(xp, ZZ), V.X) && fp_eq(fp_mul(fp_mul(yp, ZZ), V.Z), fp_neg(V.Y));
}

// f_{r-1,P}(untwist(Q)) up to Fq6 factors.  P, Q affine, Montgomery domain, neither at infinity.
// in_g1 <- r P == infinity.  For such P no multiple met on the way is infinity and the value equals the reference's for EVERY Q ON THE TWIST
// (the untwisted X of any Fq2 abscissa lies in Fq6, so the dropped vertical lines die in the final exponentiation; but the signed-digit chain
// and the reference's binary chain are the same function on the curve only — a Q off E' must take miller_g1_g2_exact, see g2_on_curve).  For P outside
// G1 the reference's behaviour depends on the order of P (it panics when a binary prefix multiple of P is infinity,
// rational_function.rs:36); the callers then take the exact path (miller_g1_g2_exact) or fail closed.
ZKT_FN Fq12 miller_g1_g2(const Fq& xp, const Fq& yp, const Fq2& xq, const Fq2& yq, bool& in_g1) {
  const Fq2 xi_inv = xi_inv_const();
  const Fq2 Xq = fq2_mul(xq, xi_inv), Yq = fq2_mul(yq, xi_inv);
  MillerPt V{xp, yp, fp_one<FqC>()};
  MillerLine l;
  Fq12 f = fq12_one(), ft;          // ping-pong f <-> ft: a destination never aliases a source
  const Fq yn = fp_neg(yp);
  // signed digits of r-1 (NAF: 58 additions/subtractions instead of the 132 additions of the binary chain; a digit -1 adds -P).
  // The chains differ by vertical lines only, which the final exponentiation kills (oracle/fast_model.py, tests/test_fast_model.py).
  for (int i = 0; i < MILLER_NAF_DIGITS; ++i) {
    uint32_t nz = 0, ng = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { nz = (j == (i >> 5)) ? miller_naf_nz_word(j) : nz; ng = (j == (i >> 5)) ? miller_naf_neg_word(j) : ng; }
    const bool bit = (nz >> (i & 31)) & 1, neg = (ng >> (i & 31)) & 1;          // wave-uniform (compile-time tables)
    miller_dbl_step(V, Xq, Yq, l);
    fq12_sqr_mul_line(f, l.a, l.b, l.c);
    if (bit) {
      miller_add_step(V, xp, neg ? yn : yp, Xq, Yq, l);
      fq12_mul_line_ip(f, l.a, l.b, l.c);
    }
  }
  in_g1 = miller_pt_is_neg(V, xp, yp);
  return f;
}

// ---- the 127-step loop (twisted-ate form) -------------------------------------------------------------------------------------
// For P in G1 and Q in G2 the value eta = f_{x^2,P}(Q)^((q^12-1)/r) is a fixed power of the Tate value: with s = x^2, r = s^2 - s + 1, so
// s^3 = -1 (mod r) and s^6 - 1 = L r with L = -2(s+1); f_{s^6,P} = prod_i f_{s,P}^(s^(5-i) q^(2i)) and q^2 = s on G_T give tate^L = eta^(6 s^5), i.e.
//   tate = eta^(2 x^2 - 1) = pi^2(eta)^2 * conj(eta)          (q = x (mod r): on G_T the Frobenius IS the power by x)
// — half the Miller loop for one cyclotomic squaring and one product.  oracle/fast_model.py (tate_short) is the python model;
// tests/test_fast_model.py checks it against the faithful oracle and checks both membership tests below against r P = infinity, r Q = infinity.
// The identity needs P in G1 and Q in G2, so both are TESTED here; Q on the twist but outside G2 takes the 255-step loop (valid for every Q ON E'),
// P outside G1 or a point off its curve takes the reference's own chain — the callers mark such elements and re-run them.
ZKT_HD Fq g1_beta_const() { Fq b;
#pragma unroll
  for (int i = 0; i < FqC::N; ++i) b.v[i] = g1_beta_limb(i);
  return b; }
ZKT_HD Fq fq_four() { return fp_dbl(fp_dbl(fp_one<FqC>())); }
// P on E: y^2 = x^3 + 4 (the reference never checks; off-curve points keep their old route)
ZKT_FN bool g1_on_curve(const Fq& x, const Fq& y) { return fp_eq(fp_sqr(y), fp_add(fp_mul(fp_sqr(x), x), fq_four())); }
// Q on E': y^2 = x^3 + 4(1+u).  Off the twist even the 255-step loop is not the reference's value: two addition chains for the same multiple of P
// give functions that agree on the curve only, and the untwisted Q is then not on it — such elements take the reference's own chain.
ZKT_FN bool g2_on_curve(const Fq2& xq, const Fq2& yq) {
  const Fq four = fq_four();
  const Fq2 b{four, four};
  return fq2_eq(fq2_sqr(yq), fq2_add(fq2_mul(fq2_sqr(xq), xq), b));
}
// Q in G2  <=>  Q on E' (the caller's check) and psi(Q) = [x] Q, psi = twist o Frobenius o untwist (x < 0: psi(Q) = -[|x|] Q).
// psi^2 - t psi + q = 0 on E', t = x + 1 and q = x (mod r); r is prime to the cofactor of E'(Fq2), so the r-torsion of E'(Fq2) is G2.
ZKT_HD bool g2_in_subgroup_inl(const Fq2& xq, const Fq2& yq) {
  const Aff<Fq2Ops> q{xq, yq, false};
  Jac<Fq2Ops> a = jac_from_aff(q);
  for (int i = 62; i >= 0; --i) {                              // |x| = 0xd201000000010000: 63 doublings, 5 additions
    a = jac_dbl(a);
    if ((BLS_X_ABS >> i) & 1) a = jac_add_aff(a, q);
  }
  if (jac_is_inf(a)) return false;
  const Fq2 px = fq2_mul(fq2_const([](int i) { return g2_psi_x_limb(0, i); }, [](int i) { return g2_psi_x_limb(1, i); }), fq2_conj(xq));
  const Fq2 py = fq2_mul(fq2_const([](int i) { return g2_psi_y_limb(0, i); }, [](int i) { return g2_psi_y_limb(1, i); }), fq2_conj(yq));
  const Fq2 ZZ = fq2_sqr(a.Z);
  return fq2_eq(a.X, fq2_mul(px, ZZ)) && fq2_eq(a.Y, fq2_neg(fq2_mul(py, fq2_mul(ZZ, a.Z))));
}
ZKT_FN bool g2_in_subgroup(const Fq2& xq, const Fq2& yq) { return g2_in_subgroup_inl(xq, yq); }
// V == (BETA xp, -yp) for a Jacobian V?  After the loop over x^2, V = x^2 P; -x^2 is the eigenvalue of phi(x, y) = (BETA x, y) on G1, so this is
// phi(P) = [-x^2] P, and phi^2 + phi + 1 = 0 turns it into [x^4 - x^2 + 1] P = r P = infinity (and conversely) — six multiplications, once per pairing.
ZKT_FN bool miller_pt_is_x2(const MillerPt& V, const Fq& xp, const Fq& yp) {
  if (fp_is_zero(V.Z)) return false;
  const Fq ZZ = fp_sqr(V.Z);
  return fp_eq(fp_mul(fp_mul(g1_beta_const(), xp), ZZ), V.X) && fp_eq(fp_mul(fp_mul(yp, ZZ), V.Z), fp_neg(V.Y));
}
// f_{x^2,P}(untwist(Q)) up to Fq6 factors; ok <- P in G1 (P on the curve is the caller's check)
ZKT_FN Fq12 miller_g1_g2_short(const Fq& xp, const Fq& yp, const Fq2& xq, const Fq2& yq, bool& ok) {
  const Fq2 xi_inv = xi_inv_const();
  const Fq2 Xq = fq2_mul(xq, xi_inv), Yq = fq2_mul(yq, xi_inv);
  MillerPt V{xp, yp, fp_one<FqC>()};
  MillerLine l;
  Fq12 f = fq12_one(), ft;
  for (int i = 0; i < MILLER_X2_NBITS; ++i) {
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) w = (j == (i >> 5)) ? miller_x2_bits_word(j) : w;
    const bool bit = (w >> (i & 31)) & 1;                                       // wave-uniform (compile-time table)
    miller_dbl_step(V, Xq, Yq, l);
    fq12_sqr_mul_line(f, l.a, l.b, l.c);
    if (bit) {
      miller_add_step(V, xp, yp, Xq, Yq, l);
      fq12_mul_line_ip(f, l.a, l.b, l.c);
    }
  }
  ok = miller_pt_is_x2(V, xp, yp);
  return f;
}

// prod_k f_{r-1,P_k}(untwist(Q_k)) for K pairs sharing ONE squaring chain (row f-2: the Groth16 verifier's three pairings,
// verifier.rs:30-54, as a single multi-Miller loop + a single final exponentiation).
template <int K>
ZKT_FN Fq12 miller_g1_g2_multi(const Fq* xp, const Fq* yp, const Fq2* xq, const Fq2* yq, bool& all_in_g1) {
  const Fq2 xi_inv = xi_inv_const();
  Fq2 Xq[K], Yq[K]; MillerPt V[K];
  for (int k = 0; k < K; ++k) { Xq[k] = fq2_mul(xq[k], xi_inv); Yq[k] = fq2_mul(yq[k], xi_inv); V[k] = MillerPt{xp[k], yp[k], fp_one<FqC>()}; }
  MillerLine l;
  Fq12 f = fq12_one(), ft;
  Fq yn[K];
  for (int k = 0; k < K; ++k) yn[k] = fp_neg(yp[k]);
  for (int i = 0; i < MILLER_NAF_DIGITS; ++i) {                                  // signed digits of r-1, as in miller_g1_g2
    uint32_t nz = 0, ng = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { nz = (j == (i >> 5)) ? miller_naf_nz_word(j) : nz; ng = (j == (i >> 5)) ? miller_naf_neg_word(j) : ng; }
    const bool bit = (nz >> (i & 31)) & 1, neg = (ng >> (i & 31)) & 1;
    for (int k = 0; k < K; ++k) {
      miller_dbl_step(V[k], Xq[k], Yq[k], l);
      if (k == 0) fq12_sqr_mul_line(f, l.a, l.b, l.c);          // the shared squaring, fused with the first pair's line (f stays out of scratch between them)
      else fq12_mul_line_ip(f, l.a, l.b, l.c);
    }
    if (bit) {
      for (int k = 0; k < K; ++k) {
        miller_add_step(V[k], xp[k], neg ? yn[k] : yp[k], Xq[k], Yq[k], l);
        fq12_mul_line_ip(f, l.a, l.b, l.c);
      }
    }
  }
  all_in_g1 = true;
  for (int k = 0; k < K; ++k) all_in_g1 = all_in_g1 && miller_pt_is_neg(V[k], xp[k], yp[k]);
  return f;
}

// The same product over the 127-step loop: prod_k f_{x^2,P_k}(untwist(Q_k)).  After the final exponentiation this is (prod_k tate_k)^(1/(2x^2-1)):
// equal to one exactly when the Tate product is (2x^2-1 is a unit mod r), and final_exponentiation_t<true> turns it into the Tate product itself.
// Needs every Q_k in G2 and every P_k on E (the callers test both); all_in_g1 <- every P_k in G1, from the points the loops end on.
template <int K>
ZKT_FN Fq12 miller_g1_g2_multi_short(const Fq* xp, const Fq* yp, const Fq2* xq, const Fq2* yq, bool& all_in_g1) {
  const Fq2 xi_inv = xi_inv_const();
  Fq2 Xq[K], Yq[K]; MillerPt V[K];
  for (int k = 0; k < K; ++k) { Xq[k] = fq2_mul(xq[k], xi_inv); Yq[k] = fq2_mul(yq[k], xi_inv); V[k] = MillerPt{xp[k], yp[k], fp_one<FqC>()}; }
  MillerLine l;
  Fq12 f = fq12_one(), ft;
  for (int i = 0; i < MILLER_X2_NBITS; ++i) {
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) w = (j == (i >> 5)) ? miller_x2_bits_word(j) : w;
    const bool bit = (w >> (i & 31)) & 1;
    for (int k = 0; k < K; ++k) {
      miller_dbl_step(V[k], Xq[k], Yq[k], l);
      if (k == 0) fq12_sqr_mul_line(f, l.a, l.b, l.c);          // the shared squaring, fused with the first pair's line (f stays out of scratch between them)
      else fq12_mul_line_ip(f, l.a, l.b, l.c);
    }
    if (bit) {
      for (int k = 0; k < K; ++k) {
        miller_add_step(V[k], xp[k], yp[k], Xq[k], Yq[k], l);
        fq12_mul_line_ip(f, l.a, l.b, l.c);
      }
    }
  }
  all_in_g1 = true;
  for (int k = 0; k < K; ++k) all_in_g1 = all_in_g1 && miller_pt_is_x2(V[k], xp[k], yp[k]);
  return f;
}
// every P_k on E, every Q_k on E' and in G2: what the 127-step product needs before it starts.  Bit k of q_known_good: Q_k was tested by the caller already
// (a point shared by the whole batch — the verifying key's gamma, delta — is tested once per launch, not once per lane).
template <int K>
ZKT_FN bool pairing_args_fit_short_loop(const Fq* xp, const Fq* yp, const Fq2* xq, const Fq2* yq, unsigned q_known_good = 0) {
  for (int k = 0; k < K; ++k) {
    if (!g1_on_curve(xp[k], yp[k])) return false;
    if (!((q_known_good >> k) & 1) && (!g2_on_curve(xq[k], yq[k]) || !g2_in_subgroup(xq[k], yq[k]))) return false;
  }
  return true;
}

// a^|x| , |x| = 0xd201000000010000 (bits 63,62,60,57,48,16)
// (a in the cyclotomic subgroup: Granger-Scott squarings)
ZKT_FN Fq12 fq12_pow_xabs(const Fq12& a) {
  ZKT_FORCE_FRAME();
  Fq12 r = a;
  int run = 0;                                                  // squarings owed: done as ONE in-register run before the next product (fq12_cyclotomic_sqr_n)
  for (int i = 62; i >= 0; --i) {
    ++run;
    if ((BLS_X_ABS >> i) & 1) { fq12_cyclotomic_sqr_n_mul(r, run, &a); run = 0; }
  }
  if (run) fq12_cyclotomic_sqr_n(r, run);
  return r;
}
// a^x for x = -|x|, a in the cyclotomic subgroup (inverse = conjugate)
ZKT_HD Fq12 fq12_pow_x(const Fq12& a) { return fq12_conj(fq12_pow_xabs(a)); }

// a^e1, e1 = (x-1)^2/3, a in the cyclotomic subgroup: width-3 signed digits {0, +-1, +-3} (a^-1 = conj(a)), 126 Granger-Scott squarings
// and 30 products (+ a^3) instead of the 47 products of the binary chain
ZKT_FN Fq12 fq12_pow_e1(const Fq12& a) {
  Fq12 a3, r, t, ac, a3c;
  t = fq12_cyclotomic_sqr(a); a3 = fq12_mul(t, a);
  ac = fq12_conj(a); a3c = fq12_conj(a3);                         // the four multipliers once, so that a digit hands over a POINTER (a copy + a conjugation per digit were ~3 KB of traffic each)
  bool started = false; int run = 0;                              // squarings owed, done as one in-register run (fq12_cyclotomic_sqr_n)
  for (int i = 0; i < E1_WNAF_DIGITS; ++i) {
    uint32_t nz = 0, ng = 0, th = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { nz = (j == (i >> 5)) ? e1_wnaf_nz_word(j) : nz; ng = (j == (i >> 5)) ? e1_wnaf_neg_word(j) : ng; th = (j == (i >> 5)) ? e1_wnaf_three_word(j) : th; }
    if (started) ++run;
    if ((nz >> (i & 31)) & 1) {                                   // wave-uniform (compile-time tables)
      const bool three = (th >> (i & 31)) & 1, neg = (ng >> (i & 31)) & 1;
      const Fq12* m = three ? (neg ? &a3c : &a3) : (neg ? &ac : &a);
      if (started) { fq12_cyclotomic_sqr_n_mul(r, run, m); run = 0; } else { r = *m; started = true; }
    }
  }
  if (run) fq12_cyclotomic_sqr_n(r, run);
  return r;
}

// f^((q^12-1)/r), exact.  Four named Fq12 buffers (g, a, b, t) are reused; a destination never aliases a source.
// SHORT: the argument is the value of the 127-step loop (miller_g1_g2_short) and the result is raised to 2 x^2 - 1 on the way out, which makes it the
// Tate value: eta^(2x^2-1) = pi^2(eta)^2 conj(eta), one Frobenius, one cyclotomic squaring, one product (derivation at miller_g1_g2_short).
// (Round 2 found that the very same four calls placed in tate_short, after this function had returned, faulted inside the scratch aperture.  Round 3 found why:
//  this function keeps a base pointer (s34) and restores its stack pointer from it on the way out; fq12_frob<1>, as a called function, parked a literal in s34, which
//  this toolchain lets a callee do — the function returned with a garbage stack pointer and only callers that made no further call survived.  pi^1 is now inlined here
//  (fq12_frob_inl) and tools/check_base_pointer.py scans every object for the pattern.)
template <bool SHORT> ZKT_FN Fq12 final_exponentiation_t(const Fq12& f) {
  Fq12 g, a, b, t;
  t = fq12_inv(f);
  a = fq12_conj(f);
  g = fq12_mul(a, t);                      // ^(q^6-1)
  t = fq12_frob<2>(g);
  a = fq12_mul(t, g);                      // ^(q^2+1): easy part, now in a
  g = a;
  a = fq12_pow_e1(g);                      // ^e1
  t = fq12_pow_xabs(a); fq12_conj_ip(t);  // a^x (x < 0: conjugate = inverse in the cyclotomic subgroup)
  b = fq12_frob_inl<1>(a);
  a = fq12_mul(t, b);                      // a := ^(x+q)
  t = fq12_pow_xabs(a); fq12_conj_ip(t);
  b = fq12_pow_xabs(t); fq12_conj_ip(b);  // b = a^(x^2)
  t = fq12_frob<2>(a);
  b = fq12_mul(b, t) ;                     // (aliasing a source here costs one temporary; kept for clarity)
  t = fq12_conj(a);
  a = fq12_mul(b, t);                      // ^(x^2+q^2-1)
  if constexpr (!SHORT) return fq12_mul(a, g);
  else {
    t = fq12_mul(a, g);                    // eta
    b = fq12_frob<2>(t);                   // eta^(x^2): on G_T the q^2-Frobenius is the power by x^2
    a = fq12_cyclotomic_sqr(b);
    b = fq12_conj(t);
    return fq12_mul(a, b);
  }
}
ZKT_HD Fq12 final_exponentiation(const Fq12& f) { return final_exponentiation_t<false>(f); }
// f^(3 (q^12-1)/r) for the entry points that only DECIDE: the hard part as (x-1)^2 (x+q)(x^2+q^2-1) + 3, three times the exact exponent, reached without the
// division by three that costs the exact one its windowed 126-bit power (two more powers by x and a cube instead: ~18 Fq12 products fewer).  3 is prime to r, so
// "== 1" and "== the key's constant raised the same way" decide exactly what they decide with the exact exponent (oracle/fast_model.py final_exp_3h,
// tests/test_fast_model.py).  Values keep final_exponentiation_t.
ZKT_FN Fq12 final_exponentiation_3h(const Fq12& f) {
  Fq12 g, a, b, t;
  t = fq12_inv(f);
  a = fq12_conj(f);
  g = fq12_mul(a, t);                      // ^(q^6-1)
  t = fq12_frob<2>(g);
  a = fq12_mul(t, g);                      // ^(q^2+1): easy part
  g = a;
  t = fq12_pow_xabs(g); fq12_conj_ip(t);  // g^x
  b = fq12_conj(g);
  a = fq12_mul(t, b);                      // g^(x-1)
  t = fq12_pow_xabs(a); fq12_conj_ip(t);
  b = fq12_conj(a);
  a = fq12_mul(t, b);                      // g^((x-1)^2)
  t = fq12_pow_xabs(a); fq12_conj_ip(t);
  b = fq12_frob_inl<1>(a);                 // inlined: this function keeps a base pointer (see final_exponentiation_t)
  a = fq12_mul(t, b);                      // ^(x+q)
  t = fq12_pow_xabs(a); fq12_conj_ip(t);
  b = fq12_pow_xabs(t); fq12_conj_ip(b);  // a^(x^2)
  t = fq12_frob<2>(a);
  b = fq12_mul(b, t);
  t = fq12_conj(a);
  a = fq12_mul(b, t);                      // ^(x^2+q^2-1)
  t = fq12_cyclotomic_sqr(g);
  b = fq12_mul(t, g);                      // g^3
  return fq12_mul(a, b);
}

// One Tate pairing through the 127-step loop, with every precondition tested.  Returns TATE_ROUTE_SHORT and the value in r, or the route the element
// must take instead: TATE_ROUTE_LONG (Q on E' but outside G2: miller_g1_g2 + final_exponentiation) or TATE_ROUTE_EXACT (a point off its curve, or
// r P != infinity: miller_g1_g2_exact).  tate = eta^(2 x^2 - 1) for eta = final_exponentiation(f_{x^2,P}(Q)): final_exponentiation_t<true>.
// Kept as ONE function so that the kernels that use it stay small: load, call, store.
enum { TATE_ROUTE_SHORT = 0, TATE_ROUTE_LONG = 1, TATE_ROUTE_EXACT = 2 };
// (two functions on purpose: the one that holds the Fq12 keeps a base pointer and must not call g2_in_subgroup, which parks literals in s34 — see ZKT_ATE_STEP)
ZKT_FN bool tate_short_value(const Fq& xp, const Fq& yp, const Fq2& xq, const Fq2& yq, Fq12& r) {
  bool in_g1;
  Fq12 f = miller_g1_g2_short(xp, yp, xq, yq, in_g1);
  if (!in_g1) return false;
  r = final_exponentiation_t<true>(f);
  return true;
}
ZKT_HD int tate_short(const Fq& xp, const Fq& yp, const Fq2& xq, const Fq2& yq, Fq12& r) {
  if (!g1_on_curve(xp, yp) || !g2_on_curve(xq, yq)) return TATE_ROUTE_EXACT;
  if (!g2_in_subgroup(xq, yq)) return TATE_ROUTE_LONG;
  return tate_short_value(xp, yp, xq, yq, r) ? TATE_ROUTE_SHORT : TATE_ROUTE_EXACT;
}

// ---- the 63-step loop (optimal ate) for the entry points that only DECIDE ---------------------------------------------------------
// verifier.rs:30-54, signature.rs:34-39, pinocchio/verifier.rs:31-85 compare GTPoints and return a bool.  For P in G1, Q in G2 the ate pairing
// a(Q, P) = f_{|x|,Q}(P)^((q^12-1)/r) is bilinear and non-degenerate on the same cyclic groups of prime order r, hence a(Q, P) = tate(P, Q)^c for ONE unit
// c (mod r): prod_k tate(P_k, Q_k) == 1  <=>  prod_k a(Q_k, P_k) == 1.  (c has no closed form the model could check — the two argument orders of the Tate
// pairing are tied by the Weil pairing — so the entry points that return VALUES keep the 127-step loop.)  Python model: oracle/fast_model.py
// (ate_product_is_one), checked against the faithful oracle's tate() products by tests/test_fast_model.py.
// The loop runs on Q (Jacobian over Fq2 on the twist E'); a line evaluated at P, multiplied by w and by Fq2 factors the final exponentiation kills, is
//   a0 xp + a1 yp w + c4 w^4       with (a0, a1, c4) in Fq2 depending on Q ONLY
// so a Q shared by a batch (a key's gamma, delta) is tabulated once (k_ate_key_prep, zkt_pairing.hip) and costs no point arithmetic in the lanes.
// The chain ends on [|x|] Q: the G2 membership test psi(Q) = [x] Q is free.  P in G1 no longer is (the chain on P is gone): g1_in_subgroup.
// The two steps and the end test are INLINED into the loops that use them (ZKT_ATE_STEP).  As called functions they broke their callers on this toolchain: a
// function with 64-byte aligned locals keeps its incoming stack pointer in s34 (the base pointer), the inter-procedural register allocation does not count s34 among
// the registers such a caller needs kept, and these three park literals there (found as a silent abort of k_pairing_product_check_ate; DESIGN.md §5 "A compiler limit").
// tools/check_base_pointer.py scans an object for that pattern.
#define ZKT_ATE_STEP ZKT_HD
struct AteLine { Fq2 a0, a1, c4; };
static constexpr int ATE_LINES = 68;                       // 63 doublings + 5 additions: |x| = 0xd201000000010000
static constexpr int ATE_LINE_WORDS = 6 * FqC::N;          // a table entry: the three Fq2 in the kernels' own limb form (not an ABI type)
// tangent at T scaled by 2YZ^3 xi:  a0 = -xi 3X^2 Z^2,  a1 = xi 2YZ^3,  c4 = 3X^3 - 2Y^2;  T <- 2T
ZKT_ATE_STEP void ate_dbl_step(Jac<Fq2Ops>& T, AteLine& l) {
  const Fq2 X = T.X, Y = T.Y, Z = T.Z;
  const Fq2 A = fq2_sqr(X), B = fq2_sqr(Y), C = fq2_sqr(B), ZZ = fq2_sqr(Z);
  const Fq2 D = fq2_dbl(fq2_sub(fq2_sub(fq2_sqr(fq2_add(X, B)), A), C));
  const Fq2 E = fq2_add(fq2_dbl(A), A);
  const Fq2 X3 = fq2_sub2(fq2_sqr(E), fq2_zero(), D);
  const Fq2 Y3 = fq2_sub(fq2_mul(E, fq2_sub(D, X3)), fq2_dbl(fq2_dbl(fq2_dbl(C))));
  const Fq2 Z3 = fq2_dbl(fq2_mul(Y, Z));
  l.a0 = fq2_mul_xi(fq2_neg(fq2_mul(E, ZZ)));
  l.a1 = fq2_mul_xi(fq2_mul(Z3, ZZ));
  l.c4 = fq2_sub(fq2_mul(E, X), fq2_dbl(B));
  T.X = X3; T.Y = Y3; T.Z = Z3;
}
// chord through T and Q scaled by Z3 xi:  a0 = -xi R,  a1 = xi Z3,  c4 = R xq - Z3 yq;  T <- T + Q.  T = +-Q (possible only for a Q of small order,
// outside G2) gives Z3 = 0, which every later step keeps: the membership test at the end of the loop then fails and the element takes its old route.
ZKT_ATE_STEP void ate_add_step(Jac<Fq2Ops>& T, const Fq2& xq, const Fq2& yq, AteLine& l) {
  const Fq2 X = T.X, Y = T.Y, Z = T.Z;
  const Fq2 ZZ = fq2_sqr(Z), H = fq2_sub(fq2_mul(xq, ZZ), X), Rr = fq2_sub(fq2_mul(fq2_mul(yq, ZZ), Z), Y);
  const Fq2 HH = fq2_sqr(H), HHH = fq2_mul(H, HH), V = fq2_mul(X, HH);
  const Fq2 X3 = fq2_sub2(fq2_sqr(Rr), HHH, V);
  const Fq2 Y3 = fq2_mulsub(Rr, fq2_sub(V, X3), Y, HHH);
  const Fq2 Z3 = fq2_mul(Z, H);
  l.a0 = fq2_mul_xi(fq2_neg(Rr));
  l.a1 = fq2_mul_xi(Z3);
  l.c4 = fq2_mulsub(Rr, xq, Z3, yq);
  T.X = X3; T.Y = Y3; T.Z = Z3;
}
// T == psi(Q) up to sign: T = [|x|] Q and x < 0, so Q in G2 <=> T == -psi(Q) (g2_in_subgroup above, with the chain already done)
ZKT_ATE_STEP bool ate_end_is_psi(const Jac<Fq2Ops>& T, const Fq2& xq, const Fq2& yq) {
  if (fq2_is_zero(T.Z)) return false;
  const Fq2 px = fq2_mul(fq2_const([](int i) { return g2_psi_x_limb(0, i); }, [](int i) { return g2_psi_x_limb(1, i); }), fq2_conj(xq));
  const Fq2 py = fq2_mul(fq2_const([](int i) { return g2_psi_y_limb(0, i); }, [](int i) { return g2_psi_y_limb(1, i); }), fq2_conj(yq));
  const Fq2 ZZ = fq2_sqr(T.Z);
  return fq2_eq(T.X, fq2_mul(px, ZZ)) && fq2_eq(T.Y, fq2_neg(fq2_mul(py, fq2_mul(ZZ, T.Z))));
}
// P in G1 <=> P on E (the caller's check) and [x^2] P == (BETA x, -y): the 127-step chain on P alone (no lines)
ZKT_FN bool g1_in_subgroup(const Fq& xp, const Fq& yp) {
  const Aff<FqOps> p{xp, yp, false};
  Jac<FqOps> a = jac_from_aff(p);
  for (int i = 0; i < MILLER_X2_NBITS; ++i) {
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) w = (j == (i >> 5)) ? miller_x2_bits_word(j) : w;
    a = jac_dbl(a);
    if ((w >> (i & 31)) & 1) a = jac_add_aff(a, p);
  }
  if (jac_is_inf(a)) return false;
  const Fq ZZ = fp_sqr(a.Z);
  return fp_eq(fp_mul(fp_mul(g1_beta_const(), xp), ZZ), a.X) && fp_eq(fp_mul(fp_mul(yp, ZZ), a.Z), fp_neg(a.Y));
}
// f * (c0 + c1 w + c4 w^4) = (x + y w)(A + c1 w),  A = c0 + c4 v^2:  13 Fq2 products.
//   x A and (x + y)(A + c1) are products with the sparse Fq6 element (s0, 0, s4): x0 s0, x2 s4, (x0 + x2)(s0 + s4), x1 s0, x1 s4 — five each;
//   y c1 is three; then c0' = xA + v (y c1),  c1' = (x + y)(A + c1) - xA - y c1.
ZKT_HD void fq6_mul_sparse04(const Fq6& x, const Fq2& s0, const Fq2& s4, Fq2& r0, Fq2& r1, Fq2& r2) {
  const Fq2 p0 = fq2_mul(x.c0, s0), p2 = fq2_mul(x.c2, s4), q0 = fq2_mul(x.c1, s0), q4 = fq2_mul(x.c1, s4);
  r2 = fq2_subsub(fq2_mul(fq2_add(x.c0, x.c2), fq2_add(s0, s4)), p0, p2);      // x0 s4 + x2 s0
  r0 = fq2_add_mul_xi(p0, q4);                                                 // x0 s0 + xi x1 s4
  r1 = fq2_add_mul_xi(q0, p2);                                                 // x1 s0 + xi x2 s4
}
ZKT_HD Fq12 fq12_mul_ate_line_body(const Fq12& f, const Fq2& c0, const Fq2& c1, const Fq2& c4) {
  const Fq6 &x = f.c0, &y = f.c1;
  Fq2 a0, a1, a2, s0, s1, s2;
  fq6_mul_sparse04(x, c0, c4, a0, a1, a2);
  const Fq2 b0 = fq2_mul(y.c0, c1), b1 = fq2_mul(y.c1, c1), b2 = fq2_mul(y.c2, c1);
  fq6_mul_sparse04(fq6_add(x, y), fq2_add(c0, c1), c4, s0, s1, s2);
  Fq12 r;
  r.c0.c0 = fq2_add_mul_xi(a0, b2); r.c0.c1 = fq2_add(a1, b0); r.c0.c2 = fq2_add(a2, b1);
  r.c1.c0 = fq2_subsub(s0, a0, b0); r.c1.c1 = fq2_subsub(s1, a1, b1); r.c1.c2 = fq2_subsub(s2, a2, b2);
  return r;
}
ZKT_FQ12 Fq12 fq12_mul_ate_line(const Fq12& f, const Fq2& c0, const Fq2& c1, const Fq2& c4) { return fq12_mul_ate_line_body(f, c0, c1, c4); }
ZKT_FN void fq12_mul_ate_line_ip(Fq12& f, const Fq2& c0, const Fq2& c1, const Fq2& c4) {      // f <- f * line in place (see fq12_mul_line_ip)
  ZKT_FORCE_FRAME();
  const Fq12 t = fq12_mul_ate_line_body(f, c0, c1, c4);
  f = t;
}
// f <- f^2 * line in place: the 63-step loop's shared squaring fused with the first pair's line (see fq12_sqr_mul_line)
ZKT_FN void fq12_sqr_mul_ate_line(Fq12& f, const Fq2& c0, const Fq2& c1, const Fq2& c4) {
  ZKT_FORCE_FRAME();
  const Fq12 t = fq12_sqr_body(f);
  f = fq12_mul_ate_line_body(t, c0, c1, c4);
}
ZKT_HD AteLine ld_ate_line(const uint32_t* p) {
  AteLine l; Fq* e[6] = {&l.a0.c0, &l.a0.c1, &l.a1.c0, &l.a1.c1, &l.c4.c0, &l.c4.c1};
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int i = 0; i < FqC::N; ++i) e[k]->v[i] = p[k * FqC::N + i];
  return l;
}
ZKT_HD void st_ate_line(uint32_t* p, const AteLine& l) {
  const Fq* e[6] = {&l.a0.c0, &l.a0.c1, &l.a1.c0, &l.a1.c1, &l.c4.c0, &l.c4.c1};
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int i = 0; i < FqC::N; ++i) p[k * FqC::N + i] = e[k]->v[i];
}
ZKT_HD bool ate_bit(int i) { return (BLS_X_ABS >> i) & 1; }
// The 68 line triples of one Q in loop order; returns Q in G2 (Q on E' is the caller's check).
ZKT_FN bool ate_line_table(const Fq2& xq, const Fq2& yq, uint32_t* tab) {
  Jac<Fq2Ops> T{xq, yq, fq2_one()};
  AteLine l; int li = 0;
  for (int i = 62; i >= 0; --i) {
    ate_dbl_step(T, l); st_ate_line(tab + (size_t)(li++) * ATE_LINE_WORDS, l);
    if (ate_bit(i)) { ate_add_step(T, xq, yq, l); st_ate_line(tab + (size_t)(li++) * ATE_LINE_WORDS, l); }
  }
  return ate_end_is_psi(T, xq, yq);
}
// prod_{k<KV} f_{|x|,Q_k}(P_k) * prod_{j<KF} f_{|x|,Q'_j}(P_{KV+j}) up to factors the final exponentiation kills.  The KV pairs bring their own Q_k (chain in
// the lane), the KF pairs the line table of a shared Q'_j.  q_in_g2 <- every Q_k is in G2 (Q_k on E' is the caller's check, as are all the P's).
template <int KV, int KF>
ZKT_FN Fq12 miller_ate_multi(const Fq* xp, const Fq* yp, const Fq2* xq, const Fq2* yq, const uint32_t* const* tab, bool& q_in_g2) {
  Jac<Fq2Ops> T[KV > 0 ? KV : 1];
  for (int k = 0; k < KV; ++k) T[k] = Jac<Fq2Ops>{xq[k], yq[k], fq2_one()};
  AteLine l;
  Fq12 f = fq12_one(), ft;
  int li = 0;
  for (int i = 62; i >= 0; --i) {
    // the step's squaring is owed until the first line of the step takes it along (fq12_sqr_mul_ate_line).  Fused for the two shapes that carry the protocol throughput —
    // Groth16 batch verification <1, 2> and signature verification <2, 0> — only: every fused instantiation adds ~1.5 min to the build of this object
    constexpr bool FUSE = true;      // (while the fused step had to be INLINED to keep its callers' base pointers safe, only two shapes were fused: ~1.5 min of compile time each)
    bool sq = i != 62;
    if (!FUSE && sq) { ft = fq12_sqr(f); f = ft; sq = false; }
    for (int k = 0; k < KV; ++k) {
      ate_dbl_step(T[k], l);
      if (sq) { fq12_sqr_mul_ate_line(f, fq2_mul_fq(l.a0, xp[k]), fq2_mul_fq(l.a1, yp[k]), l.c4); sq = false; }
      else fq12_mul_ate_line_ip(f, fq2_mul_fq(l.a0, xp[k]), fq2_mul_fq(l.a1, yp[k]), l.c4);
    }
    for (int j = 0; j < KF; ++j) {
      l = ld_ate_line(tab[j] + (size_t)li * ATE_LINE_WORDS);
      if (sq) { fq12_sqr_mul_ate_line(f, fq2_mul_fq(l.a0, xp[KV + j]), fq2_mul_fq(l.a1, yp[KV + j]), l.c4); sq = false; }
      else fq12_mul_ate_line_ip(f, fq2_mul_fq(l.a0, xp[KV + j]), fq2_mul_fq(l.a1, yp[KV + j]), l.c4);
    }
    if (sq) { ft = fq12_sqr(f); f = ft; }                                        // no pair at all (KV + KF = 0 is not instantiated; kept for completeness)
    ++li;
    if (ate_bit(i)) {                                                          // wave-uniform
      for (int k = 0; k < KV; ++k) {
        ate_add_step(T[k], xq[k], yq[k], l);
        fq12_mul_ate_line_ip(f, fq2_mul_fq(l.a0, xp[k]), fq2_mul_fq(l.a1, yp[k]), l.c4);
      }
      for (int j = 0; j < KF; ++j) {
        l = ld_ate_line(tab[j] + (size_t)li * ATE_LINE_WORDS);
        fq12_mul_ate_line_ip(f, fq2_mul_fq(l.a0, xp[KV + j]), fq2_mul_fq(l.a1, yp[KV + j]), l.c4);
      }
      ++li;
    }
  }
  q_in_g2 = true;
  for (int k = 0; k < KV; ++k) q_in_g2 = q_in_g2 && ate_end_is_psi(T[k], xq[k], yq[k]);
  return f;
}

// ---- raw Miller values and the Weil pairing, bit-exact (row a14) ------------------------------------
//   calc_g1_g2 / calc_g2_g1   pairing.rs:20-55      weil   pairing.rs:75-84
// These are NOT normalisation independent, so the projective loop keeps a numerator N and a denominator D
// (Fq12) plus the exact Fq / Fq2 scale factors of every line and vertical, and divides once at the end:
// f = (N*ns)/(D*ds) is the very element the reference builds with one Fq12 inversion per step
// (python model: oracle/fast_model.py calc_g1_g2_exact / calc_g2_g1_exact, tests/test_fast_model.py).
// Used by the reference's tests only, so the sparse structure is not exploited: dense Fq12 products.
ZKT_HD Fq12 fq12_sparse(const Fq2& c00, const Fq2& c02, const Fq2& c11, const Fq2& c12) {   // c00 + c02 v^2 + c11 v w + c12 v^2 w
  Fq12 r; r.c0 = Fq6{c00, fq2_zero(), c02}; r.c1 = Fq6{fq2_zero(), c11, c12}; return r;
}
ZKT_FN Fq12 fq12_scale_fq2(const Fq12& a, const Fq2& s) {
  Fq12 r;
  r.c0 = Fq6{fq2_mul(a.c0.c0, s), fq2_mul(a.c0.c1, s), fq2_mul(a.c0.c2, s)};
  r.c1 = Fq6{fq2_mul(a.c1.c0, s), fq2_mul(a.c1.c1, s), fq2_mul(a.c1.c2, s)};
  return r;
}
ZKT_HD bool miller_bit(int i) {
  uint32_t w = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) w = (j == (i >> 5)) ? miller_bits_word(j) : w;
  return (w >> (i & 31)) & 1;
}

// calc_g1_g2(P, Q): Miller loop on P in G1 (Fq Jacobian), evaluated at the untwisted Q, for ANY affine P (pairing.rs:20-55).
// `bad` <- the reference would have panicked: a multiple of P met by the binary chain is infinity ("Both points need to be rational",
// rational_function.rs:36), or a denominator vanished (inverse of zero).  V == P inside an addition step is the reference's tangent case
// (rational_function.rs:25-27); neither happens for P of order r.
ZKT_FN Fq12 miller_g1_g2_exact(const Fq& xp, const Fq& yp, const Fq2& xq, const Fq2& yq, bool& bad) {
  const Fq2 xi_inv = xi_inv_const();
  const Fq2 Xq = fq2_mul(xq, xi_inv), Yq = fq2_mul(yq, xi_inv);
  Fq X = xp, Y = yp, Z = fp_one<FqC>();
  Fq12 N = fq12_one(), D = fq12_one();
  Fq ns = fp_one<FqC>(), ds = fp_one<FqC>();
  bad = false;
  // tangent at V and vertical at 2V, scaled; V <- 2V.  square = the doubling step proper (f <- f^2 l/v); otherwise f <- f l/v (V == P in an addition step)
  auto dbl = [&](bool square) {
    Fq A = fp_sqr(X), B = fp_sqr(Y), C = fp_sqr(B), ZZ = fp_sqr(Z);
    Fq Dd = fp_dbl(fp_sub(fp_sub(fp_sqr(fp_add(X, B)), A), C));
    Fq E = fp_add(fp_dbl(A), A);
    Fq X3 = fp_sub(fp_sqr(E), fp_dbl(Dd));
    Fq Y3 = fp_sub(fp_mul(E, fp_sub(Dd, X3)), fp_dbl(fp_dbl(fp_dbl(C))));
    Fq Z3 = fp_dbl(fp_mul(Y, Z));
    Fq s = fp_mul(Z3, ZZ), Z3Z3 = fp_sqr(Z3);
    Fq12 lp = fq12_sparse(fq2_from_fq(fp_sub(fp_mul(E, X), fp_dbl(B))), fq2_mul_fq(Xq, fp_neg(fp_mul(E, ZZ))), fq2_mul_fq(Yq, s), fq2_zero());
    Fq12 vp = fq12_sparse(fq2_from_fq(fp_neg(X3)), fq2_mul_fq(Xq, Z3Z3), fq2_zero(), fq2_zero());
    if (square) { N = fq12_mul(fq12_sqr(N), lp); D = fq12_mul(fq12_sqr(D), vp); ns = fp_mul(fp_sqr(ns), Z3Z3); ds = fp_mul(fp_sqr(ds), s); }
    else { N = fq12_mul(N, lp); D = fq12_mul(D, vp); ns = fp_mul(ns, Z3Z3); ds = fp_mul(ds, s); }
    X = X3; Y = Y3; Z = Z3;
    if (fp_is_zero(Z3)) bad = true;                        // 2V = infinity
  };
  for (int i = 0; i < MILLER_NBITS && !bad; ++i) {
    dbl(true);
    if (bad) break;
    if (miller_bit(i)) {
      Fq ZZ = fp_sqr(Z), H = fp_sub(fp_mul(xp, ZZ), X), Rr = fp_sub(fp_mul(fp_mul(yp, ZZ), Z), Y);
      if (fp_is_zero(H)) {
        if (fp_is_zero(Rr)) { dbl(false); continue; }       // V == P: tangent, V + P = 2P
        bad = true; break;                                  // V == -P: V + P = infinity
      }
      Fq HH = fp_sqr(H), HHH = fp_mul(H, HH), V = fp_mul(X, HH);
      Fq X3 = fp_sub(fp_sub(fp_sqr(Rr), HHH), fp_dbl(V));
      Fq Y3 = fp_sub(fp_mul(Rr, fp_sub(V, X3)), fp_mul(Y, HHH));
      Fq Z3 = fp_mul(Z, H), Z3Z3 = fp_sqr(Z3);
      Fq12 lp = fq12_sparse(fq2_from_fq(fp_sub(fp_mul(Rr, xp), fp_mul(Z3, yp))), fq2_mul_fq(Xq, fp_neg(Rr)), fq2_mul_fq(Yq, Z3), fq2_zero());
      Fq12 vp = fq12_sparse(fq2_from_fq(fp_neg(X3)), fq2_mul_fq(Xq, Z3Z3), fq2_zero(), fq2_zero());
      N = fq12_mul(N, lp); D = fq12_mul(D, vp);
      ns = fp_mul(ns, Z3Z3); ds = fp_mul(ds, Z3);
      X = X3; Y = Y3; Z = Z3;
    }
  }
  if (bad) return fq12_one();
  const Fq12 den = fq12_scale_fq2(D, fq2_from_fq(ds));
  if (fq12_is_zero(den)) { bad = true; return fq12_one(); }   // a vertical line through Q: the reference's inv() of zero
  return fq12_mul(fq12_scale_fq2(N, fq2_from_fq(ns)), fq12_inv(den));
}

// calc_g2_g1(Q, P): Miller loop on Q in G2 (Fq2 Jacobian), evaluated at the embedded P.  With x' = (x/xi) v^2 and
// y' = (y/xi) v w an affine slope lam becomes (lam/xi) v^2 w, so
//   line(P) = yp + [(lam x1 - y1)/xi] v w + [-(lam/xi) xp] v^2 w,   vertical(P) = xp - (x/xi) v^2.
ZKT_FN Fq12 miller_g2_g1_exact(const Fq2& xq, const Fq2& yq, const Fq& xp, const Fq& yp) {
  const Fq2 xi_inv = xi_inv_const();
  const Fq nxp = fp_neg(xp);
  Fq2 X = xq, Y = yq, Z = fq2_one();
  Fq12 N = fq12_one(), D = fq12_one();
  Fq2 ns = fq2_one(), ds = fq2_one();
  for (int i = 0; i < MILLER_NBITS; ++i) {
    {
      Fq2 A = fq2_sqr(X), B = fq2_sqr(Y), C = fq2_sqr(B), ZZ = fq2_sqr(Z);
      Fq2 Dd = fq2_dbl(fq2_sub(fq2_sub(fq2_sqr(fq2_add(X, B)), A), C));
      Fq2 E = fq2_add(fq2_dbl(A), A);
      Fq2 X3 = fq2_sub(fq2_sqr(E), fq2_dbl(Dd));
      Fq2 Y3 = fq2_sub(fq2_mul(E, fq2_sub(Dd, X3)), fq2_dbl(fq2_dbl(fq2_dbl(C))));
      Fq2 Z3 = fq2_dbl(fq2_mul(Y, Z));
      Fq2 s = fq2_mul(Z3, ZZ), Z3Z3 = fq2_sqr(Z3);
      Fq12 lp = fq12_sparse(fq2_mul_fq(s, yp), fq2_zero(), fq2_mul(fq2_sub(fq2_mul(E, X), fq2_dbl(B)), xi_inv),
                            fq2_mul_fq(fq2_mul(fq2_mul(E, ZZ), xi_inv), nxp));
      Fq12 vp = fq12_sparse(fq2_mul_fq(Z3Z3, xp), fq2_neg(fq2_mul(X3, xi_inv)), fq2_zero(), fq2_zero());
      N = fq12_mul(fq12_sqr(N), lp); D = fq12_mul(fq12_sqr(D), vp);
      ns = fq2_mul(fq2_sqr(ns), Z3Z3); ds = fq2_mul(fq2_sqr(ds), s);
      X = X3; Y = Y3; Z = Z3;
    }
    if (miller_bit(i)) {
      Fq2 ZZ = fq2_sqr(Z), H = fq2_sub(fq2_mul(xq, ZZ), X), Rr = fq2_sub(fq2_mul(fq2_mul(yq, ZZ), Z), Y);
      Fq2 HH = fq2_sqr(H), HHH = fq2_mul(H, HH), V = fq2_mul(X, HH);
      Fq2 X3 = fq2_sub(fq2_sub(fq2_sqr(Rr), HHH), fq2_dbl(V));
      Fq2 Y3 = fq2_sub(fq2_mul(Rr, fq2_sub(V, X3)), fq2_mul(Y, HHH));
      Fq2 Z3 = fq2_mul(Z, H), Z3Z3 = fq2_sqr(Z3);
      Fq12 lp = fq12_sparse(fq2_mul_fq(Z3, yp), fq2_zero(), fq2_mul(fq2_sub(fq2_mul(Rr, xq), fq2_mul(Z3, yq)), xi_inv),
                            fq2_mul_fq(fq2_mul(Rr, xi_inv), nxp));
      Fq12 vp = fq12_sparse(fq2_mul_fq(Z3Z3, xp), fq2_neg(fq2_mul(X3, xi_inv)), fq2_zero(), fq2_zero());
      N = fq12_mul(N, lp); D = fq12_mul(D, vp);
      ns = fq2_mul(ns, Z3Z3); ds = fq2_mul(ds, Z3);
      X = X3; Y = Y3; Z = Z3;
    }
  }
  return fq12_mul(fq12_scale_fq2(N, ns), fq12_inv(fq12_scale_fq2(D, ds)));
}

}  // namespace zkt
