// Self-test program for the lazily reduced Fq arithmetic (fp.h, W = 28), shared by the host build (hostcheck.cpp, CPU tests) and the
// device diagnostic entry point zkt_selftest_fq_program (zkt_field.hip, GPU tests): the same straight-line program must give the same
// residues as plain mod-p arithmetic on both compilers.
#pragma once
#include "abi.h"

namespace zkt {

// Lazy-limb invariants of the W = 28 field (fp.h): every value a function returns has limbs < 2^28 and is < 4p.
ZKT_HD bool fq_lazy_ok(const Fq& x) {
  uint32_t hi = 0;
  for (int i = 0; i < FqC::N; ++i) hi |= x.v[i] >> 28;
  if (hi) return false;
  for (int i = FqC::N - 1; i >= 0; --i) {            // x < 4p, limb-wise from the top
    const uint32_t p4 = (uint32_t)((((uint64_t)FqC::mod(i) << 2) | (i ? FqC::mod(i - 1) >> 26 : 0)) & 0x0fffffffu);
    if (x.v[i] != p4) return x.v[i] < p4;
  }
  return false;
}
// the same residue as x, moved to x' = canonical(x) + j*p (j < 3): a non-canonical representative a kernel may meet
ZKT_HD Fq fq_lift(const Fq& x, int j) {
  uint32_t w[FqC::ABI_N]; fp_to_words(x, w);
  Fq c = fp_from_words<FqC>(w), r; uint32_t carry = 0;   // c < 1.01p
  for (int i = 0; i < FqC::N; ++i) { uint32_t t = c.v[i] + FqC::kp(j, i) + carry; r.v[i] = t & 0x0fffffffu; carry = t >> 28; }
  return r;
}
// A pseudo-random straight-line program over 4 Fq registers (ops and operands drawn from an LCG the test replays in
// python): exercises long add/sub/neg chains that push values towards the 4p bound, non-canonical representatives,
// and the zero / equality tests on them.  Returns the number of invariant violations; out = the 4 registers, canonical.
ZKT_HD int fq_program(uint64_t seed, int steps, const uint32_t* in4, uint32_t* out4) {
  Fq r[4]; int bad = 0;
  for (int i = 0; i < 4; ++i) r[i] = ld_fp<FqC>(in4 + i * FqC::ABI_N);
  uint64_t st = seed;
  for (int k = 0; k < steps; ++k) {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    const int op = (int)((st >> 33) % 16), d = (int)((st >> 40) & 3), a = (int)((st >> 42) & 3), b = (int)((st >> 44) & 3), j = (int)((st >> 46) % 3);
    switch (op) {
      case 0: case 1: r[d] = fp_add(r[a], r[b]); break;
      case 2: case 3: r[d] = fp_sub(r[a], r[b]); break;
      case 4: r[d] = fp_neg(r[a]); break;
      case 5: r[d] = fp_dbl(r[a]); break;
      case 6: r[d] = fp_mul(r[a], r[b]); break;
      case 7: r[d] = fp_sqr(r[a]); break;
      case 8: r[d] = fq_lift(r[a], j); break;
      case 10: r[d] = fp_sub2(r[a], r[b], r[(b + 1) & 3]); break;                       // a - b - 2c, one reduction
      case 11: r[d] = fp_mulsub(r[a], r[b], r[(a + 1) & 3], r[(b + 2) & 3]); break;     // a b - c d, one Montgomery reduction
      case 12: r[d] = fp_muladd(r[a], r[b], r[(a + 1) & 3], r[(b + 2) & 3]); break;     // a b + c d
      case 13: r[d] = fp_add3(r[a], r[b], r[(b + 1) & 3]); break;
      case 14: r[d] = fp_addsub(r[a], r[b], r[(b + 1) & 3]); break;
      case 15: r[d] = fp_subsub(r[a], r[b], r[(b + 1) & 3]); break;
      default: {                                           // predicates must see through the representative
        Fq z = fp_sub(fq_lift(r[a], j), r[a]);
        if (!fp_is_zero(z) || !fp_eq(fq_lift(r[a], j), r[a]) || fp_is_zero(r[a]) != fp_is_zero(fq_lift(r[a], (j + 1) % 3))) ++bad;
        r[d] = fp_eq(r[a], r[b]) ? fp_one<FqC>() : fp_add(r[a], fp_one<FqC>());
      }
    }
    if (!fq_lazy_ok(r[d])) ++bad;
  }
  for (int i = 0; i < 4; ++i) st_fp<FqC>(out4 + i * FqC::ABI_N, r[i]);
  return bad;
}
}  // namespace zkt
