// Boundary layouts (include/zkt.h) <-> internal Montgomery values.
// ABI values are canonical residues, little-endian 64-bit limbs == little-endian
// 32-bit limbs byte for byte, in the reference's struct field order:
//   Fq2 {u1,u0} (fq2.rs:16-19), Fq6 {v2,v1,v0} (fq6.rs:16-20), Fq12 {w1,w0} (fq12.rs:18-21),
//   points {x, y, is_infinity} (g1_point.rs:32-36, g2_point.rs:30-34).
#pragma once
#include "pairing.h"

namespace zkt {

static constexpr int ABI_G1_WORDS = 26;   // u32 words per zkt_g1_affine (104 B)
static constexpr int ABI_G2_WORDS = 50;   // zkt_g2_affine (200 B)
static constexpr int ABI_SECP_WORDS = 18; // zkt_secp_affine (72 B)

template <class C> ZKT_HD Fp<C> ld_raw(const uint32_t* p) { Fp<C> r;
#pragma unroll
  for (int i = 0; i < C::N; ++i) r.v[i] = p[i]; return r; }
template <class C> ZKT_HD void st_raw(uint32_t* p, const Fp<C>& a) {
#pragma unroll
  for (int i = 0; i < C::N; ++i) p[i] = a.v[i]; }
// canonical in memory (C::ABI_N words) <-> Montgomery in registers (C::N limbs)
template <class C> ZKT_HD Fp<C> ld_fp(const uint32_t* p) { return fp_from_words<C>(p); }
template <class C> ZKT_HD void st_fp(uint32_t* p, const Fp<C>& a) { fp_to_words(a, p); }

ZKT_HD Fq2 ld_fq2(const uint32_t* p) { Fq2 r; r.c1 = ld_fp<FqC>(p); r.c0 = ld_fp<FqC>(p + 12); return r; }
ZKT_HD void st_fq2(uint32_t* p, const Fq2& a) { st_fp<FqC>(p, a.c1); st_fp<FqC>(p + 12, a.c0); }
ZKT_HD Fq6 ld_fq6(const uint32_t* p) { Fq6 r; r.c2 = ld_fq2(p); r.c1 = ld_fq2(p + 24); r.c0 = ld_fq2(p + 48); return r; }
ZKT_HD void st_fq6(uint32_t* p, const Fq6& a) { st_fq2(p, a.c2); st_fq2(p + 24, a.c1); st_fq2(p + 48, a.c0); }
ZKT_HD Fq12 ld_fq12(const uint32_t* p) { Fq12 r; r.c1 = ld_fq6(p); r.c0 = ld_fq6(p + 72); return r; }
ZKT_HD void st_fq12(uint32_t* p, const Fq12& a) { st_fq6(p, a.c1); st_fq6(p + 72, a.c0); }

// point loaders per coordinate field
template <class F> struct PtIO;
template <class C> struct PtIO<PrimeOps<C>> {
  static constexpr int A = C::ABI_N, WORDS = 2 * A + 2;
  ZKT_HD static Aff<PrimeOps<C>> ld(const uint32_t* p) {
    Aff<PrimeOps<C>> a; a.inf = p[2 * A] != 0;
    a.x = ld_fp<C>(p); a.y = ld_fp<C>(p + A);
    return a;
  }
  ZKT_HD static void st(uint32_t* p, const Aff<PrimeOps<C>>& a) {
    if (a.inf) {
#pragma unroll
      for (int i = 0; i < 2 * A; ++i) p[i] = 0;
      p[2 * A] = 1; p[2 * A + 1] = 0; return;
    }
    st_fp<C>(p, a.x); st_fp<C>(p + A, a.y); p[2 * A] = 0; p[2 * A + 1] = 0;
  }
};
template <> struct PtIO<Fq2Ops> {
  static constexpr int WORDS = 50;
  ZKT_HD static Aff<Fq2Ops> ld(const uint32_t* p) {
    Aff<Fq2Ops> a; a.inf = p[48] != 0; a.x = ld_fq2(p); a.y = ld_fq2(p + 24); return a;
  }
  ZKT_HD static void st(uint32_t* p, const Aff<Fq2Ops>& a) {
    if (a.inf) {
      for (int i = 0; i < 48; ++i) p[i] = 0;
      p[48] = 1; p[49] = 0; return;
    }
    st_fq2(p, a.x); st_fq2(p + 24, a.y); p[48] = 0; p[49] = 0;
  }
};

}  // namespace zkt
