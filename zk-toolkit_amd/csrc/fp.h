// Prime-field arithmetic for gfx950 (CDNA4), Montgomery form internally, canonical
// residues in [0,p) at every ABI boundary.  Two limb layouts, chosen per field by C::W:
//   W = 32  N x 32-bit limbs, values always canonical (< p): Fr and the secp256k1 fields.
//   W = 28  N x 28-bit limbs, R = 2^(28N), values lazily reduced (< 4p, limbs < 2^28):
//           BLS12-381 Fq, the field under every hot kernel.  A column of the product
//           scan sums 2N products < 2^56 and fits one 64-bit accumulator, so the
//           multiply is pure v_mad_u64_u32 (no v_addc per product), squaring can share
//           cross products, and add/sub are carry-free limb adds plus one fused
//           "subtract floor-estimate * p and propagate" pass (fp_lazy_reduce).
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

// Gives the enclosing function a 64-byte aligned stack object, i.e. a realigned frame with its own base pointer (s34 saved on entry, restored on exit).  For the few
// real functions that hold whole Fq12 values in REGISTERS (round 4: the in-register squaring runs and fused Miller steps): without a frame of their own the register allocator
// hands them s34 as scratch, and their callers keep their base pointer there (DESIGN §5 "A compiler limit"; tools/check_base_pointer.py flags exactly this).
#define ZKT_FORCE_FRAME() do { alignas(64) volatile uint32_t zkt_frame_anchor_[16]; zkt_frame_anchor_[0] = 0u; } while (0)

template <class C> ZKT_HD Fp<C> fp_zero() { Fp<C> r;
#pragma unroll
  for (int i = 0; i < C::N; ++i) r.v[i] = 0; return r; }
template <class C> ZKT_HD Fp<C> fp_one() { Fp<C> r;   // Montgomery 1 = R mod p
#pragma unroll
  for (int i = 0; i < C::N; ++i) r.v[i] = C::one(i); return r; }
// W = 28: a lazily reduced value is zero iff it is one of 0, p, 2p, 3p.  The first limbs of
// those four differ from a random limb with probability 1 - 2^-26, so test limb 0 first.
template <class C> ZKT_HD bool fp_is_zero(const Fp<C>& a) {
  if constexpr (C::W == 28) {
    const uint32_t a0 = a.v[0];
    if (!((a0 == C::kp(0, 0)) | (a0 == C::kp(1, 0)) | (a0 == C::kp(2, 0)) | (a0 == C::kp(3, 0)))) return false;
    uint32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0;
#pragma unroll
    for (int i = 0; i < C::N; ++i) { o0 |= a.v[i]; o1 |= a.v[i] ^ C::kp(1, i); o2 |= a.v[i] ^ C::kp(2, i); o3 |= a.v[i] ^ C::kp(3, i); }
    return (o0 == 0) | (o1 == 0) | (o2 == 0) | (o3 == 0);
  } else {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < C::N; ++i) o |= a.v[i];
    return o == 0;
  }
}

// t (N limbs + carry word `top`) -> t - p if t >= p.  Requires t < 2p.
template <class C> ZKT_HD void fp_cond_sub(Fp<C>& r, uint32_t top) {
  uint32_t s[C::N]; uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < C::N; ++i) s[i] = subb(r.v[i], C::mod(i), bw);
  bool keep = (top == 0) && bw;   // t < p
#pragma unroll
  for (int i = 0; i < C::N; ++i) r.v[i] = keep ? r.v[i] : s[i];
}

// W = 28.  v: limbs < 2^31, value < 12p.  Returns the same residue with limbs < 2^28 and value < 4p.
// q = floor(v_top * QEST_M / 2^32) never exceeds floor(v/p) and is at most 3 below it for v < 12p
// (checked exhaustively over the reachable top limbs in tests/test_hostcheck.py); the pass adds
// q*(2^(WN) - p) limb by limb while propagating carries, and the 2^(WN) bit falls off the top.
template <class C> ZKT_HD Fp<C> fp_lazy_reduce(const uint32_t* v) {
  constexpr uint32_t M = (1u << C::W) - 1;
  Fp<C> r; const uint32_t q = (uint32_t)(((uint64_t)v[C::N - 1] * C::QEST_M) >> 32);
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < C::N; ++i) {
    uint64_t t = (uint64_t)q * C::comp(i) + (uint64_t)(v[i] + carry);
    r.v[i] = (uint32_t)t & M; carry = (uint32_t)(t >> C::W);
  }
  return r;
}

// plus (prime_field_elem.rs:278-286)
template <class C> ZKT_HD Fp<C> fp_add(const Fp<C>& a, const Fp<C>& b) {
  if constexpr (C::W == 28) {
    uint32_t v[C::N];
#pragma unroll
    for (int i = 0; i < C::N; ++i) v[i] = a.v[i] + b.v[i];
    return fp_lazy_reduce<C>(v);
  } else {
    Fp<C> r; uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < C::N; ++i) r.v[i] = addc(a.v[i], b.v[i], c);
    fp_cond_sub(r, c);
    return r;
  }
}
// minus (prime_field_elem.rs:288-300): a<b -> p-(b-a).  W = 28: a + 8p - b with 8p spread so no limb borrows.
template <class C> ZKT_HD Fp<C> fp_sub(const Fp<C>& a, const Fp<C>& b) {
  if constexpr (C::W == 28) {
    uint32_t v[C::N];
#pragma unroll
    for (int i = 0; i < C::N; ++i) v[i] = a.v[i] + C::subk(i) - b.v[i];
    return fp_lazy_reduce<C>(v);
  } else {
    Fp<C> r; uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < C::N; ++i) r.v[i] = subb(a.v[i], b.v[i], bw);
    uint32_t mask = 0u - bw, c = 0;
#pragma unroll
    for (int i = 0; i < C::N; ++i) r.v[i] = addc(r.v[i], C::mod(i) & mask, c);
    return r;
  }
}
// negate (prime_field_elem.rs:448-457): 0 stays 0
template <class C> ZKT_HD Fp<C> fp_neg(const Fp<C>& a) {
  if constexpr (C::W == 28) {
    return fp_sub(fp_zero<C>(), a);
  } else {
    Fp<C> r; uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < C::N; ++i) r.v[i] = subb(C::mod(i), a.v[i], bw);
    uint32_t mask = fp_is_zero(a) ? 0u : 0xffffffffu;
#pragma unroll
    for (int i = 0; i < C::N; ++i) r.v[i] &= mask;
    return r;
  }
}
template <class C> ZKT_HD bool fp_eq(const Fp<C>& a, const Fp<C>& b) {
  if constexpr (C::W == 28) {
    return fp_is_zero(fp_sub(a, b));
  } else {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < C::N; ++i) o |= a.v[i] ^ b.v[i];
    return o == 0;
  }
}
template <class C> ZKT_HD Fp<C> fp_dbl(const Fp<C>& a) { return fp_add(a, a); }
// three-term sums in ONE reduction pass (the tower's Karatsuba recombinations): all operands normalised (< 4p, limbs < 2^28)
//   fp_add3   a + b + c          limbs < 3*2^28, value < 12p
//   fp_addsub a + b - c          a + b + 8p - c < 16p
//   fp_subsub a - b - c          a + 16p - b - c < 20p (the 16p spread covers two subtrahend limbs)
template <class C> ZKT_HD Fp<C> fp_add3(const Fp<C>& a, const Fp<C>& b, const Fp<C>& c) {
  if constexpr (C::W == 28) {
    uint32_t v[C::N];
#pragma unroll
    for (int i = 0; i < C::N; ++i) v[i] = a.v[i] + b.v[i] + c.v[i];
    return fp_lazy_reduce<C>(v);
  } else return fp_add(fp_add(a, b), c);
}
template <class C> ZKT_HD Fp<C> fp_addsub(const Fp<C>& a, const Fp<C>& b, const Fp<C>& c) {
  if constexpr (C::W == 28) {
    uint32_t v[C::N];
#pragma unroll
    for (int i = 0; i < C::N; ++i) v[i] = a.v[i] + b.v[i] + C::subk(i) - c.v[i];
    return fp_lazy_reduce<C>(v);
  } else return fp_sub(fp_add(a, b), c);
}
template <class C> ZKT_HD Fp<C> fp_subsub(const Fp<C>& a, const Fp<C>& b, const Fp<C>& c) {
  if constexpr (C::W == 28) {
    uint32_t v[C::N];
#pragma unroll
    for (int i = 0; i < C::N; ++i) v[i] = a.v[i] + C::subk3(i) - b.v[i] - c.v[i];
    return fp_lazy_reduce<C>(v);
  } else return fp_sub(fp_sub(a, b), c);
}
// a - b - 2c in one reduction pass (the x-coordinate of every addition formula).  W = 28: a + 16p - b - 2c with 16p spread so that
// no limb borrows (limbs < 2^31, value < 20p: the quotient estimate of fp_lazy_reduce still leaves < 4p, tests/test_hostcheck.py).
template <class C> ZKT_HD Fp<C> fp_sub2(const Fp<C>& a, const Fp<C>& b, const Fp<C>& c) {
  if constexpr (C::W == 28) {
    uint32_t v[C::N];
#pragma unroll
    for (int i = 0; i < C::N; ++i) v[i] = a.v[i] + C::subk3(i) - b.v[i] - 2 * c.v[i];
    return fp_lazy_reduce<C>(v);
  } else {
    return fp_sub(fp_sub(a, b), fp_dbl(c));
  }
}

// acc += a*b on one 64-bit column accumulator (W = 28 path): a single v_mad_u64_u32
ZKT_HD uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * b + c; }

// Montgomery product a*b*R^-1 mod p.  times (prime_field_elem.rs:302-308) in the Montgomery domain.
// W = 32: inputs and output in [0,p).
// W = 28: inputs < 4p with limbs < 2^28; output (ab + mp)/R < p(1 + 16p/R) < 1.01p, limbs < 2^28.
//         Column sums: 28 products < 2^56 plus a carry < 2^36 stay below 2^61.
template <class C> ZKT_HD Fp<C> fp_mul_impl(const Fp<C>& a, const Fp<C>& b) {
  constexpr int N = C::N;
  Fp<C> r; uint32_t m[N];
  if constexpr (C::W == 28) {
    constexpr uint32_t M = (1u << 28) - 1;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
      for (int i = 0; i <= k; ++i) acc = mad64(a.v[i], b.v[k - i], acc);
#pragma unroll
      for (int j = 0; j < k; ++j) acc = mad64(m[j], C::mod(k - j), acc);
      m[k] = ((uint32_t)acc * C::INV) & M;
      acc = mad64(m[k], C::mod(0), acc);
      acc >>= 28;
    }
#pragma unroll
    for (int k = N; k < 2 * N; ++k) {
#pragma unroll
      for (int i = k - N + 1; i < N; ++i) acc = mad64(a.v[i], b.v[k - i], acc);
#pragma unroll
      for (int j = k - N + 1; j < N; ++j) acc = mad64(m[j], C::mod(k - j), acc);
      r.v[k - N] = (uint32_t)acc & M;
      acc >>= 28;
    }
    return r;
  } else {
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
}

// W = 28 squaring: each cross product once against the pre-doubled operand (limbs < 2^29, so the
// doubled products stay < 2^57 and a column still fits 64 bits): 105 + 196 MADs instead of 392.
template <class C> ZKT_HD Fp<C> fp_sqr_impl(const Fp<C>& a) {
  static_assert(C::W == 28, "lazy-limb fields only");
  constexpr int N = C::N; constexpr uint32_t M = (1u << 28) - 1;
  Fp<C> r; uint32_t m[N], d[N]; uint64_t acc = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) d[i] = a.v[i] << 1;
#pragma unroll
  for (int k = 0; k < 2 * N; ++k) {
#pragma unroll
    for (int i = (k < N ? 0 : k - N + 1); 2 * i < k; ++i) acc = mad64(d[i], a.v[k - i], acc);
    if ((k & 1) == 0) acc = mad64(a.v[k / 2], a.v[k / 2], acc);
    if (k < N) {
#pragma unroll
      for (int j = 0; j < k; ++j) acc = mad64(m[j], C::mod(k - j), acc);
      m[k] = ((uint32_t)acc * C::INV) & M;
      acc = mad64(m[k], C::mod(0), acc);
    } else {
#pragma unroll
      for (int j = k - N + 1; j < N; ++j) acc = mad64(m[j], C::mod(k - j), acc);
      r.v[k - N] = (uint32_t)acc & M;
    }
    acc >>= 28;
  }
  return r;
}

// a*b - c*d with ONE Montgomery reduction (y = m(x1 - x3) - y1 style terms): the subtrahend enters as c * (8p - d) with the
// borrow-safe spread of 8p, limbs < 2^29, so a column sums 14 products < 2^56, 14 < 2^57 and 14 reduction products < 2^56:
// below 2^62.  Output (ab + c(8p-d) + mp)/R < p (1 + 48 p/R) < 1.04p.
// SUB = false: a*b + c*d the same way (no spread needed).
template <class C, bool SUB> ZKT_HD Fp<C> fp_mul2_impl(const Fp<C>& a, const Fp<C>& b, const Fp<C>& c, const Fp<C>& d) {
  static_assert(C::W == 28, "lazy-limb fields only");
  constexpr int N = C::N; constexpr uint32_t M = (1u << 28) - 1;
  Fp<C> r; uint32_t m[N], nd[N]; uint64_t acc = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) nd[i] = SUB ? C::subk(i) - d.v[i] : d.v[i];
#pragma unroll
  for (int k = 0; k < 2 * N; ++k) {
#pragma unroll
    for (int i = (k < N ? 0 : k - N + 1); i <= (k < N ? k : N - 1); ++i) { acc = mad64(a.v[i], b.v[k - i], acc); acc = mad64(c.v[i], nd[k - i], acc); }
    if (k < N) {
#pragma unroll
      for (int j = 0; j < k; ++j) acc = mad64(m[j], C::mod(k - j), acc);
      m[k] = ((uint32_t)acc * C::INV) & M;
      acc = mad64(m[k], C::mod(0), acc);
    } else {
#pragma unroll
      for (int j = k - N + 1; j < N; ++j) acc = mad64(m[j], C::mod(k - j), acc);
      r.v[k - N] = (uint32_t)acc & M;
    }
    acc >>= 28;
  }
  return r;
}

// (a0 + a1 u)(b0 + b1 u), u^2 = -1, as THREE product scans under two Montgomery reductions (Karatsuba on the unreduced columns):
//   P0 = a0 b0,  P1' = a1 (8p - b1),  P2 = (a0 + a1)(b0 + b1)      re = P0 + P1',   im = P2 - P0 + P1'  (= a0 b1 + a1 b0 + 8p a1)
// 588 + 392 MADs instead of the 784 + 392 of two fp_mul2_impl; the price is ~250 cheap adds, which a single wave per SIMD issues in
// the shadow of its multiply-adds (profiles/r02_valu_ubench.txt: v_mad_u64_u32 holds a wave 9.5 cycles, the others ~3).
// Column-wise P2_k >= P0_k (every limb is non-negative), so no column ever goes negative.  Bounds for inputs < 4p with limbs < 2^28:
// sums < 2^29 per limb, P2_k < 14 * 2^58, P1'_k < 14 * 2^57, reduction < 14 * 2^56: a column stays below 2^62.7;
// re < 48 p^2, im < 64 p^2, so both outputs are < p (1 + 64 p / R) < 1.03 p.
template <class C> ZKT_HD void fp2_mul_kara(const Fp<C>& a0, const Fp<C>& a1, const Fp<C>& b0, const Fp<C>& b1, Fp<C>& re, Fp<C>& im) {
  static_assert(C::W == 28, "lazy-limb fields only");
  constexpr int N = C::N; constexpr uint32_t M = (1u << 28) - 1;
  uint32_t sa[N], sb[N], nd[N], mr[N], mi[N]; uint64_t accr = 0, acci = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) { sa[i] = a0.v[i] + a1.v[i]; sb[i] = b0.v[i] + b1.v[i]; nd[i] = C::subk(i) - b1.v[i]; }
#pragma unroll
  for (int k = 0; k < 2 * N; ++k) {
    uint64_t p0 = 0, p1 = 0;
#pragma unroll
    for (int i = (k < N ? 0 : k - N + 1); i <= (k < N ? k : N - 1); ++i) {
      p0 = mad64(a0.v[i], b0.v[k - i], p0); p1 = mad64(a1.v[i], nd[k - i], p1); acci = mad64(sa[i], sb[k - i], acci);
    }
    accr += p0 + p1; acci += p1 - p0;            // the difference may wrap; the sum with the P2 column already in acci is non-negative
    if (k < N) {
#pragma unroll
      for (int j = 0; j < k; ++j) { accr = mad64(mr[j], C::mod(k - j), accr); acci = mad64(mi[j], C::mod(k - j), acci); }
      mr[k] = ((uint32_t)accr * C::INV) & M; mi[k] = ((uint32_t)acci * C::INV) & M;
      accr = mad64(mr[k], C::mod(0), accr); acci = mad64(mi[k], C::mod(0), acci);
    } else {
#pragma unroll
      for (int j = k - N + 1; j < N; ++j) { accr = mad64(mr[j], C::mod(k - j), accr); acci = mad64(mi[j], C::mod(k - j), acci); }
      re.v[k - N] = (uint32_t)accr & M; im.v[k - N] = (uint32_t)acci & M;
    }
    accr >>= 28; acci >>= 28;
  }
}

// Call policy.  One inlined multiply is ~1000 instructions (8 KB); curve and
// pairing kernels contain hundreds of them, far beyond the 64 KB instruction
// cache, so by default the multiply is ONE function per field and kernel image,
// called with both operands in VGPRs (IPRA keeps the caller's live values out of
// the callee's clobber set, so nothing is spilled around the call).  Small
// kernels define ZKT_INLINE_MUL before including this header.
#if !defined(ZKT_INLINE_MUL)
template <class C> ZKT_FN Fp<C> fp_mul(Fp<C> a, Fp<C> b) { return fp_mul_impl(a, b); }
template <class C> ZKT_FN Fp<C> fp_sqr_fn(Fp<C> a) { return fp_sqr_impl(a); }
#if defined(ZKT_INLINE_MUL2)     // four 14-register operands exceed the 32 argument VGPRs of the calling convention: inline the two-product form only
template <class C> ZKT_HD Fp<C> fp_mulsub_fn(const Fp<C>& a, const Fp<C>& b, const Fp<C>& c, const Fp<C>& d) { return fp_mul2_impl<C, true>(a, b, c, d); }
template <class C> ZKT_HD Fp<C> fp_muladd_fn(const Fp<C>& a, const Fp<C>& b, const Fp<C>& c, const Fp<C>& d) { return fp_mul2_impl<C, false>(a, b, c, d); }
#else
template <class C> ZKT_FN Fp<C> fp_mulsub_fn(Fp<C> a, Fp<C> b, Fp<C> c, Fp<C> d) { return fp_mul2_impl<C, true>(a, b, c, d); }
template <class C> ZKT_FN Fp<C> fp_muladd_fn(Fp<C> a, Fp<C> b, Fp<C> c, Fp<C> d) { return fp_mul2_impl<C, false>(a, b, c, d); }
#endif
#else
template <class C> ZKT_HD Fp<C> fp_mul(const Fp<C>& a, const Fp<C>& b) { return fp_mul_impl(a, b); }
template <class C> ZKT_HD Fp<C> fp_sqr_fn(const Fp<C>& a) { return fp_sqr_impl(a); }
template <class C> ZKT_HD Fp<C> fp_mulsub_fn(const Fp<C>& a, const Fp<C>& b, const Fp<C>& c, const Fp<C>& d) { return fp_mul2_impl<C, true>(a, b, c, d); }
template <class C> ZKT_HD Fp<C> fp_muladd_fn(const Fp<C>& a, const Fp<C>& b, const Fp<C>& c, const Fp<C>& d) { return fp_mul2_impl<C, false>(a, b, c, d); }
#endif
template <class C> ZKT_HD Fp<C> fp_mulsub(const Fp<C>& a, const Fp<C>& b, const Fp<C>& c, const Fp<C>& d) {
  if constexpr (C::W == 28) return fp_mulsub_fn(a, b, c, d); else return fp_sub(fp_mul(a, b), fp_mul(c, d));
}
template <class C> ZKT_HD Fp<C> fp_muladd(const Fp<C>& a, const Fp<C>& b, const Fp<C>& c, const Fp<C>& d) {
  if constexpr (C::W == 28) return fp_muladd_fn(a, b, c, d); else return fp_add(fp_mul(a, b), fp_mul(c, d));
}

// Montgomery square.  sq (prime_field_elem.rs:330-335).  With 32-bit limbs a dedicated squaring
// pays ~6 shift/add ops per column for the 96-bit doubling and does not win, so it is the product;
// with 28-bit limbs the doubling is free (see fp_sqr_impl).
template <class C> ZKT_HD Fp<C> fp_sqr(const Fp<C>& a) {
  if constexpr (C::W == 28) return fp_sqr_fn(a); else return fp_mul(a, a);
}

// canonical words (ABI_N x 32 bit, < p) <-> internal Montgomery form
template <class C> ZKT_HD Fp<C> fp_from_words(const uint32_t* w) {
  Fp<C> x, r2;
#pragma unroll
  for (int i = 0; i < C::N; ++i) r2.v[i] = C::r2(i);
  if constexpr (C::W == 28) {
#pragma unroll
    for (int i = 0; i < C::N; ++i) {
      const int lo = (28 * i) >> 5, sh = (28 * i) & 31;
      uint32_t t = w[lo] >> sh;
      if (sh > 4 && lo + 1 < C::ABI_N) t |= w[lo + 1] << (32 - sh);
      x.v[i] = t & 0x0fffffffu;
    }
  } else {
#pragma unroll
    for (int i = 0; i < C::N; ++i) x.v[i] = w[i];
    x = fp_canon32(x);
  }
  return fp_mul(x, r2);
}
template <class C> ZKT_HD void fp_to_words(const Fp<C>& a, uint32_t* w) {
  Fp<C> one = fp_zero<C>(); one.v[0] = 1;
  Fp<C> x = fp_mul(a, one);
  if constexpr (C::W == 28) {
    // x < 1.01p: one conditional subtraction makes it canonical
    uint32_t s[C::N], bw = 0;
#pragma unroll
    for (int i = 0; i < C::N; ++i) { uint32_t t = x.v[i] - C::mod(i) - bw; bw = t >> 31; s[i] = t & 0x0fffffffu; }
#pragma unroll
    for (int i = 0; i < C::N; ++i) s[i] = bw ? x.v[i] : s[i];
#pragma unroll
    for (int j = 0; j < C::ABI_N; ++j) {
      uint32_t t = 0;
#pragma unroll
      for (int i = 0; i < C::N; ++i) {
        const int sft = 28 * i - 32 * j;
        if (sft > -28 && sft < 32) t |= sft >= 0 ? (s[i] << (sft & 31)) : (s[i] >> ((-sft) & 31));
      }
      w[j] = t;
    }
  } else {
#pragma unroll
    for (int i = 0; i < C::N; ++i) w[i] = x.v[i];
  }
}
// W = 32 fields keep canonical values: an ABI word vector may be any 256-bit integer (PrimeFieldElem::new reduces e mod order,
// prime_field_elem.rs:263-272).  2^256 < 3r and < 2p, 2n for the secp256k1 fields: two conditional subtractions reduce it.
// (W = 28: fp_from_words needs no such step — the Montgomery product by R^2 accepts any 384-bit integer and returns its residue.)
template <class C> ZKT_HD Fp<C> fp_canon32(Fp<C> x) {
  static_assert(C::W == 32, "canonical 32-bit-limb fields only");
  fp_cond_sub(x, 0); fp_cond_sub(x, 0);
  return x;
}

// Inverse by the binary extended Euclid on the plain integers (odd p): ~2*bits iterations of
// shifts and carry-chain adds instead of ~1.5*bits Montgomery products — about 4x cheaper than
// fp_inv_fermat on this machine, and it is the tail of every affine normalisation.
// u: non-zero residue < p as ABI_N 32-bit words; overwritten with u^-1 mod p.
template <class C> ZKT_HD void bgcd_inverse_classic(uint32_t* io) {
  constexpr int N = C::ABI_N;
  uint32_t u[N + 1], v[N + 1], x1[N + 1], x2[N + 1];
#pragma unroll
  for (int i = 0; i < N; ++i) { u[i] = io[i]; v[i] = C::mod32(i); x1[i] = 0; x2[i] = 0; }
  u[N] = v[N] = x1[N] = x2[N] = 0; x1[0] = 1;
  {   // zero has no inverse and would never leave the halving loop below: callers test for it, this is the backstop (result 0)
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) any |= u[i];
    if (any == 0) return;
  }
  auto is_one = [&](const uint32_t* t) { uint32_t o = t[0] ^ 1u;
#pragma unroll
    for (int i = 1; i <= N; ++i) o |= t[i]; return o == 0; };
  auto shr1 = [&](uint32_t* t) {
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = (t[i] >> 1) | (t[i + 1] << 31); t[N] >>= 1; };
  auto add_p = [&](uint32_t* t) { uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = addc(t[i], C::mod32(i), c); t[N] += c; };
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
#pragma unroll
  for (int i = 0; i < N; ++i) io[i] = use1 ? x1[i] : x2[i];
}

// The inverse the kernels use: the binary GCD with word-sized inner steps (Pornin, "Optimized Binary GCD for Modular Inversion", 2020).  The classic
// loop above spends ~100 instructions on EVERY one of its ~2 * bits steps in multi-word shifts, compares and subtractions, and it is the tail of every
// affine normalisation (0.25-0.3 ms of a 0.45 ms k_combine, measured, round 3).  Here 31 steps at a time run on 64-bit approximations of (a, b) — their
// top 33 and low 31 bits — and produce a 2 x 2 matrix of factors |f|, |g| <= 2^31, which is then applied once to the full-length a, b (exactly: the
// low 31 bits of a f0 + b g0 vanish by construction) and to the cofactors u, v, there with a Montgomery-style division by 2^31 modulo p:
//     a = u y,  b = v y  (mod p)  throughout;   a -> 0, b -> gcd = 1,  so  v = y^-1.
// 2 * bits - 1 inner steps reach the gcd; extra outer rounds leave (a, b, v) = (0, 1, y^-1) where it is.  Same residue as the classic loop and as
// the reference's extended Euclid (prime_field_elem.rs:384-446): an inverse is unique.  tests/test_hostcheck.py runs both against python's pow(x, -1, p);
// a final b != 1 (never observed) falls back to the classic loop.
#if !defined(__HIP_DEVICE_COMPILE__)
inline unsigned long& bgcd_fallbacks() { static unsigned long n = 0; return n; }      // host builds (tests/test_hostcheck.py): how often the backstop below ran
#endif
template <class C> ZKT_HD void bgcd_inverse(uint32_t* io) {
  constexpr int N = C::ABI_N;
  constexpr int ITER = (2 * 32 * N - 1 + 30) / 31 + 1;
  uint32_t a[N], b[N], u[N], v[N];
  uint32_t any = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) { a[i] = io[i]; b[i] = C::mod32(i); u[i] = 0; v[i] = 0; any |= io[i]; }
  if (any == 0) return;                                  // zero has no inverse: callers test for it, this is the backstop (result 0)
  u[0] = 1;
  uint32_t minv = C::mod32(0);                           // p^-1 mod 2^32 by Newton's iteration (p odd: p * p = 1 mod 8)
#pragma unroll
  for (int k = 0; k < 4; ++k) minv *= 2u - C::mod32(0) * minv;
  minv = (0u - minv) & 0x7fffffffu;                      // -p^-1 mod 2^31
  // (x f + y g) / 2^31 for x, y < 2^(32N) and |f| + |g| <= 2^31, streamed limb by limb in two's complement: every partial sum
  // carry + x_i f + y_i g lies in [-2^63, 2^63), so one signed 64-bit accumulator carries it.  The low 31 bits of the sum are zero by construction
  // of (f, g); |result| < 2^(32N).  Writes the magnitude, returns the sign.  (No N-word temporaries: with them the function took 270 registers and,
  // called three levels deep from k_tower_op<12>, trampled its callers' live registers on this compiler.)
  auto comb_shift = [&](const uint32_t* x, const uint32_t* y, int64_t f, int64_t g, uint32_t* out) -> bool {
    const uint64_t uf = (uint64_t)f, ug = (uint64_t)g;
    uint64_t acc = (uint64_t)x[0] * uf + (uint64_t)y[0] * ug;
    uint32_t prev = (uint32_t)acc; int64_t carry = (int64_t)acc >> 32;
#pragma unroll
    for (int i = 1; i < N; ++i) {
      acc = (uint64_t)carry + (uint64_t)x[i] * uf + (uint64_t)y[i] * ug;
      const uint32_t w = (uint32_t)acc; carry = (int64_t)acc >> 32;
      out[i - 1] = (prev >> 31) | (w << 1); prev = w;
    }
    out[N - 1] = (prev >> 31) | ((uint32_t)carry << 1);
    const bool neg = carry < 0;
    if (neg) { uint32_t c = 1;
#pragma unroll
      for (int i = 0; i < N; ++i) out[i] = addc(~out[i], 0u, c); }
    return neg;
  };
  // (x f + y g) / 2^31 mod p for x, y < p: q p is added so that the low 31 bits vanish (q = -(x f + y g) / p mod 2^31), in a second carry chain
  // beside the signed one; the quotient lies in (-p, 2p) and is brought into [0, p); `flip` negates the result (the row's sign was turned).
  auto comb_mont = [&](const uint32_t* x, const uint32_t* y, int64_t f, int64_t g, bool flip, uint32_t* out) {
    const uint64_t uf = (uint64_t)f, ug = (uint64_t)g;
    uint64_t a1 = (uint64_t)x[0] * uf + (uint64_t)y[0] * ug;
    const uint32_t q = ((uint32_t)a1 * minv) & 0x7fffffffu;
    uint64_t a2 = (uint64_t)q * C::mod32(0);
    uint64_t w = (uint64_t)(uint32_t)a1 + (uint32_t)a2;                // low word of the sum: its low 31 bits are zero
    uint32_t prev = (uint32_t)w; uint64_t cw = w >> 32;
    int64_t c1 = (int64_t)a1 >> 32; uint64_t c2 = a2 >> 32;
    uint32_t r[N];
#pragma unroll
    for (int i = 1; i < N; ++i) {
      a1 = (uint64_t)c1 + (uint64_t)x[i] * uf + (uint64_t)y[i] * ug; c1 = (int64_t)a1 >> 32;
      a2 = c2 + (uint64_t)q * C::mod32(i); c2 = a2 >> 32;
      w = (uint64_t)(uint32_t)a1 + (uint32_t)a2 + cw; cw = w >> 32;
      r[i - 1] = (prev >> 31) | ((uint32_t)w << 1); prev = (uint32_t)w;
    }
    const int64_t top = c1 + (int64_t)c2 + (int64_t)cw;               // the words above 2^(32N): |top| < 2^32
    r[N - 1] = (prev >> 31) | ((uint32_t)top << 1);
    const int64_t hi = top >> 31;                                       // the quotient is hi * 2^(32N) + r, hi in {-1, 0, 1}
    uint32_t s2[N], bw = 0, c = 0;
    if (hi < 0) {                                                       // negative: + p
#pragma unroll
      for (int i = 0; i < N; ++i) r[i] = addc(r[i], C::mod32(i), c);
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) s2[i] = subb(r[i], C::mod32(i), bw);
      const bool ge = hi > 0 || !bw;
#pragma unroll
      for (int i = 0; i < N; ++i) r[i] = ge ? s2[i] : r[i];
    }
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) nz |= r[i];
    if (flip && nz) { bw = 0;
#pragma unroll
      for (int i = 0; i < N; ++i) r[i] = subb(C::mod32(i), r[i], bw); }
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = r[i];
  };
#pragma unroll 1
  for (int it = 0; it < ITER; ++it) {
    // the two top words of max(a, b) (same position for both) without indexing registers dynamically
    uint32_t ah = 0, al = 0, bh = 0, bl = 0; bool found = false, pair10 = false;
#pragma unroll
    for (int j = N - 1; j >= 1; --j) {
      const bool hit = !found && ((a[j] | b[j]) != 0);
      ah = hit ? a[j] : ah; al = hit ? a[j - 1] : al; bh = hit ? b[j] : bh; bl = hit ? b[j - 1] : bl;
      pair10 = hit ? (j == 1) : pair10; found = found || hit;
    }
    uint64_t xa, xb;
    if (!found || pair10) { xa = ((uint64_t)a[1] << 32) | a[0]; xb = ((uint64_t)b[1] << 32) | b[0]; }      // at most 64 bits: exact
    else {                                                               // low 31 bits (what 31 parity decisions read) + the top 33 bits
      const int s = __builtin_clz(ah | bh);
      const uint64_t ta = ((((uint64_t)ah << 32) | al) << s) >> 31, tb = ((((uint64_t)bh << 32) | bl) << s) >> 31;
      xa = (ta << 31) | (a[0] & 0x7fffffffu); xb = (tb << 31) | (b[0] & 0x7fffffffu);
    }
    int64_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
#pragma unroll 1
    for (int k = 0; k < 31; ++k) {
      const uint64_t odd = 0ull - (xa & 1ull);
      const uint64_t sw = odd & (xa < xb ? ~0ull : 0ull);
      const uint64_t tx = (xa ^ xb) & sw; xa ^= tx; xb ^= tx;
      const int64_t tf = (f0 ^ f1) & (int64_t)sw; f0 ^= tf; f1 ^= tf;
      const int64_t tg = (g0 ^ g1) & (int64_t)sw; g0 ^= tg; g1 ^= tg;
      xa -= xb & odd; f0 -= f1 & (int64_t)odd; g0 -= g1 & (int64_t)odd;
      xa >>= 1; f1 <<= 1; g1 <<= 1;
    }
    uint32_t na[N], nb[N], nu[N], nv[N];
    const bool sa = comb_shift(a, b, f0, g0, na), sb = comb_shift(a, b, f1, g1, nb);
    comb_mont(u, v, f0, g0, sa, nu); comb_mont(u, v, f1, g1, sb, nv);                 // cofactors: same rows, negated where a row's result was
#pragma unroll
    for (int i = 0; i < N; ++i) { a[i] = na[i]; b[i] = nb[i]; u[i] = nu[i]; v[i] = nv[i]; }
  }
  uint32_t rest = b[0] ^ 1u;
#pragma unroll
  for (int i = 1; i < N; ++i) rest |= b[i];
  if (rest != 0) {
#if !defined(__HIP_DEVICE_COMPILE__)
    ++bgcd_fallbacks();
#endif
    bgcd_inverse_classic<C>(io); return;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) io[i] = v[i];
}

// In: a*R (Montgomery), non-zero.  Out: a^-1*R.
// W = 32: the integer inverse of a*R is a^-1*R^-1; one Montgomery product with R^3 lifts it back.
// W = 28: leave the Montgomery domain (canonical words), invert, re-enter.
// Which loop: the word-step GCD for the 8-word fields everywhere, and for Fq in the objects built with -DZKT_WORDSTEP_INV_FQ (the MSM objects, where
// fp_inv is called straight from a kernel).  In the tower / pairing objects Fq keeps the classic loop ON PURPOSE: there fp_inv sits below functions that
// realign their stack (fq6_inv, fq12_inv, the final exponentiation: 64-byte aligned Fq6 temporaries) and keep their incoming stack pointer in s34, the
// base-pointer register — and this compiler's inter-procedural register allocation does not keep s34 out of the hands of their callees.  The word-step
// loop writes s0..s91; fq2_inv, which holds a modulus limb in an SGPR across its call to fp_inv, was thereby pushed to s34, fq6_inv returned with a
// trampled base pointer and the next callee faulted on its first stack store (rocgdb, round 3).  The classic loop stays below s34.
template <class C> ZKT_HD void fp_inverse_words(uint32_t* w) {
#if defined(ZKT_WORDSTEP_INV_FQ)
  bgcd_inverse<C>(w);
#else
  if constexpr (C::W == 28) bgcd_inverse_classic<C>(w); else bgcd_inverse<C>(w);
#endif
}
template <class C> ZKT_FN Fp<C> fp_inv(Fp<C> a) {
  uint32_t w[C::ABI_N];
  if constexpr (C::W == 28) {
    fp_to_words(a, w);
    fp_inverse_words<C>(w);
    return fp_from_words<C>(w);
  } else {
    Fp<C> r, r3;
#pragma unroll
    for (int i = 0; i < C::N; ++i) w[i] = a.v[i];
    fp_inverse_words<C>(w);
#pragma unroll
    for (int i = 0; i < C::N; ++i) { r.v[i] = w[i]; r3.v[i] = C::r3(i); }
    return fp_mul(r, r3);
  }
}

// generic power with a run-time exponent of `nlimbs` 32-bit limbs (MSB-first square-and-multiply; pow, prime_field_elem.rs:311-328,
// runs LSB-first on the same bits and computes the same residue; exponent 0 gives 1 for every base, 0 included, as there)
template <class C> ZKT_FN Fp<C> fp_pow(Fp<C> a, const uint32_t* e, int nlimbs) {
  Fp<C> r = fp_one<C>();
  int top = nlimbs * 32 - 1;
  while (top >= 0 && !((e[top >> 5] >> (top & 31)) & 1)) --top;      // squaring 1 through the leading zero bits changes nothing
  for (int i = top; i >= 0; --i) {
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
