// Pair-tree rounds in AFFINE coordinates ahead of the G2 bucket accumulation (round 4; profiles/r04_batched_affine_go_no_go.md).
//   Polynomial::eval_with_g2_hidings   src/building_block/field/polynomial.rs:283-293
//   impl_affine_add!                   src/building_block/curves/macros.rs:34-163   (the reference's own addition: one inversion per add)
// The sort stage (zkt_msm.hip) leaves the entries of every bucket side by side.  Round r sums the points of a bucket in pairs — 26 -> 13 -> 7 … —
// with the reference's affine formulas, and shares ONE inversion among the K pair additions a lane owns (Montgomery's trick: forward pass = prefix
// products of the denominators, one word-step inversion of their product, backward pass = slopes and sums).  An affine addition over Fq2 is
// 5M + 1S + inversion/K against the 8M + 2S of the XYZZ mixed addition, and — what decides it on this machine — its working set is 246 registers
// where the XYZZ accumulator needs 378-479: two waves per SIMD, no spills.  Measured per addition (tools/ubench/affine_round.hip): 465-525 ps against
// 635-676 ps.  Over Fq the same form LOSES (267 against 181 ps: the inversion's share is four times larger and the traffic per multiply-add twice):
// G1 keeps the XYZZ chain, and this file is instantiated for G2 only.
// After R rounds a bucket holds ceil(cnt / 2^R) points; the XYZZ kernel (zkt_msm_g2pair.hip, DIRECT mode) accumulates those and everything behind it
// is unchanged.  The group element is the same whatever the summation order, and the result leaves as a canonical affine point: bit-compatible.
//
// Layout.  Layer 0 = the sorted entries (index into the window-multiple table | sign), dense: bucket b at offsets[b].  Layer r >= 1 = affine points
// (2 coordinates, raw Montgomery words) + one infinity byte per slot, bucket b at (offsets[b] >> r) + b — a closed form instead of a scan per round:
// floor(x + y) >= floor(x) + floor(y) makes the ranges disjoint, at the price of at most one unused slot per bucket and layer.
// A lane owns K consecutive OUTPUT slots of its round; slot (b, j) adds inputs 2j and 2j+1 of bucket b (the odd one out is copied through).
// Exceptional cases as the reference orders them (macros.rs:43-108): an operand at infinity, P + (-P) = infinity, P + P by the tangent (y = 0: infinity).
#include "abi.h"
#include "zkt_internal.h"

namespace zkt {
namespace {

template <class F> struct ACoord;
template <> struct ACoord<Fq2Ops> {
  static constexpr int CW = 2 * FqC::N;
  __device__ static Fq2 ld(const uint32_t* p) { Fq2 r; r.c0 = ld_raw<FqC>(p); r.c1 = ld_raw<FqC>(p + FqC::N); return r; }
  __device__ static void st(uint32_t* p, const Fq2& a) { st_raw<FqC>(p, a.c0); st_raw<FqC>(p + FqC::N, a.c1); }
};
template <> struct ACoord<FqOps> {
  static constexpr int CW = FqC::N;
  __device__ static Fq ld(const uint32_t* p) { return ld_raw<FqC>(p); }
  __device__ static void st(uint32_t* p, const Fq& a) { st_raw<FqC>(p, a); }
};

// prefix products, wave-interleaved: quad q of step i of block blk at ((blk * K + i) * Q + q) * 256 + lane * 4 words (16-byte accesses, coalesced)
template <class F> __device__ inline void st_pref(uint32_t* base, const typename F::E& v) {
  constexpr int CW = ACoord<F>::CW, Q = (CW + 3) / 4;
  uint32_t w[Q * 4] = {};
  ACoord<F>::st(w, v);
#pragma unroll
  for (int q = 0; q < Q; ++q) *reinterpret_cast<uint4*>(base + q * 256) = uint4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
}
template <class F> __device__ inline typename F::E ld_pref(const uint32_t* base) {
  constexpr int CW = ACoord<F>::CW, Q = (CW + 3) / 4;
  uint32_t w[Q * 4];
#pragma unroll
  for (int q = 0; q < Q; ++q) { const uint4 t = *reinterpret_cast<const uint4*>(base + q * 256); w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w; }
  return ACoord<F>::ld(w);
}

struct AffRound {
  const uint32_t* offsets;      // dense offsets of layer 0 (B + 1 words)
  uint32_t B;                   // buckets
  int r;                        // this round's OUTPUT layer (>= 1); the input layer is r - 1
  const uint32_t* table;        // layer 0: window-multiple table
  const uint32_t* entries;      // layer 0: sorted entries
  const uint32_t* pts_in;       // layer r - 1 >= 1: points / infinity bytes
  const uint8_t* inf_in;
  uint32_t* pts_out; uint8_t* inf_out;      // layer r
  uint32_t* pref;               // prefix products of this launch
};
__device__ inline uint32_t lay_off(const uint32_t* offsets, uint32_t b, int r) { return r == 0 ? offsets[b] : (offsets[b] >> r) + b; }
__device__ inline uint32_t lay_cnt(uint32_t c0, int r) { return (c0 + ((1u << r) - 1u)) >> r; }

// the bucket a slot of layer r lies in, walked forward or backward from the previous slot's
struct Cursor {
  uint32_t b, base0, next0;     // bucket, offsets[b], offsets[b + 1]
  __device__ void seek(const AffRound& a, uint32_t o) {          // largest b with lay_off(b) <= o
    uint32_t lo = 0, hi = a.B;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (lay_off(a.offsets, mid, a.r) <= o) lo = mid; else hi = mid; }
    b = lo; base0 = a.offsets[b]; next0 = a.offsets[b + 1];
  }
  __device__ void forward(const AffRound& a, uint32_t o) { while (b + 1 < a.B && ((next0 >> a.r) + b + 1) <= o) { ++b; base0 = next0; next0 = a.offsets[b + 1]; } }
  __device__ void backward(const AffRound& a, uint32_t o) { while (b > 0 && ((base0 >> a.r) + b) > o) { --b; next0 = base0; base0 = a.offsets[b]; } }
};
// what slot o of the output layer has to do: its two input slots (i1 = NONE: copy input 0 through), or nothing (a gap)
struct SlotJob { bool valid, has1; uint32_t i0; };
__device__ inline SlotJob slot_job(const AffRound& a, const Cursor& c, uint32_t o) {
  SlotJob s;
  const uint32_t c0 = c.next0 - c.base0, j = o - ((c.base0 >> a.r) + c.b);
  s.valid = j < lay_cnt(c0, a.r);
  const uint32_t cin = lay_cnt(c0, a.r - 1);
  s.i0 = (a.r == 1 ? c.base0 : (c.base0 >> (a.r - 1)) + c.b) + 2 * j;
  s.has1 = 2 * j + 1 < cin;
  return s;
}
template <class F> struct InPt { typename F::E x, y; bool inf; };
template <class F, bool WITH_Y> __device__ inline InPt<F> load_in(const AffRound& a, uint32_t idx) {
  constexpr int CW = ACoord<F>::CW, PW = 2 * CW;
  InPt<F> p;
  if (a.r == 1) {
    const uint32_t ent = a.entries[idx];
    const uint32_t* q = a.table + (size_t)(ent & 0x7fffffffu) * PW;
    p.x = ACoord<F>::ld(q); p.inf = false;
    if (WITH_Y) { p.y = ACoord<F>::ld(q + CW); if (ent >> 31) p.y = F::neg(p.y); }
  } else {
    const uint32_t* q = a.pts_in + (size_t)idx * PW;
    p.inf = a.inf_in[idx] != 0;
    p.x = ACoord<F>::ld(q);
    if (WITH_Y) p.y = ACoord<F>::ld(q + CW);
  }
  return p;
}

template <class F, int K>
__global__ void __launch_bounds__(64) k_affine_round(AffRound a) {
  typedef typename F::E E;
  constexpr int CW = ACoord<F>::CW, PW = 2 * CW, Q = (CW + 3) / 4;
  const uint32_t total = (a.offsets[a.B] >> a.r) + a.B;               // slots of the output layer
  const uint32_t o0 = ((uint32_t)blockIdx.x * 64 + threadIdx.x) * K;
  if (o0 >= total) return;
  const uint32_t o1 = o0 + K < total ? o0 + K : total;
  uint32_t* myp = a.pref + (size_t)blockIdx.x * K * Q * 256 + threadIdx.x * 4;
  Cursor c; c.seek(a, o0);
  E acc = F::one();
  bool any = false;
  // forward: denominators x1 - x0 (2 y for a tangent), their running product; the prefix BEFORE each factor goes to memory
  for (uint32_t o = o0; o < o1; ++o) {
    c.forward(a, o);
    const SlotJob s = slot_job(a, c, o);
    if (!s.valid || !s.has1) continue;
    const InPt<F> p0 = load_in<F, false>(a, s.i0), p1 = load_in<F, false>(a, s.i0 + 1);
    if (p0.inf || p1.inf) continue;
    E d = F::sub(p1.x, p0.x);
    if (F::is_zero(d)) {                                              // same abscissa: tangent or vertical (rare: a repeated base, or a point and its negative)
      const InPt<F> q0 = load_in<F, true>(a, s.i0), q1 = load_in<F, true>(a, s.i0 + 1);
      if (!F::eq(q0.y, q1.y) || F::is_zero(q0.y)) continue;           // P + (-P), or 2P with y = 0: infinity, no denominator
      d = F::dbl(q0.y);
    }
    st_pref<F>(myp + (size_t)(o - o0) * Q * 256, acc);
    acc = F::mul(acc, d); any = true;
  }
  E inv = any ? F::inv(acc) : F::one();
  // backward: 1/d_i = inv * prefix_i, inv *= d_i; slope, sum, store
  for (uint32_t o = o1; o-- > o0;) {
    c.backward(a, o);
    const SlotJob s = slot_job(a, c, o);
    if (!s.valid) continue;
    uint32_t* out = a.pts_out + (size_t)o * PW;
    const InPt<F> p0 = load_in<F, true>(a, s.i0);
    if (!s.has1) { ACoord<F>::st(out, p0.x); ACoord<F>::st(out + CW, p0.y); a.inf_out[o] = p0.inf ? 1 : 0; continue; }
    const InPt<F> p1 = load_in<F, true>(a, s.i0 + 1);
    if (p0.inf || p1.inf) {                                           // inf + P = P (macros.rs:43-49)
      const InPt<F>& keep = p0.inf ? p1 : p0;
      ACoord<F>::st(out, keep.x); ACoord<F>::st(out + CW, keep.y); a.inf_out[o] = (p0.inf && p1.inf) ? 1 : 0; continue;
    }
    E d = F::sub(p1.x, p0.x), num;
    if (F::is_zero(d)) {
      if (!F::eq(p0.y, p1.y) || F::is_zero(p0.y)) { a.inf_out[o] = 1; continue; }      // macros.rs:52-63
      d = F::dbl(p0.y);
      const E xx = F::sqr(p0.x); num = F::add(F::dbl(xx), xx);          // m = 3 x^2 / (2 y)   (macros.rs:65-70)
    } else num = F::sub(p1.y, p0.y);                                    // m = (y2 - y1) / (x2 - x1)   (macros.rs:88-92)
    const E pp = ld_pref<F>(myp + (size_t)(o - o0) * Q * 256);
    const E dinv = F::mul(inv, pp);
    inv = F::mul(inv, d);
    const E lam = F::mul(num, dinv);
    const E x3 = F::sub(F::sub(F::sqr(lam), p0.x), p1.x);               // x3 = m^2 - x1 - x2
    const E y3 = F::sub(F::mul(lam, F::sub(p0.x, x3)), p0.y);           // y3 = m (x1 - x3) - y1
    ACoord<F>::st(out, x3); ACoord<F>::st(out + CW, y3); a.inf_out[o] = 0;
  }
}

// counts and gapped offsets of the final layer for the task list and the DIRECT accumulate: cntR[b] = ceil(cnt[b] / 2^R), offR[b] = (offsets[b] >> R) + b
__global__ void __launch_bounds__(256) k_affine_final_layer(const uint32_t* __restrict__ offsets, uint32_t B, int R, uint32_t* __restrict__ cntR, uint32_t* __restrict__ offR) {
  const uint32_t b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const uint32_t o = offsets[b], c0 = offsets[b + 1] - o;
  cntR[b] = lay_cnt(c0, R); offR[b] = (o >> R) + b;
}
}  // namespace

static constexpr int AFF_K = 32;
size_t msm_affine_ws_bytes(size_t entries, size_t nbuckets, int rounds, int coord_words) {
  if (rounds <= 0) return 0;
  const size_t PW = 2 * (size_t)coord_words * 4, Q = ((size_t)coord_words + 3) / 4;
  size_t b = 2 * (nbuckets + 1) * 4 + 512;
  for (int r = 1; r <= rounds; ++r) { const size_t S = (entries >> r) + nbuckets + 64; b += S * PW + S + 512; }
  const size_t S1 = (entries >> 1) + nbuckets, blocks = (S1 + 64 * AFF_K - 1) / (64 * AFF_K);
  b += blocks * AFF_K * Q * 256 * 4 + 512;
  return b;
}
MsmAffineWs msm_affine_carve(void* base, size_t entries, size_t nbuckets, int rounds, int coord_words) {
  MsmAffineWs w{};
  const size_t PW = 2 * (size_t)coord_words * 4, Q = ((size_t)coord_words + 3) / 4;
  uint8_t* p = (uint8_t*)(((uintptr_t)base + 255) & ~(uintptr_t)255);
  w.cntR = (uint32_t*)p; p += (nbuckets + 1) * 4;
  w.offR = (uint32_t*)p; p += (nbuckets + 1) * 4;
  for (int r = 1; r <= rounds; ++r) {
    const size_t S = (entries >> r) + nbuckets + 64;
    p = (uint8_t*)(((uintptr_t)p + 255) & ~(uintptr_t)255);
    w.pts[r] = (uint32_t*)p; p += S * PW;
    w.inf[r] = p; p += S;
  }
  p = (uint8_t*)(((uintptr_t)p + 255) & ~(uintptr_t)255);
  w.pref = (uint32_t*)p;
  (void)Q;
  return w;
}
hipError_t launch_msm_affine_final_layer(const uint32_t* offsets, size_t nbuckets, int rounds, const MsmAffineWs& w, hipStream_t s) {
  hipLaunchKernelGGL(k_affine_final_layer, dim3((unsigned)((nbuckets + 255) / 256)), dim3(256), 0, s, offsets, (uint32_t)nbuckets, rounds, w.cntR, w.offR);
  return hipGetLastError();
}
// rounds 1..R for G2; `entries_bound` = an upper bound of the sorted entries (nwin * n): the grid is sized for it, lanes beyond the real count leave at once
hipError_t launch_msm_affine_rounds_g2(const uint32_t* table, const uint32_t* entries, const uint32_t* offsets, size_t nbuckets, size_t entries_bound, int rounds,
                                       const MsmAffineWs& w, hipStream_t s) {
  for (int r = 1; r <= rounds; ++r) {
    AffRound a{};
    a.offsets = offsets; a.B = (uint32_t)nbuckets; a.r = r; a.table = table; a.entries = entries;
    a.pts_in = r > 1 ? w.pts[r - 1] : nullptr; a.inf_in = r > 1 ? w.inf[r - 1] : nullptr;
    a.pts_out = w.pts[r]; a.inf_out = w.inf[r]; a.pref = w.pref;
    const size_t S = (entries_bound >> r) + nbuckets, blocks = (S + 64 * AFF_K - 1) / (64 * AFF_K);
    hipLaunchKernelGGL((k_affine_round<Fq2Ops, AFF_K>), dim3((unsigned)blocks), dim3(64), 0, s, a);
  }
  return hipGetLastError();
}

}  // namespace zkt
