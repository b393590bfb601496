// Lane-distributed Tate pairing for SMALL batches: one pairing per 12 lanes (rows a10-a13 of SURVEY §8, Pairing::tate pairing.rs:86-100).
//
// k_tate (zkt_tate.hip) runs one pairing per lane: 23 M instructions in sequence, ~66 ms however few pairings there are — a single
// Groth16 verification (verifier.rs:30-54) took 107 ms.  Here an Fq12 value is spread over a GROUP of 12 lanes in the basis
//      sum_{m<6} (x_m + y_m u) w^m,      w^6 = xi = 1 + u,  u^2 = -1        (w^2 = v: m = 2k + i is slot v^k w^i of the tower)
// lane (m, part) holding ONE Fq coefficient (part 0: x_m, part 1: y_m).  Every Fq12 product is then 12 independent output coefficients,
// each a sum of 12 Fq products that is accumulated UNREDUCED in 28 64-bit columns (168 products < 2^56 per column stay below 2^64) and
// reduced ONCE (Montgomery, 196 MADs): 2,548 v_mad_u64_u32 per lane per product, no Karatsuba recombination passes, no scratch.  The
// xi-twist of the wrapped terms is folded into precomputed s = x - y, t = x + y of the second operand, so all lanes do the same work:
//      Re c_m = sum_{j+k=m} (x_j x'_k - y_j y'_k) + sum_{j+k=m+6} (x_j s'_k - y_j t'_k)
//      Im c_m = sum_{j+k=m} (x_j y'_k + y_j x'_k) + sum_{j+k=m+6} (x_j t'_k + y_j s'_k)
// Operands are exchanged through LDS ("images": X, Y, -Y, S, T of the six coefficients).  The G1 point arithmetic of the Miller loop
// (Fq only) is spread over the lanes as well: each level of its dependency graph is one multiplication per lane on LDS slots.
// The algorithm is that of pairing.h (signed-digit Miller loop over r-1, sparse lines, exact final exponentiation): the same field
// elements, so the same bits as k_tate and as the reference.  rP != infinity is detected as there and handed to k_tate_exact_marked.
#include "abi.h"
#include "zkt_internal.h"

namespace zkt {
namespace dp {

constexpr int GL = 12;            // lanes per pairing
constexpr int GPW = 5;            // pairings per wave (60 of 64 lanes)
constexpr int SW = 16;            // dwords per LDS slot (14 limbs + 2 pad: 64 B, b128-aligned)
constexpr uint32_t M28 = (1u << 28) - 1;
// image = expanded operand: slots X[6] Y[6] NY[6] S[6] T[6]
constexpr int IX = 0, IY = 6, INY = 12, IS = 18, IT = 24, IMG_SLOTS = 30;
constexpr int NIMG = 3;
constexpr int PT_SLOTS = 52;
constexpr int GROUP_SLOTS = NIMG * IMG_SLOTS + PT_SLOTS;
constexpr int GROUP_WORDS = GROUP_SLOTS * SW;

struct Role { int g, m, part; };

__device__ inline void lst(uint32_t* s, const Fq& a) {
  uint4* p = reinterpret_cast<uint4*>(s);
  p[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]); p[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
  p[2] = make_uint4(a.v[8], a.v[9], a.v[10], a.v[11]); reinterpret_cast<uint2*>(s + 12)[0] = make_uint2(a.v[12], a.v[13]);
}
__device__ inline Fq lld(const uint32_t* s) {
  const uint4* p = reinterpret_cast<const uint4*>(s);
  uint4 a = p[0], b = p[1], c = p[2]; uint2 d = reinterpret_cast<const uint2*>(s + 12)[0];
  Fq r;
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  r.v[8] = c.x; r.v[9] = c.y; r.v[10] = c.z; r.v[11] = c.w; r.v[12] = d.x; r.v[13] = d.y;
  return r;
}
__device__ inline Fq fsel(bool c, const Fq& a, const Fq& b) { Fq r;
#pragma unroll
  for (int i = 0; i < 14; ++i) r.v[i] = c ? a.v[i] : b.v[i];
  return r; }
__device__ inline void gsync() { __syncthreads(); }       // one wave per block: orders the LDS traffic of the group

// ---- unreduced accumulation of Fq products in 28 columns, one Montgomery reduction -------------------------------------------------
struct Cols { uint64_t c[28]; };
__device__ inline void cols_zero(Cols& k) {
#pragma unroll
  for (int i = 0; i < 28; ++i) k.c[i] = 0; }
__device__ inline void cols_mac(Cols& k, const Fq& a, const Fq& b) {       // 196 MADs; limbs < 2^28 (or one operand < 2^29, see callers)
#pragma unroll
  for (int i = 0; i < 14; ++i)
#pragma unroll
    for (int j = 0; j < 14; ++j) k.c[i + j] = mad64(a.v[i], b.v[j], k.c[i + j]);
}
// (sum of products + m p) / 2^392: < 1.1 p for <= 12 products of values < 4p
__device__ inline Fq cols_reduce(Cols& k) {
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    const uint32_t m = ((uint32_t)k.c[i] * FqC::INV) & M28;
#pragma unroll
    for (int j = 0; j < 14; ++j) k.c[i + j] = mad64(m, FqC::mod(j), k.c[i + j]);
    k.c[i + 1] += k.c[i] >> 28;
  }
  Fq r; uint64_t carry = 0;
#pragma unroll
  for (int i = 0; i < 14; ++i) { const uint64_t t = k.c[14 + i] + carry; r.v[i] = (uint32_t)t & M28; carry = t >> 28; }
  return r;
}

// ---- images ---------------------------------------------------------------------------------------------------------------------------
// write the group's value (one coefficient per lane) as an expanded image: X, Y, -Y now; S = x - y, T = x + y after the exchange
__device__ inline void expand(uint32_t* img, const Fq& me, const Role& r, uint32_t* dummy) {
  const Fq neg = fp_neg(me);
  lst(img + ((r.part ? IY : IX) + r.m) * SW, me);
  lst(r.part ? img + (INY + r.m) * SW : dummy, neg);
  gsync();
  const Fq other = lld(img + ((r.part ? IX : IY) + r.m) * SW);           // the partner's half of coefficient m
  // part 0 (x): S = x - y;  part 1 (y): T = x + y
  uint32_t v[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) v[i] = me.v[i] + (r.part ? other.v[i] : FqC::subk(i) - other.v[i]);
  lst(img + ((r.part ? IT : IS) + r.m) * SW, fp_lazy_reduce<FqC>(v));
  gsync();
}
// only X, Y, -Y (the first operand of a product needs no S, T)
__device__ inline void expand_first(uint32_t* img, const Fq& me, const Role& r, uint32_t* dummy) {
  const Fq neg = fp_neg(me);
  lst(img + ((r.part ? IY : IX) + r.m) * SW, me);
  lst(r.part ? img + (INY + r.m) * SW : dummy, neg);
  gsync();
}

// my coefficient of A * B (both images complete)
__device__ __attribute__((noinline)) Fq dot_mul(const uint32_t* A, const uint32_t* B, int m, int part) {
  Cols k; cols_zero(k);
#pragma unroll 1
  for (int j = 0; j < 6; ++j) {
    int kk = m - j; const bool wrap = kk < 0; if (wrap) kk += 6;
    // Re: x_j * (x'|s') + (-y_j) * (y'|t')        Im: x_j * (y'|t') + y_j * (x'|s')
    const Fq p1 = lld(A + (IX + j) * SW);
    const Fq u1 = lld(B + ((part ? (wrap ? IT : IY) : (wrap ? IS : IX)) + kk) * SW);
    cols_mac(k, p1, u1);
    const Fq p2 = lld(A + ((part ? IY : INY) + j) * SW);
    const Fq u2 = lld(B + ((part ? (wrap ? IS : IX) : (wrap ? IT : IY)) + kk) * SW);
    cols_mac(k, p2, u2);
  }
  return cols_reduce(k);
}
// my coefficient of A * A: the sums above are symmetric in (j, k), so every unordered pair is evaluated once and counted twice through a
// doubled first operand (limbs < 2^29: 8 scans of 14 products < 2^57 stay below 2^64) — at most 4 pairs per lane instead of 6: 8 scans, not 12
__device__ __attribute__((noinline)) Fq dot_sqr(const uint32_t* A, int m, int part) {
  Cols k; cols_zero(k);
  const int nA = m / 2 + 1;                                  // pairs j <= k with j + k = m; then pairs with j + k = m + 6
#pragma unroll 1
  for (int it = 0; it < 4; ++it) {
    const bool wrap = it >= nA;
    const int j = wrap ? m + 1 + (it - nA) : it, kk = (wrap ? m + 6 : m) - j;
    const bool valid = j <= kk && j < 6;
    const int jj = valid ? j : 0, kq = valid ? kk : 0;       // harmless in-range loads for the idle iterations
    const uint32_t sh = (valid && j < kk) ? 1u : 0u;
    Fq p1 = lld(A + (IX + jj) * SW), p2 = lld(A + ((part ? IY : INY) + jj) * SW);
#pragma unroll
    for (int i = 0; i < 14; ++i) { p1.v[i] = valid ? p1.v[i] << sh : 0u; p2.v[i] = valid ? p2.v[i] << sh : 0u; }
    cols_mac(k, p1, lld(A + ((part ? (wrap ? IT : IY) : (wrap ? IS : IX)) + kq) * SW));
    cols_mac(k, p2, lld(A + ((part ? (wrap ? IS : IX) : (wrap ? IT : IY)) + kq) * SW));
  }
  return cols_reduce(k);
}
// Granger-Scott squaring in the cyclotomic subgroup (fq12_cyclotomic_sqr, tower.h) on the distributed form.  With the pairs (a, b) =
// (w^0,w^3), (w^1,w^4), (w^2,w^5):  t = a^2 + xi b^2 gives the new w^0, w^2, w^4 as 3t - 2 old;  t = 2ab (xi 2ab for the last pair) the new
// w^3, w^5 (w^1) as 3t + 2 old.  At most four products per lane (980 MADs) instead of eight.
struct CycRow { uint8_t fa[4], ia[4], fb[4], ib[4], dbl[4], n; };
#define CY_C0(a, b) {{IX, INY, IX, INY}, {a, a, b, b}, {IX, IY, IS, IT}, {a, a, b, b}, {0, 0, 0, 0}, 4}, \
                    {{IX, IX, IY, IX}, {a, b, b, 0}, {IY, IT, IS, IX}, {a, b, b, 0}, {1, 0, 0, 0}, 3}
#define CY_C1(a, b) {{IX, INY, IX, IX}, {a, a, 0, 0}, {IX, IY, IX, IX}, {b, b, 0, 0}, {1, 1, 0, 0}, 2}, \
                    {{IX, IY, IX, IX}, {a, a, 0, 0}, {IY, IX, IX, IX}, {b, b, 0, 0}, {1, 1, 0, 0}, 2}
#define CY_X1(a, b) {{IX, INY, IX, IX}, {a, a, 0, 0}, {IS, IT, IX, IX}, {b, b, 0, 0}, {1, 1, 0, 0}, 2}, \
                    {{IX, IY, IX, IX}, {a, a, 0, 0}, {IT, IS, IX, IX}, {b, b, 0, 0}, {1, 1, 0, 0}, 2}
__device__ const CycRow CYC[12] = { CY_C0(0, 3), CY_X1(2, 5), CY_C0(1, 4), CY_C1(0, 3), CY_C0(2, 5), CY_C1(1, 4) };     // rows 2m + part
__device__ __attribute__((noinline)) Fq dot_cyc(const uint32_t* A, const Fq& old, int m, int part) {
  const CycRow row = CYC[2 * m + part];
  Cols k; cols_zero(k);
#pragma unroll 1
  for (int it = 0; it < 4; ++it) {
    const bool valid = it < row.n;
    Fq p = lld(A + (row.fa[it] + row.ia[it]) * SW);
    const uint32_t sh = row.dbl[it];
#pragma unroll
    for (int i = 0; i < 14; ++i) p.v[i] = valid ? p.v[i] << sh : 0u;
    cols_mac(k, p, lld(A + (row.fb[it] + row.ib[it]) * SW));
  }
  const Fq t = cols_reduce(k);
  uint32_t v[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) v[i] = 3 * t.v[i] + ((m & 1) ? 2 * old.v[i] : 2 * (FqC::subk(i) - old.v[i]));      // 3t +- 2 old, one pass (< 20p)
  return fp_lazy_reduce<FqC>(v);
}
// my coefficient of A * L for a sparse second operand with coefficients 0 (in Fq: only X[0] is non-zero), 3 and 4 — a Miller line
// a + c w^3 + b w^4.  L is an image whose slots X/Y/S/T [0], [3], [4] are filled.
__device__ __attribute__((noinline)) Fq dot_line(const uint32_t* A, const uint32_t* L, int m, int part) {
  Cols k; cols_zero(k);
  // coefficient 0 of the line is real: contributes l0 * (x_m | y_m)
  cols_mac(k, lld(A + ((part ? IY : IX) + m) * SW), lld(L + (IX + 0) * SW));
#pragma unroll 1
  for (int kk = 3; kk <= 4; ++kk) {
    int j = m - kk; const bool wrap = j < 0; if (wrap) j += 6;
    const Fq p1 = lld(A + (IX + j) * SW);
    const Fq u1 = lld(L + ((part ? (wrap ? IT : IY) : (wrap ? IS : IX)) + kk) * SW);
    cols_mac(k, p1, u1);
    const Fq p2 = lld(A + ((part ? IY : INY) + j) * SW);
    const Fq u2 = lld(L + ((part ? (wrap ? IS : IX) : (wrap ? IT : IY)) + kk) * SW);
    cols_mac(k, p2, u2);
  }
  return cols_reduce(k);
}

// ---- group-level Fq12 operations (value = one Fq per lane) --------------------------------------------------------------------------
struct Ctx {
  uint32_t* base; uint32_t* dummy; Role r;
  __device__ uint32_t* img(int i) const { return base + i * IMG_SLOTS * SW; }
  __device__ uint32_t* pt(int s) const { return base + (NIMG * IMG_SLOTS + s) * SW; }
};

__device__ inline Fq d_mul(const Ctx& c, const Fq& a, const Fq& b) {        // uses images 0 and 1
  expand_first(c.img(0), a, c.r, c.dummy);
  expand(c.img(1), b, c.r, c.dummy);
  return dot_mul(c.img(0), c.img(1), c.r.m, c.r.part);
}
__device__ inline Fq d_sqr(const Ctx& c, const Fq& a) {
  expand(c.img(0), a, c.r, c.dummy);
  return dot_sqr(c.img(0), c.r.m, c.r.part);
}
// a * (image `bi`, already expanded): exponentiation loops keep their constant factor expanded
__device__ inline Fq d_mul_img(const Ctx& c, const Fq& a, int bi) {
  expand_first(c.img(0), a, c.r, c.dummy);
  return dot_mul(c.img(0), c.img(bi), c.r.m, c.r.part);
}
__device__ inline Fq d_cyc_sqr(const Ctx& c, const Fq& a) {
  expand(c.img(0), a, c.r, c.dummy);
  return dot_cyc(c.img(0), a, c.r.m, c.r.part);
}
__device__ inline Fq d_conj(const Ctx& c, const Fq& a) { return fsel(c.r.m & 1, fp_neg(a), a); }      // w -> -w
// the partner lane's coefficient (x <-> y of the same m): lanes g and g^1 (groups start at even lanes)
__device__ inline Fq partner(const Fq& a) { Fq r;
#pragma unroll
  for (int i = 0; i < 14; ++i) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], 0xB1, 0xF, 0xF, true);
  return r; }
__device__ inline Fq frob_limbs(int K, int idx, int comp) { Fq g;
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    uint32_t v = 0;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      const uint32_t w = K == 1 ? (comp ? frob1_limb(t, 1, i) : frob1_limb(t, 0, i)) : frob2_limb(t, 0, i);
      v = (t == idx) ? w : v;
    }
    g.v[i] = v;
  }
  return g; }
// pi^1: conj_u on every coefficient, times gamma_m = g0 + g1 u:  (x - y u)(g0 + g1 u) = (x g0 + y g1) + (x g1 - y g0) u
__device__ inline Fq d_frob1(const Ctx& c, const Fq& a) {
  const Fq o = partner(a);
  const Fq g0 = frob_limbs(1, c.r.m, 0), g1 = frob_limbs(1, c.r.m, 1);
  // part 0: me = x, o = y: x g0 + y g1      part 1: me = y, o = x: x g1 + (-y) g0
  Cols k; cols_zero(k);
  cols_mac(k, c.r.part ? o : a, c.r.part ? g1 : g0);
  cols_mac(k, c.r.part ? fp_neg(a) : o, c.r.part ? g0 : g1);
  return cols_reduce(k);
}
__device__ inline Fq d_frob2(const Ctx& c, const Fq& a) { return fp_mul(a, frob_limbs(2, c.r.m, 0)); }     // gamma^(2) lies in Fq, no conjugation

// ABI position (u32 words) of coefficient (m, part) inside an Fq12 {w1,w0}{v2,v1,v0}{u1,u0}
__device__ inline int abi_word(int m, int part) { const int i = m & 1, k = m >> 1; return (i ? 0 : 72) + (2 - k) * 24 + (part ? 0 : 12); }

// inverse: gathered into the tower form by the group's first lane (once per pairing)
__device__ __attribute__((noinline)) Fq d_inv(const Ctx& c, const Fq& a) {
  lst(c.img(0) + ((c.r.part ? IY : IX) + c.r.m) * SW, a);
  gsync();
  if (c.r.g == 0) {
    auto co = [&](int m) { return Fq2{lld(c.img(0) + (IX + m) * SW), lld(c.img(0) + (IY + m) * SW)}; };
    Fq12 t; t.c0 = Fq6{co(0), co(2), co(4)}; t.c1 = Fq6{co(1), co(3), co(5)};
    const Fq12 r = fq12_is_zero(t) ? t : fq12_inv(t);      // a Miller value of 0 (P outside G1, discarded later) must not reach the inversion
    auto put = [&](int m, const Fq2& v) { lst(c.img(0) + (IX + m) * SW, v.c0); lst(c.img(0) + (IY + m) * SW, v.c1); };
    put(0, r.c0.c0); put(2, r.c0.c1); put(4, r.c0.c2); put(1, r.c1.c0); put(3, r.c1.c1); put(5, r.c1.c2);
  }
  gsync();
  const Fq out = lld(c.img(0) + ((c.r.part ? IY : IX) + c.r.m) * SW);
  gsync();
  return out;
}

__device__ inline Ctx make_ctx(uint32_t* lds) {
  const int lane = threadIdx.x;
  const bool shadow = lane >= GPW * GL;                  // lanes 60-63 shadow lanes 0-3 of the last group: same loads, same (duplicate) LDS stores, no results
  const int grp = shadow ? GPW - 1 : lane / GL, g = shadow ? lane - GPW * GL : lane % GL;
  Ctx c; c.base = lds + grp * GROUP_WORDS; c.dummy = lds + GPW * GROUP_WORDS + (lane & 3) * SW;
  c.r.g = shadow ? g + 100 : g;                          // a shadow is never "the group's first lane"
  c.r.m = g >> 1; c.r.part = g & 1;
  return c;
}
constexpr int LDS_WORDS = GPW * GROUP_WORDS + 4 * SW;

// ---- diagnostic: Fq12 operations on the distributed form, against zkt_fq12_*_batch (tests/test_gpu_dpairing.py) --------------------
// op: 0 mul, 1 square, 2 frobenius, 3 frobenius^2, 4 conjugate (q^6), 5 inverse, 6 Granger-Scott square (input in the cyclotomic subgroup)
__global__ void __launch_bounds__(64) k_dfq12_op(int op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t* __restrict__ out, size_t n) {
  __shared__ uint32_t lds[LDS_WORDS];
  const Ctx c = make_ctx(lds);
  const int lane = threadIdx.x;
  size_t e = (size_t)blockIdx.x * GPW + (lane / GL < GPW ? lane / GL : GPW - 1);
  const bool live = e < n && lane < GPW * GL;
  if (e >= n) e = n - 1;
  const int w = abi_word(c.r.m, c.r.part);
  const Fq x = ld_fp<FqC>(a + e * 144 + w), y = ld_fp<FqC>((b ? b : a) + e * 144 + w);
  Fq r;
  switch (op) {            // uniform
    case 0: r = d_mul(c, x, y); break;
    case 1: r = d_sqr(c, x); break;
    case 2: r = d_frob1(c, x); break;
    case 3: r = d_frob2(c, x); break;
    case 4: r = d_conj(c, x); break;
    case 6: r = d_cyc_sqr(c, x); break;
    default: r = d_inv(c, x); break;
  }
  if (live) st_fp<FqC>(out + e * 144 + w, r);
}


// =====================================================================================================================================
// The G1 side of the Miller loop, spread over the lanes.  Fq values live in numbered LDS slots of the group; one LEVEL of the dependency
// graph of a point step is one (or a sum of two) Fq products per lane on slots named by a per-lane table, plus an optional linear
// follow-up  c1 * prod +- aux  (E = 3A, X + B, 2 Y Z, U - X ...).  Cheap linear glue between levels (D, X3, ...) is computed by every
// lane and stored by the group's first lane.  Formulas: miller_dbl_step / miller_add_step of pairing.h.
// =====================================================================================================================================
constexpr int LINE_IMG = 2;
constexpr int SL(int img, int field, int k) { return img * IMG_SLOTS + field + k; }              // absolute slot of an image entry
constexpr int PT0 = NIMG * IMG_SLOTS;
enum : int { P_X = PT0, P_Y, P_Z, P_XP, P_YP, P_YN, P_XQ0, P_XQ1, P_XQS, P_XQT, P_YQ0, P_YQ1, P_YQS, P_YQT, P_ZERO, P_DUMMY,
             P_A, P_B, P_ZZ, P_YZ, P_XB, P_E, P_Z3, P_C, P_T, P_EE, P_EX, P_EZ, P_Z3Z, P_DX, P_C8, P_Y3P,
             P_U, P_H, P_ZZZ, P_S, P_R, P_HH, P_HHH, P_VV, P_RR, P_X3, P_VX, P_END };
static_assert(P_END - PT0 <= PT_SLOTS, "point slot file too small");
constexpr int L_A = SL(LINE_IMG, IX, 0);                                                             // line coefficient 0 (real)
constexpr int L_C0 = SL(LINE_IMG, IX, 3), L_C1 = SL(LINE_IMG, IY, 3), L_CS = SL(LINE_IMG, IS, 3), L_CT = SL(LINE_IMG, IT, 3);   // coefficient 3 = c (v w)
constexpr int L_B0 = SL(LINE_IMG, IX, 4), L_B1 = SL(LINE_IMG, IY, 4), L_BS = SL(LINE_IMG, IS, 4), L_BT = SL(LINE_IMG, IT, 4);   // coefficient 4 = b (v^2)

struct LOp { uint8_t a1, b1, a2, b2, dst, aux, dst2, c1, sub_aux, neg_a1, neg_a2; };
#define NOP_ {P_ZERO, P_ZERO, P_ZERO, P_ZERO, P_DUMMY, P_ZERO, P_DUMMY, 1, 0, 0, 0}
//                          a1      b1      a2      b2      dst     aux     dst2     c1 sub nega1 nega2
__device__ const LOp DBL_L1[12] = {
  {P_X, P_X, P_ZERO, P_ZERO, P_A, P_ZERO, P_E, 3, 0, 0, 0},        // A = X^2, E = 3A
  {P_Y, P_Y, P_ZERO, P_ZERO, P_B, P_X, P_XB, 1, 0, 0, 0},          // B = Y^2, XB = X + B
  {P_Z, P_Z, P_ZERO, P_ZERO, P_ZZ, P_ZERO, P_DUMMY, 1, 0, 0, 0},   // ZZ
  {P_Y, P_Z, P_ZERO, P_ZERO, P_YZ, P_ZERO, P_Z3, 2, 0, 0, 0},      // YZ, Z3 = 2 Y Z
  NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_};
__device__ const LOp DBL_L2[12] = {
  {P_B, P_B, P_ZERO, P_ZERO, P_C, P_ZERO, P_DUMMY, 1, 0, 0, 0},
  {P_XB, P_XB, P_ZERO, P_ZERO, P_T, P_ZERO, P_DUMMY, 1, 0, 0, 0},
  {P_E, P_E, P_ZERO, P_ZERO, P_EE, P_ZERO, P_DUMMY, 1, 0, 0, 0},
  {P_E, P_X, P_ZERO, P_ZERO, P_EX, P_ZERO, P_DUMMY, 1, 0, 0, 0},
  {P_E, P_ZZ, P_ZERO, P_ZERO, P_EZ, P_ZERO, P_DUMMY, 1, 0, 0, 0},
  {P_Z3, P_ZZ, P_ZERO, P_ZERO, P_Z3Z, P_ZERO, P_DUMMY, 1, 0, 0, 0},
  NOP_, NOP_, NOP_, NOP_, NOP_, NOP_};
__device__ const LOp DBL_L3[12] = {
  {P_E, P_DX, P_ZERO, P_ZERO, P_DUMMY, P_C8, P_Y, 1, 1, 0, 0},     // Y3 = E (D - X3) - 8C
  {P_EZ, P_XQ0, P_ZERO, P_ZERO, L_B0, P_ZERO, P_DUMMY, 1, 0, 1, 0},   // b = Xq * (-E ZZ): four products (x, y, s, t forms)
  {P_EZ, P_XQ1, P_ZERO, P_ZERO, L_B1, P_ZERO, P_DUMMY, 1, 0, 1, 0},
  {P_EZ, P_XQS, P_ZERO, P_ZERO, L_BS, P_ZERO, P_DUMMY, 1, 0, 1, 0},
  {P_EZ, P_XQT, P_ZERO, P_ZERO, L_BT, P_ZERO, P_DUMMY, 1, 0, 1, 0},
  {P_Z3Z, P_YQ0, P_ZERO, P_ZERO, L_C0, P_ZERO, P_DUMMY, 1, 0, 0, 0},  // c = Yq * (Z3 ZZ)
  {P_Z3Z, P_YQ1, P_ZERO, P_ZERO, L_C1, P_ZERO, P_DUMMY, 1, 0, 0, 0},
  {P_Z3Z, P_YQS, P_ZERO, P_ZERO, L_CS, P_ZERO, P_DUMMY, 1, 0, 0, 0},
  {P_Z3Z, P_YQT, P_ZERO, P_ZERO, L_CT, P_ZERO, P_DUMMY, 1, 0, 0, 0},
  NOP_, NOP_, NOP_};
// addition step V + (xp, yp'):  yp' = P_YP or P_YN (digit -1), chosen by the caller through the `ysel` offset on slots named P_YP
__device__ const LOp ADD_L1[12] = { {P_Z, P_Z, P_ZERO, P_ZERO, P_ZZ, P_ZERO, P_DUMMY, 1, 0, 0, 0}, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_};
__device__ const LOp ADD_L2[12] = {
  {P_XP, P_ZZ, P_ZERO, P_ZERO, P_U, P_X, P_H, 1, 1, 0, 0},          // U = xp ZZ, H = U - X
  {P_ZZ, P_Z, P_ZERO, P_ZERO, P_ZZZ, P_ZERO, P_DUMMY, 1, 0, 0, 0},
  NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_};
__device__ const LOp ADD_L3[12] = {
  {P_YP, P_ZZZ, P_ZERO, P_ZERO, P_S, P_Y, P_R, 1, 1, 0, 0},         // S = yp ZZZ, R = S - Y
  {P_H, P_H, P_ZERO, P_ZERO, P_HH, P_ZERO, P_DUMMY, 1, 0, 0, 0},
  {P_Z, P_H, P_ZERO, P_ZERO, P_Z3, P_ZERO, P_DUMMY, 1, 0, 0, 0},
  NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_};
__device__ const LOp ADD_L4[12] = {
  {P_H, P_HH, P_ZERO, P_ZERO, P_HHH, P_ZERO, P_DUMMY, 1, 0, 0, 0},
  {P_X, P_HH, P_ZERO, P_ZERO, P_VV, P_ZERO, P_DUMMY, 1, 0, 0, 0},
  {P_R, P_R, P_ZERO, P_ZERO, P_RR, P_ZERO, P_DUMMY, 1, 0, 0, 0},
  {P_R, P_XP, P_YP, P_Z3, L_A, P_ZERO, P_DUMMY, 1, 0, 0, 1},       // a = R xp - Z3 yp  (two products, one reduction)
  {P_R, P_XQ0, P_ZERO, P_ZERO, L_B0, P_ZERO, P_DUMMY, 1, 0, 1, 0},  // b = Xq * (-R)
  {P_R, P_XQ1, P_ZERO, P_ZERO, L_B1, P_ZERO, P_DUMMY, 1, 0, 1, 0},
  {P_R, P_XQS, P_ZERO, P_ZERO, L_BS, P_ZERO, P_DUMMY, 1, 0, 1, 0},
  {P_R, P_XQT, P_ZERO, P_ZERO, L_BT, P_ZERO, P_DUMMY, 1, 0, 1, 0},
  {P_Z3, P_YQ0, P_ZERO, P_ZERO, L_C0, P_ZERO, P_DUMMY, 1, 0, 0, 0}, // c = Yq * Z3
  {P_Z3, P_YQ1, P_ZERO, P_ZERO, L_C1, P_ZERO, P_DUMMY, 1, 0, 0, 0},
  {P_Z3, P_YQS, P_ZERO, P_ZERO, L_CS, P_ZERO, P_DUMMY, 1, 0, 0, 0},
  {P_Z3, P_YQT, P_ZERO, P_ZERO, L_CT, P_ZERO, P_DUMMY, 1, 0, 0, 0}};
__device__ const LOp ADD_L5[12] = {
  {P_R, P_VX, P_Y, P_HHH, P_Y, P_ZERO, P_DUMMY, 1, 0, 0, 1},        // Y3 = R (V - X3) - Y HHH  (reads Y, writes Y: same lane)
  NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_};

// one level: prod = a1 b1 (+ a2 b2) on slots; store; optional  c1 prod +- aux  into a second slot.  `ysel` is added to every slot index
// equal to P_YP (selects -P for a digit -1).  All lanes run the same code; g picks the table row.
template <bool TWO>
__device__ __attribute__((noinline)) void level(const Ctx& c, const LOp* tab, int ysel) {
  const int g = c.r.g >= 100 ? c.r.g - 100 : c.r.g;
  const LOp op = tab[g];
  auto slot = [&](int s) { return c.base + (s == P_YP ? s + ysel : s) * SW; };
  Fq a1 = lld(slot(op.a1));
  if (op.neg_a1) a1 = fp_neg(a1);
  const Fq b1 = lld(slot(op.b1));
  Cols k; cols_zero(k);
  cols_mac(k, a1, b1);
  if (TWO) {
    Fq a2 = lld(slot(op.a2));
    a2 = fsel(op.neg_a2, fp_neg(a2), a2);
    cols_mac(k, a2, lld(slot(op.b2)));
  }
  const Fq prod = cols_reduce(k);
  const Fq aux = lld(slot(op.aux));
  uint32_t v[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) v[i] = op.c1 * prod.v[i] + (op.sub_aux ? FqC::subk(i) - aux.v[i] : aux.v[i]);
  const Fq lin = fp_lazy_reduce<FqC>(v);
  gsync();                                   // every operand of this level has been read (a level may overwrite its own inputs)
  if (c.r.g < 100) { lst(c.base + op.dst * SW, prod); lst(c.base + op.dst2 * SW, lin); }
  gsync();
}
// NOTE: rows whose dst/dst2 is P_DUMMY all write the same slot (garbage, never read).  Rows with aux = P_ZERO and c1 = 1 write lin = prod.

__device__ inline void store0(const Ctx& c, int s, const Fq& v) { if (c.r.g == 0) lst(c.base + s * SW, v); }
__device__ inline Fq slotv(const Ctx& c, int s) { return lld(c.base + s * SW); }

// V <- 2V and the tangent line into the line image
__device__ __attribute__((noinline)) void point_dbl(const Ctx& c) {
  level<false>(c, DBL_L1, 0);
  level<false>(c, DBL_L2, 0);
  {   // D = 2(t - A - C); X3 = E^2 - 2D; DX = D - X3; C8 = 8C; a = E X - 2B      (every lane computes, the first stores)
    const Fq t = slotv(c, P_T), A = slotv(c, P_A), C = slotv(c, P_C), EE = slotv(c, P_EE), EX = slotv(c, P_EX), B = slotv(c, P_B), Z3 = slotv(c, P_Z3);
    const Fq D = fp_dbl(fp_subsub(t, A, C));
    const Fq X3 = fp_sub2(EE, fp_zero<FqC>(), D);
    gsync();
    store0(c, P_DX, fp_sub(D, X3)); store0(c, P_C8, fp_dbl(fp_dbl(fp_dbl(C)))); store0(c, L_A, fp_sub2(EX, fp_zero<FqC>(), B));
    store0(c, P_X, X3); store0(c, P_Z, Z3);
    gsync();
  }
  level<false>(c, DBL_L3, 0);
}
// V <- V + (xp, +-yp) and the chord line
__device__ __attribute__((noinline)) void point_add(const Ctx& c, bool neg) {
  const int ysel = neg ? P_YN - P_YP : 0;
  level<false>(c, ADD_L1, ysel);
  level<false>(c, ADD_L2, ysel);
  level<false>(c, ADD_L3, ysel);
  level<true>(c, ADD_L4, ysel);
  {   // X3 = R^2 - HHH - 2V; VX = V - X3
    const Fq RR = slotv(c, P_RR), HHH = slotv(c, P_HHH), VV = slotv(c, P_VV), Z3 = slotv(c, P_Z3);
    const Fq X3 = fp_sub2(RR, HHH, VV);
    gsync();
    store0(c, P_VX, fp_sub(VV, X3)); store0(c, P_X3, X3);
    gsync();
  }
  level<true>(c, ADD_L5, ysel);
  {
    const Fq X3 = slotv(c, P_X3), Z3 = slotv(c, P_Z3);
    gsync();
    store0(c, P_X, X3); store0(c, P_Z, Z3);
    gsync();
  }
}

// f * line, line already in the line image
__device__ inline Fq d_mul_line(const Ctx& c, const Fq& f) {
  expand_first(c.img(0), f, c.r, c.dummy);
  return dot_line(c.img(0), c.img(LINE_IMG), c.r.m, c.r.part);
}
__device__ inline Fq d_one(const Ctx& c) { return fsel(c.r.m == 0 && c.r.part == 0, fp_one<FqC>(), fp_zero<FqC>()); }

// a^|x| (cyclotomic subgroup), |x| = 0xd201000000010000;  uses image 1 for the base
__device__ __attribute__((noinline)) Fq d_pow_xabs(const Ctx& c, const Fq& a) {
  expand(c.img(1), a, c.r, c.dummy);
  Fq r = a;
#pragma unroll 1
  for (int i = 62; i >= 0; --i) {
    r = d_cyc_sqr(c, r);
    if ((BLS_X_ABS >> i) & 1) r = d_mul_img(c, r, 1);
  }
  return r;
}
// a^e1, e1 = (x-1)^2/3: width-3 signed digits {0, +-1, +-3} as fq12_pow_e1;  a in image 1, a^3 in image 2;  r * conj(m) = conj(conj(r) * m)
__device__ __attribute__((noinline)) Fq d_pow_e1(const Ctx& c, const Fq& a) {
  const Fq a3 = d_mul(c, d_cyc_sqr(c, a), a);
  expand(c.img(1), a, c.r, c.dummy);
  expand(c.img(2), a3, c.r, c.dummy);
  Fq r = a; bool started = false;
#pragma unroll 1
  for (int i = 0; i < E1_WNAF_DIGITS; ++i) {
    uint32_t nz = 0, ng = 0, th = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { nz = (j == (i >> 5)) ? e1_wnaf_nz_word(j) : nz; ng = (j == (i >> 5)) ? e1_wnaf_neg_word(j) : ng; th = (j == (i >> 5)) ? e1_wnaf_three_word(j) : th; }
    if (started) r = d_cyc_sqr(c, r);
    if ((nz >> (i & 31)) & 1) {
      const bool three = (th >> (i & 31)) & 1, neg = (ng >> (i & 31)) & 1;
      if (!started) { r = three ? a3 : a; if (neg) r = d_conj(c, r); started = true; }
      else if (neg) r = d_conj(c, d_mul_img(c, d_conj(c, r), three ? 2 : 1));
      else r = d_mul_img(c, r, three ? 2 : 1);
    }
  }
  return r;
}
// f^((q^12-1)/r), exact: the sequence of final_exponentiation (pairing.h)
// SHORT: f comes from the 127-step loop and the result is raised to 2 x^2 - 1 on the way out (final_exponentiation_t<true>, pairing.h)
template <bool SHORT> __device__ __attribute__((noinline)) Fq d_final_exp(const Ctx& c, const Fq& f) {
  Fq t = d_inv(c, f);
  Fq a = d_conj(c, f);
  Fq g = d_mul(c, a, t);                         // ^(q^6-1)
  t = d_frob2(c, g);
  g = d_mul(c, t, g);                            // ^(q^2+1): easy part
  a = d_pow_e1(c, g);
  t = d_conj(c, d_pow_xabs(c, a));               // a^x
  Fq b = d_frob1(c, a);
  a = d_mul(c, t, b);                            // ^(x+q)
  t = d_conj(c, d_pow_xabs(c, a));
  b = d_conj(c, d_pow_xabs(c, t));               // a^(x^2)
  t = d_frob2(c, a);
  b = d_mul(c, b, t);
  t = d_conj(c, a);
  a = d_mul(c, b, t);                            // ^(x^2+q^2-1)
  if constexpr (!SHORT) return d_mul(c, a, g);
  else {
    t = d_mul(c, a, g);                          // eta
    b = d_cyc_sqr(c, d_frob2(c, t));             // eta^(2 x^2)
    return d_mul(c, b, d_conj(c, t));
  }
}

// f^(3 (q^12-1)/r): final_exponentiation_3h (pairing.h) — for the deciding kernels only
__device__ __attribute__((noinline)) Fq d_final_exp_3h(const Ctx& c, const Fq& f) {
  Fq t = d_inv(c, f);
  Fq a = d_conj(c, f);
  Fq g = d_mul(c, a, t);
  t = d_frob2(c, g);
  g = d_mul(c, t, g);                            // easy part
  t = d_mul(c, d_conj(c, d_pow_xabs(c, g)), d_conj(c, g));      // g^(x-1)
  a = d_mul(c, d_conj(c, d_pow_xabs(c, t)), d_conj(c, t));      // g^((x-1)^2)
  t = d_conj(c, d_pow_xabs(c, a));
  Fq b = d_frob1(c, a);
  a = d_mul(c, t, b);                            // ^(x+q)
  t = d_conj(c, d_pow_xabs(c, a));
  b = d_conj(c, d_pow_xabs(c, t));
  t = d_frob2(c, a);
  b = d_mul(c, b, t);
  t = d_conj(c, a);
  a = d_mul(c, b, t);                            // ^(x^2+q^2-1)
  return d_mul(c, a, d_mul(c, d_cyc_sqr(c, g), g));             // * g^3
}

// f_{r-1,P}(untwist(Q)) up to Fq6 factors for the group's pair, as miller_g1_g2 (pairing.h); in_g1 <- r P == infinity.
// SHORT: f_{x^2,P} over the 127 bits of x^2 as miller_g1_g2_short — the callers have Q's membership of G2 and the curve equations checked by
// k_short_loop_guards beside this kernel, and redo what fails them.
template <bool SHORT> __device__ __attribute__((noinline)) Fq d_miller(const Ctx& c, const Aff<FqOps>& p, const Aff<Fq2Ops>& q, bool& in_g1) {
  {   // slot file: every lane computes the same values, the group's first lane stores them
    const Fq2 xi_inv = xi_inv_const();
    const Fq2 Xq = fq2_mul(q.x, xi_inv), Yq = fq2_mul(q.y, xi_inv);
    store0(c, P_X, p.x); store0(c, P_Y, p.y); store0(c, P_Z, fp_one<FqC>());
    store0(c, P_XP, p.x); store0(c, P_YP, p.y); store0(c, P_YN, fp_neg(p.y));
    store0(c, P_XQ0, Xq.c0); store0(c, P_XQ1, Xq.c1); store0(c, P_XQS, fp_sub(Xq.c0, Xq.c1)); store0(c, P_XQT, fp_add(Xq.c0, Xq.c1));
    store0(c, P_YQ0, Yq.c0); store0(c, P_YQ1, Yq.c1); store0(c, P_YQS, fp_sub(Yq.c0, Yq.c1)); store0(c, P_YQT, fp_add(Yq.c0, Yq.c1));
    store0(c, P_ZERO, fp_zero<FqC>());
    gsync();                               // (line coefficient 0 is real: its Y / S / T slots are never read by dot_line)
  }
  Fq f = d_one(c);
#pragma unroll 1
  for (int i = 0; i < (SHORT ? MILLER_X2_NBITS : MILLER_NAF_DIGITS); ++i) {
    uint32_t nz = 0, ng = 0;
    if constexpr (SHORT) {
#pragma unroll
      for (int j = 0; j < 4; ++j) nz = (j == (i >> 5)) ? miller_x2_bits_word(j) : nz;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) { nz = (j == (i >> 5)) ? miller_naf_nz_word(j) : nz; ng = (j == (i >> 5)) ? miller_naf_neg_word(j) : ng; }
    }
    const bool bit = (nz >> (i & 31)) & 1, neg = (ng >> (i & 31)) & 1;
    f = d_sqr(c, f);
    point_dbl(c);
    f = d_mul_line(c, f);
    if (bit) { point_add(c, neg); f = d_mul_line(c, f); }
  }
  // r P == infinity?  V = (r-1) P must equal -P;  after the short loop V = x^2 P must equal (BETA xp, -yp)  (miller_pt_is_x2, pairing.h)
  const Fq X = slotv(c, P_X), Y = slotv(c, P_Y), Z = slotv(c, P_Z);
  const Fq ZZ = fp_sqr(Z);
  const Fq xw = SHORT ? fp_mul(g1_beta_const(), p.x) : p.x;
  in_g1 = !fp_is_zero(Z) && fp_eq(fp_mul(xw, ZZ), X) && fp_eq(fp_mul(fp_mul(p.y, ZZ), Z), fp_neg(Y));
  return f;
}

// one Tate pairing per group.  Marks elements whose P is outside G1 for k_tate_exact_marked, exactly as k_tate does.
template <bool SHORT> __global__ void __launch_bounds__(64) k_dtate(const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g2, uint32_t* __restrict__ out, size_t n,
                                              unsigned long long* err, uint32_t mark_word, uint32_t mark) {
  __shared__ uint32_t lds[LDS_WORDS];
  const Ctx c = make_ctx(lds);
  const int lane = threadIdx.x;
  size_t e = (size_t)blockIdx.x * GPW + (lane / GL < GPW ? lane / GL : GPW - 1);
  const bool live = e < n && lane < GPW * GL;
  if (e >= n) e = n - 1;
  Aff<FqOps> p = PtIO<FqOps>::ld(g1 + e * ABI_G1_WORDS);
  Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(g2 + e * ABI_G2_WORDS);
  const bool inf = p.inf || q.inf;
  if (inf) {                                   // RationalFunction::new_* / eval_with_* panic on infinity: report, then run on dummy data to keep the barriers uniform
    if (live) atomicMin(err, (unsigned long long)e);
    p.x = fp_one<FqC>(); p.y = fp_one<FqC>(); q.x = fq2_one(); q.y = fq2_one();
  }
  bool in_g1;
  const Fq f = d_miller<SHORT>(c, p, q, in_g1);
  const Fq r = d_final_exp<SHORT>(c, f);
  if (!live || inf) return;
  if (!in_g1) { if (c.r.g == 0) out[e * 144 + mark_word] = mark; return; }
  st_fp<FqC>(out + e * 144 + abi_word(c.r.m, c.r.part), r);
}

// prod_k tate(+-P_k, Q_k) == target (or == 1) per element, K <= 4 pairs: the K Miller loops run side by side in K groups of one wave,
// the group results meet in the first group's LDS image, ONE final exponentiation follows.  Same contract as k_pairing_product_check /
// k_groth16_verify (zkt_pairing.hip): infinity -> error index, a G1 argument outside the order-r subgroup -> ok = 0.
template <int K, bool SHORT>
__global__ void __launch_bounds__(64) k_dproduct(PairArgs a, const uint32_t* __restrict__ target, uint32_t* __restrict__ ok, size_t n, unsigned long long* err,
                                                 const uint8_t* __restrict__ kcount) {
  __shared__ uint32_t lds[LDS_WORDS];
  const Ctx c = make_ctx(lds);
  constexpr int EPB = GPW / K;                                    // elements per wave
  const int lane = threadIdx.x;
  const int grp = lane / GL < GPW ? lane / GL : GPW - 1;
  const bool idle = grp >= EPB * K;                               // spare groups (and lanes 60-63) repeat the first pair and never report
  const int eb = idle ? 0 : grp / K, pair = idle ? 0 : grp % K;
  size_t e = (size_t)blockIdx.x * EPB + eb;
  const bool live = !idle && e < n && lane < GPW * GL;
  if (e >= n) e = n - 1;
  uint32_t* lead = lds + (idle ? grp : eb * K) * GROUP_WORDS;     // the element's first group
  Aff<FqOps> p = PtIO<FqOps>::ld(a.g1[pair] + e * a.s1[pair]);
  Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(a.g2[pair] + e * a.s2[pair]);
  bool inf = p.inf || q.inf;
  if (inf) { p.x = fp_one<FqC>(); p.y = fp_one<FqC>(); q.x = fq2_one(); q.y = fq2_one(); }
  if (a.neg[pair]) p.y = fp_neg(p.y);
  bool in_g1;
  Fq f = d_miller<SHORT>(c, p, q, in_g1);
  // kcount (optional): element e multiplies only its first kcount[e] pairs.  The caller fills the unused slots with a copy of pair 0, so everything up to
  // here ran on valid points; the spare group hands over one and raises no flag (Pinocchio's 2-pair and 3-pair equalities in ONE launch of K = 3).
  if (kcount && pair >= (int)kcount[e]) { f = d_one(c); in_g1 = true; inf = false; }
  // element-wide flags: every lane of the element's K groups must agree
  const int start = eb * K * GL;
  const unsigned long long emask = (K * GL >= 64 ? ~0ull : ((1ull << (K * GL)) - 1ull)) << start;
  const unsigned long long bad_inf = __ballot(inf) & emask, bad_g1 = __ballot(!in_g1) & emask;
  // the other groups' Miller values into the first group's image 1, one after the other; everybody multiplies (only the first group's product is used)
#pragma unroll 1
  for (int k = 1; k < K; ++k) {
    expand(pair == k && !idle ? lead + 1 * IMG_SLOTS * SW : c.img(2), f, c.r, c.dummy);
    expand_first(c.img(0), f, c.r, c.dummy);
    const Fq prod = dot_mul(c.img(0), lead + 1 * IMG_SLOTS * SW, c.r.m, c.r.part);
    f = (pair == 0) ? prod : f;                                   // groups k > 0 keep their own value until it has been handed over
  }
  // SHORT: (prod tate)^(1/(2x^2-1)) is one exactly when the Tate product is; against a target the corrected exponentiation gives the product itself
  const Fq r = (SHORT && target) ? d_final_exp<true>(c, f) : d_final_exp<false>(c, f);
  uint32_t w[12]; fp_to_words(r, w);
  uint32_t diff = 0;
  const int off = abi_word(c.r.m, c.r.part);
#pragma unroll
  for (int i = 0; i < 12; ++i) diff |= w[i] ^ (target ? target[off + i] : (off == 132 && i == 0 ? 1u : 0u));      // canonical one: w0.v0.u0 = 1
  const unsigned long long lmask = ((1ull << GL) - 1ull) << start;                  // the first group's lanes
  const unsigned long long neq = __ballot(diff != 0) & lmask;
  if (live && pair == 0 && c.r.g == 0) {
    if (bad_inf) { atomicMin(err, (unsigned long long)e); ok[e] = 0; }
    else ok[e] = bad_g1 ? 3u : (neq == 0 ? 1u : 0u);      // 3 = OK_EXACT (zkt_pairing.hip): a G1 argument outside the order-r subgroup — decided the reference's way behind
  }
}


// =====================================================================================================================================
// The 63-step loop (optimal ate, pairing.h "Decisions through the 63-step loop") on the lane groups, for the DECIDING entry points at small batch sizes.
// The chain runs on Q: a Jacobian point over Fq2 on the twist, its six Fq coordinates in numbered slots.  An Fq2 product or square is TWO rows of a level
// (real part a0 b0 - a1 b1, imaginary part a0 b1 + a1 b0; a square's imaginary part 2 a0 a1 is one product and a doubling in the linear follow-up), so
// a level of twelve lanes carries six of them: a doubling is three levels, an addition five, as on the G1 side.  Formulas and scalings are those of
// ate_dbl_step / ate_add_step (pairing.h): line = a0 xp + a1 yp w + c4 w^4 with a0 = -xi E ZZ (or -xi R), a1 = xi Z3 ZZ (or xi Z3), c4 = E X - 2B (or R xq - Z3 yq),
// D = 4 X B in place of 2((X + B)^2 - A - C).  The line image holds coefficients 0, 1 and 4 (S = x - y, T = x + y forms for the two that can wrap).
// The chain ends on [|x|] Q, which is compared with -psi(Q): Q in G2.  P on E and in G1, Q on E' are tested by k_ate_guards beside this kernel.
// =====================================================================================================================================
enum : int { A_TX0 = PT0, A_TX1, A_TY0, A_TY1, A_TZ0, A_TZ1, A_QX0, A_QX1, A_QY0, A_QY1, A_XP, A_YP, A_ZERO, A_DUMMY,
             A_E0, A_E1, A_B0, A_B1, A_ZZ0, A_ZZ1, A_Z30, A_Z31, A_C0, A_C1, A_XB0, A_XB1, A_EE0, A_EE1, A_EX0, A_EX1, A_EZ0, A_EZ1, A_ZQ0, A_ZQ1,
             A_DX0, A_DX1, A_C80, A_C81, A_LA0, A_LA1, A_LB0, A_LB1, A_LBS, A_LBT, A_RV0, A_RV1, A_YH0, A_YH1, A_END,
             // the addition step reuses the doubling's temporaries (nothing but T, Q and P lives across steps)
             A_H0 = A_E0, A_H1 = A_E1, A_ZZZ0 = A_B0, A_ZZZ1 = A_B1, A_R0 = A_C0, A_R1 = A_C1, A_HH0 = A_XB0, A_HH1 = A_XB1, A_HHH0 = A_EE0, A_HHH1 = A_EE1,
             A_V0 = A_EX0, A_V1 = A_EX1, A_RR0 = A_EZ0, A_RR1 = A_EZ1, A_RQ0 = A_ZQ0, A_RQ1 = A_ZQ1, A_YQ0 = A_DX0, A_YQ1 = A_DX1, A_VX0 = A_C80, A_VX1 = A_C81 };
static_assert(A_END - PT0 <= PT_SLOTS, "point slot file too small for the ate step");
constexpr int LA_X(int k) { return SL(LINE_IMG, IX, k); }
constexpr int LA_Y(int k) { return SL(LINE_IMG, IY, k); }
constexpr int LA_S(int k) { return SL(LINE_IMG, IS, k); }
constexpr int LA_T(int k) { return SL(LINE_IMG, IT, k); }
#define ANOP_ {A_ZERO, A_ZERO, A_ZERO, A_ZERO, A_DUMMY, A_ZERO, A_DUMMY, 1, 0, 0, 0}
// real / imaginary part of the Fq2 product (u0 + u1 i)(v0 + v1 i) into `d`;  square: real part as a product with itself, imaginary part 2 u0 u1 through the follow-up
#define ARE(u0, u1, v0, v1, d) {u0, v0, u1, v1, d, A_ZERO, A_DUMMY, 1, 0, 0, 1}
#define AIM(u0, u1, v0, v1, d) {u0, v1, u1, v0, d, A_ZERO, A_DUMMY, 1, 0, 0, 0}
#define ASQI(u0, u1, d, c) {u0, u1, A_ZERO, A_ZERO, A_DUMMY, A_ZERO, d, c, 0, 0, 0}
#define AFQ(u, s, d) {u, s, A_ZERO, A_ZERO, d, A_ZERO, A_DUMMY, 1, 0, 0, 0}
__device__ const LOp ADBL_L1[12] = {
  {A_TX0, A_TX0, A_TX1, A_TX1, A_DUMMY, A_ZERO, A_E0, 3, 0, 0, 1},          // E = 3 X^2
  ASQI(A_TX0, A_TX1, A_E1, 6),
  ARE(A_TY0, A_TY1, A_TY0, A_TY1, A_B0), ASQI(A_TY0, A_TY1, A_B1, 2),        // B = Y^2
  ARE(A_TZ0, A_TZ1, A_TZ0, A_TZ1, A_ZZ0), ASQI(A_TZ0, A_TZ1, A_ZZ1, 2),      // ZZ = Z^2
  {A_TY0, A_TZ0, A_TY1, A_TZ1, A_DUMMY, A_ZERO, A_Z30, 2, 0, 0, 1},          // Z3 = 2 Y Z
  {A_TY0, A_TZ1, A_TY1, A_TZ0, A_DUMMY, A_ZERO, A_Z31, 2, 0, 0, 0},
  ANOP_, ANOP_, ANOP_, ANOP_};
__device__ const LOp ADBL_L2[12] = {
  ARE(A_B0, A_B1, A_B0, A_B1, A_C0), ASQI(A_B0, A_B1, A_C1, 2),              // C = B^2
  ARE(A_TX0, A_TX1, A_B0, A_B1, A_XB0), AIM(A_TX0, A_TX1, A_B0, A_B1, A_XB1),   // X B  (D = 4 X B)
  ARE(A_E0, A_E1, A_E0, A_E1, A_EE0), ASQI(A_E0, A_E1, A_EE1, 2),
  ARE(A_E0, A_E1, A_TX0, A_TX1, A_EX0), AIM(A_E0, A_E1, A_TX0, A_TX1, A_EX1),
  ARE(A_E0, A_E1, A_ZZ0, A_ZZ1, A_EZ0), AIM(A_E0, A_E1, A_ZZ0, A_ZZ1, A_EZ1),
  ARE(A_Z30, A_Z31, A_ZZ0, A_ZZ1, A_ZQ0), AIM(A_Z30, A_Z31, A_ZZ0, A_ZZ1, A_ZQ1)};
__device__ const LOp ADBL_L3[12] = {
  {A_E0, A_DX0, A_E1, A_DX1, A_DUMMY, A_C80, A_TY0, 1, 1, 0, 1},             // Y3 = E (D - X3) - 8C
  {A_E0, A_DX1, A_E1, A_DX0, A_DUMMY, A_C81, A_TY1, 1, 1, 0, 0},
  AFQ(A_LA0, A_XP, LA_X(0)), AFQ(A_LA1, A_XP, LA_Y(0)),                      // c0 = a0 xp
  AFQ(A_LB0, A_YP, LA_X(1)), AFQ(A_LB1, A_YP, LA_Y(1)), AFQ(A_LBS, A_YP, LA_S(1)), AFQ(A_LBT, A_YP, LA_T(1)),      // c1 = a1 yp and its s, t forms
  ANOP_, ANOP_, ANOP_, ANOP_};
__device__ const LOp AADD_L1[12] = { ARE(A_TZ0, A_TZ1, A_TZ0, A_TZ1, A_ZZ0), ASQI(A_TZ0, A_TZ1, A_ZZ1, 2), ANOP_, ANOP_, ANOP_, ANOP_, ANOP_, ANOP_, ANOP_, ANOP_, ANOP_, ANOP_};
__device__ const LOp AADD_L2[12] = {
  {A_QX0, A_ZZ0, A_QX1, A_ZZ1, A_DUMMY, A_TX0, A_H0, 1, 1, 0, 1},            // H = xq ZZ - X
  {A_QX0, A_ZZ1, A_QX1, A_ZZ0, A_DUMMY, A_TX1, A_H1, 1, 1, 0, 0},
  ARE(A_ZZ0, A_ZZ1, A_TZ0, A_TZ1, A_ZZZ0), AIM(A_ZZ0, A_ZZ1, A_TZ0, A_TZ1, A_ZZZ1),
  ANOP_, ANOP_, ANOP_, ANOP_, ANOP_, ANOP_, ANOP_, ANOP_};
__device__ const LOp AADD_L3[12] = {
  {A_QY0, A_ZZZ0, A_QY1, A_ZZZ1, A_DUMMY, A_TY0, A_R0, 1, 1, 0, 1},          // R = yq ZZZ - Y   (R overwrites nothing it reads: ZZZ lives in the B slots, R in the C slots)
  {A_QY0, A_ZZZ1, A_QY1, A_ZZZ0, A_DUMMY, A_TY1, A_R1, 1, 1, 0, 0},
  ARE(A_H0, A_H1, A_H0, A_H1, A_HH0), ASQI(A_H0, A_H1, A_HH1, 2),
  ARE(A_TZ0, A_TZ1, A_H0, A_H1, A_Z30), AIM(A_TZ0, A_TZ1, A_H0, A_H1, A_Z31),       // Z3 = Z H
  ANOP_, ANOP_, ANOP_, ANOP_, ANOP_, ANOP_};
__device__ const LOp AADD_L4[12] = {
  ARE(A_H0, A_H1, A_HH0, A_HH1, A_HHH0), AIM(A_H0, A_H1, A_HH0, A_HH1, A_HHH1),
  ARE(A_TX0, A_TX1, A_HH0, A_HH1, A_V0), AIM(A_TX0, A_TX1, A_HH0, A_HH1, A_V1),
  ARE(A_R0, A_R1, A_R0, A_R1, A_RR0), ASQI(A_R0, A_R1, A_RR1, 2),
  ARE(A_R0, A_R1, A_QX0, A_QX1, A_RQ0), AIM(A_R0, A_R1, A_QX0, A_QX1, A_RQ1),
  ARE(A_Z30, A_Z31, A_QY0, A_QY1, A_YQ0), AIM(A_Z30, A_Z31, A_QY0, A_QY1, A_YQ1),
  ANOP_, ANOP_};
__device__ const LOp AADD_L5[12] = {
  ARE(A_R0, A_R1, A_VX0, A_VX1, A_RV0), AIM(A_R0, A_R1, A_VX0, A_VX1, A_RV1),
  ARE(A_TY0, A_TY1, A_HHH0, A_HHH1, A_YH0), AIM(A_TY0, A_TY1, A_HHH0, A_HHH1, A_YH1),
  AFQ(A_LA0, A_XP, LA_X(0)), AFQ(A_LA1, A_XP, LA_Y(0)),
  AFQ(A_LB0, A_YP, LA_X(1)), AFQ(A_LB1, A_YP, LA_Y(1)), AFQ(A_LBS, A_YP, LA_S(1)), AFQ(A_LBT, A_YP, LA_T(1)),
  ANOP_, ANOP_};
// what both steps leave for the line: a0 = -xi e, a1 = xi z (with the s, t forms of a1 for the w^1 coefficient's wrap) and c4 with its four forms
__device__ inline void ate_line_glue(const Ctx& c, const Fq& e0, const Fq& e1, const Fq& z0, const Fq& z1, const Fq& c40, const Fq& c41) {
  const Fq b0 = fp_sub(z0, z1), b1 = fp_add(z0, z1);
  store0(c, A_LA0, fp_sub(e1, e0)); store0(c, A_LA1, fp_neg(fp_add(e0, e1)));
  store0(c, A_LB0, b0); store0(c, A_LB1, b1); store0(c, A_LBS, fp_sub(b0, b1)); store0(c, A_LBT, fp_add(b0, b1));
  store0(c, LA_X(4), c40); store0(c, LA_Y(4), c41); store0(c, LA_S(4), fp_sub(c40, c41)); store0(c, LA_T(4), fp_add(c40, c41));
}
__device__ __attribute__((noinline)) void apoint_dbl(const Ctx& c) {
  level<true>(c, ADBL_L1, 0);
  level<true>(c, ADBL_L2, 0);
  {   // D = 4 XB; X3 = E^2 - 2D; DX = D - X3; C8 = 8C; c4 = E X - 2B; a0 = -xi E ZZ; a1 = xi Z3 ZZ     (every lane computes, the first stores)
    const Fq xb0 = slotv(c, A_XB0), xb1 = slotv(c, A_XB1), ee0 = slotv(c, A_EE0), ee1 = slotv(c, A_EE1), c0 = slotv(c, A_C0), c1 = slotv(c, A_C1);
    const Fq ex0 = slotv(c, A_EX0), ex1 = slotv(c, A_EX1), b0 = slotv(c, A_B0), b1 = slotv(c, A_B1), ez0 = slotv(c, A_EZ0), ez1 = slotv(c, A_EZ1);
    const Fq zq0 = slotv(c, A_ZQ0), zq1 = slotv(c, A_ZQ1), z30 = slotv(c, A_Z30), z31 = slotv(c, A_Z31);
    const Fq d0 = fp_dbl(fp_dbl(xb0)), d1 = fp_dbl(fp_dbl(xb1));
    const Fq x30 = fp_sub2(ee0, fp_zero<FqC>(), d0), x31 = fp_sub2(ee1, fp_zero<FqC>(), d1);
    gsync();
    store0(c, A_DX0, fp_sub(d0, x30)); store0(c, A_DX1, fp_sub(d1, x31));
    store0(c, A_C80, fp_dbl(fp_dbl(fp_dbl(c0)))); store0(c, A_C81, fp_dbl(fp_dbl(fp_dbl(c1))));
    ate_line_glue(c, ez0, ez1, zq0, zq1, fp_sub2(ex0, fp_zero<FqC>(), b0), fp_sub2(ex1, fp_zero<FqC>(), b1));
    store0(c, A_TX0, x30); store0(c, A_TX1, x31); store0(c, A_TZ0, z30); store0(c, A_TZ1, z31);
    gsync();
  }
  level<true>(c, ADBL_L3, 0);
}
__device__ __attribute__((noinline)) void apoint_add(const Ctx& c) {
  level<true>(c, AADD_L1, 0);
  level<true>(c, AADD_L2, 0);
  level<true>(c, AADD_L3, 0);
  level<true>(c, AADD_L4, 0);
  {   // X3 = R^2 - HHH - 2V; VX = V - X3; c4 = R xq - Z3 yq; a0 = -xi R; a1 = xi Z3
    const Fq rr0 = slotv(c, A_RR0), rr1 = slotv(c, A_RR1), h0 = slotv(c, A_HHH0), h1 = slotv(c, A_HHH1), v0 = slotv(c, A_V0), v1 = slotv(c, A_V1);
    const Fq r0 = slotv(c, A_R0), r1 = slotv(c, A_R1), z30 = slotv(c, A_Z30), z31 = slotv(c, A_Z31);
    const Fq rq0 = slotv(c, A_RQ0), rq1 = slotv(c, A_RQ1), yq0 = slotv(c, A_YQ0), yq1 = slotv(c, A_YQ1);
    const Fq x30 = fp_sub2(rr0, h0, v0), x31 = fp_sub2(rr1, h1, v1);
    gsync();
    store0(c, A_VX0, fp_sub(v0, x30)); store0(c, A_VX1, fp_sub(v1, x31));
    ate_line_glue(c, r0, r1, z30, z31, fp_sub(rq0, yq0), fp_sub(rq1, yq1));
    store0(c, A_TX0, x30); store0(c, A_TX1, x31); store0(c, A_TZ0, z30); store0(c, A_TZ1, z31);
    gsync();
  }
  level<true>(c, AADD_L5, 0);
  {   // Y3 = R (V - X3) - Y HHH
    const Fq rv0 = slotv(c, A_RV0), rv1 = slotv(c, A_RV1), yh0 = slotv(c, A_YH0), yh1 = slotv(c, A_YH1);
    gsync();
    store0(c, A_TY0, fp_sub(rv0, yh0)); store0(c, A_TY1, fp_sub(rv1, yh1));
    gsync();
  }
}
// my coefficient of A * L for a line with coefficients 0, 1 and 4 (all in Fq2): three complex products, the wrapped ones through the s, t forms
__device__ __attribute__((noinline)) Fq dot_line_ate(const uint32_t* A, const uint32_t* L, int m, int part) {
  Cols k; cols_zero(k);
#pragma unroll 1
  for (int t = 0; t < 3; ++t) {
    const int kk = t == 0 ? 0 : (t == 1 ? 1 : 4);
    int j = m - kk; const bool wrap = j < 0; if (wrap) j += 6;
    const Fq p1 = lld(A + (IX + j) * SW);
    const Fq u1 = lld(L + ((part ? (wrap ? IT : IY) : (wrap ? IS : IX)) + kk) * SW);
    cols_mac(k, p1, u1);
    const Fq p2 = lld(A + ((part ? IY : INY) + j) * SW);
    const Fq u2 = lld(L + ((part ? (wrap ? IS : IX) : (wrap ? IT : IY)) + kk) * SW);
    cols_mac(k, p2, u2);
  }
  return cols_reduce(k);
}
__device__ inline Fq d_mul_line_ate(const Ctx& c, const Fq& f) {
  expand_first(c.img(0), f, c.r, c.dummy);
  return dot_line_ate(c.img(0), c.img(LINE_IMG), c.r.m, c.r.part);
}
// f_{|x|,Q}(P) up to factors the final exponentiation kills, as miller_ate_multi<1,0> (pairing.h); q_in_g2 <- the chain ended on -psi(Q)
__device__ __attribute__((noinline)) Fq d_miller_ate(const Ctx& c, const Aff<FqOps>& p, const Aff<Fq2Ops>& q, bool& q_in_g2) {
  store0(c, A_TX0, q.x.c0); store0(c, A_TX1, q.x.c1); store0(c, A_TY0, q.y.c0); store0(c, A_TY1, q.y.c1); store0(c, A_TZ0, fp_one<FqC>()); store0(c, A_TZ1, fp_zero<FqC>());
  store0(c, A_QX0, q.x.c0); store0(c, A_QX1, q.x.c1); store0(c, A_QY0, q.y.c0); store0(c, A_QY1, q.y.c1);
  store0(c, A_XP, p.x); store0(c, A_YP, p.y); store0(c, A_ZERO, fp_zero<FqC>());
  gsync();
  Fq f = d_one(c);
#pragma unroll 1
  for (int i = 62; i >= 0; --i) {
    if (i != 62) f = d_sqr(c, f);
    apoint_dbl(c);
    f = d_mul_line_ate(c, f);
    if ((BLS_X_ABS >> i) & 1) { apoint_add(c); f = d_mul_line_ate(c, f); }
  }
  Jac<Fq2Ops> T;
  T.X = Fq2{slotv(c, A_TX0), slotv(c, A_TX1)}; T.Y = Fq2{slotv(c, A_TY0), slotv(c, A_TY1)}; T.Z = Fq2{slotv(c, A_TZ0), slotv(c, A_TZ1)};
  q_in_g2 = ate_end_is_psi(T, q.x, q.y);
  return f;
}
// prod_k a(Q_k, +-P_k) == target (or == 1) per element on the 63-step loop, K <= 4 pairs in K groups of one wave (layout and contract of k_dproduct).  `target` is the ate
// counterpart of a key's alpha_beta (k_ate_key_prep).  An element with a Q outside G2 gets ok = 2 (OK_REDO of zkt_pairing.hip): the 255-step kernel behind decides it.
template <int K>
__global__ void __launch_bounds__(64) k_dproduct_ate(PairArgs a, const uint32_t* __restrict__ target, uint32_t* __restrict__ ok, size_t n, unsigned long long* err,
                                                     const uint8_t* __restrict__ kcount) {
  __shared__ uint32_t lds[LDS_WORDS];
  const Ctx c = make_ctx(lds);
  constexpr int EPB = GPW / K;
  const int lane = threadIdx.x;
  const int grp = lane / GL < GPW ? lane / GL : GPW - 1;
  const bool idle = grp >= EPB * K;
  const int eb = idle ? 0 : grp / K, pair = idle ? 0 : grp % K;
  size_t e = (size_t)blockIdx.x * EPB + eb;
  const bool live = !idle && e < n && lane < GPW * GL;
  if (e >= n) e = n - 1;
  uint32_t* lead = lds + (idle ? grp : eb * K) * GROUP_WORDS;
  Aff<FqOps> p = PtIO<FqOps>::ld(a.g1[pair] + e * a.s1[pair]);
  Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(a.g2[pair] + e * a.s2[pair]);
  bool inf = p.inf || q.inf;
  if (inf) { p.x = fp_one<FqC>(); p.y = fp_one<FqC>(); q.x = fq2_one(); q.y = fq2_one(); }
  if (a.neg[pair]) p.y = fp_neg(p.y);
  bool q_ok;
  Fq f = d_miller_ate(c, p, q, q_ok);
  if (kcount && pair >= (int)kcount[e]) { f = d_one(c); q_ok = true; inf = false; }
  const int start = eb * K * GL;
  const unsigned long long emask = (K * GL >= 64 ? ~0ull : ((1ull << (K * GL)) - 1ull)) << start;
  const unsigned long long bad_inf = __ballot(inf) & emask, bad_q = __ballot(!q_ok) & emask;
#pragma unroll 1
  for (int k = 1; k < K; ++k) {
    expand(pair == k && !idle ? lead + 1 * IMG_SLOTS * SW : c.img(2), f, c.r, c.dummy);
    expand_first(c.img(0), f, c.r, c.dummy);
    const Fq prod = dot_mul(c.img(0), lead + 1 * IMG_SLOTS * SW, c.r.m, c.r.part);
    f = (pair == 0) ? prod : f;
  }
  const Fq r = d_final_exp_3h(c, f);
  uint32_t w[12]; fp_to_words(r, w);
  uint32_t diff = 0;
  const int off = abi_word(c.r.m, c.r.part);
#pragma unroll
  for (int i = 0; i < 12; ++i) diff |= w[i] ^ (target ? target[off + i] : (off == 132 && i == 0 ? 1u : 0u));
  const unsigned long long lmask = ((1ull << GL) - 1ull) << start;
  const unsigned long long neq = __ballot(diff != 0) & lmask;
  if (live && pair == 0 && c.r.g == 0) {
    if (bad_inf) { atomicMin(err, (unsigned long long)e); ok[e] = 0; }
    else if (bad_q) ok[e] = 2u;
    else ok[e] = neq == 0 ? 1u : 0u;
  }
}
// The ate counterpart of a verifying key's alpha_beta for k_ate_key_prep's buffer: a(beta, alpha) raised like the deciding kernels raise their products, by one lane group
// (4 ms; on ONE lane of the tower code it was 25 of the 45 ms a key cost when first seen).  out: 144 words in the ABI's Fq12 layout; flag |= bit when beta's chain ended on -psi(beta).
// Block 1 of the same launch (round 4): tate(alpha, beta) itself, which the host compares with the key's stored alpha_beta (verifier.rs:48 compares against that GTPoint, so
// only a key whose alpha_beta IS the pairing of its alpha and beta may be served by the 63-step loop).  Two launches one after the other were 4.0 + 4.8 ms of a key's entry;
// side by side they are 4.8.  gt (144 words, zeroed by the host) stays zero — never a pairing value — when alpha is outside G1 or an argument is infinity.
__global__ void __launch_bounds__(64) k_key_ab(const uint32_t* __restrict__ alpha, const uint32_t* __restrict__ beta, uint32_t* __restrict__ out, uint32_t* __restrict__ flag, uint32_t bit,
                                               uint32_t* __restrict__ gt) {
  __shared__ uint32_t lds[LDS_WORDS];
  const Ctx c = make_ctx(lds);
  Aff<FqOps> p = PtIO<FqOps>::ld(alpha);
  Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(beta);
  const bool inf = p.inf || q.inf;
  if (inf) { p.x = fp_one<FqC>(); p.y = fp_one<FqC>(); q.x = fq2_one(); q.y = fq2_one(); }      // run on dummy data: the barriers stay uniform
  if (blockIdx.x == 0) {
    bool q_ok;
    const Fq f = d_miller_ate(c, p, q, q_ok);
    const Fq r = d_final_exp_3h(c, f);
    if (inf || !q_ok || threadIdx.x >= GL) return;                                                  // every group computed the same pairing; the first one reports
    st_fp<FqC>(out + abi_word(c.r.m, c.r.part), r);
    if (threadIdx.x == 0) atomicOr(flag, bit);
  } else {
    bool in_g1;
    const Fq f = d_miller<true>(c, p, q, in_g1);
    const Fq r = d_final_exp<true>(c, f);
    if (inf || !in_g1 || threadIdx.x >= GL) return;
    st_fp<FqC>(gt + abi_word(c.r.m, c.r.part), r);
  }
}
}  // namespace dp

hipError_t launch_dfq12_op(int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(dp::k_dfq12_op, dim3((unsigned)((n + dp::GPW - 1) / dp::GPW)), dim3(64), 0, s, op, a, b, out, n);
  return hipGetLastError();
}

hipError_t launch_dproduct(const PairArgs& a, int K, const uint32_t* target, uint32_t* ok, size_t n, unsigned long long* err, bool short_loop, hipStream_t s, const uint8_t* kcount) {
  if (n == 0) return hipSuccess;
  auto blocks = [&](int epb) { return dim3((unsigned)((n + epb - 1) / epb)); };
#define ZKT_DPRODUCT(KK, EPB) if (short_loop) hipLaunchKernelGGL((dp::k_dproduct<KK, true>), blocks(EPB), dim3(64), 0, s, a, target, ok, n, err, kcount); \
                             else hipLaunchKernelGGL((dp::k_dproduct<KK, false>), blocks(EPB), dim3(64), 0, s, a, target, ok, n, err, kcount)
  switch (K) {
    case 1: ZKT_DPRODUCT(1, 5); break;
    case 2: ZKT_DPRODUCT(2, 2); break;
    case 3: ZKT_DPRODUCT(3, 1); break;
    case 4: ZKT_DPRODUCT(4, 1); break;
    default: return hipErrorInvalidValue;
  }
#undef ZKT_DPRODUCT
  return hipGetLastError();
}
hipError_t launch_key_ab(const uint32_t* alpha, const uint32_t* beta, uint32_t* out, uint32_t* flag, uint32_t bit, uint32_t* gt, hipStream_t s) {
  hipLaunchKernelGGL(dp::k_key_ab, dim3(2), dim3(64), 0, s, alpha, beta, out, flag, bit, gt);
  return hipGetLastError();
}
hipError_t launch_dproduct_ate(const PairArgs& a, int K, const uint32_t* target, uint32_t* ok, size_t n, unsigned long long* err, hipStream_t s, const uint8_t* kcount) {
  if (n == 0) return hipSuccess;
  auto blocks = [&](int epb) { return dim3((unsigned)((n + epb - 1) / epb)); };
  switch (K) {
    case 1: hipLaunchKernelGGL((dp::k_dproduct_ate<1>), blocks(5), dim3(64), 0, s, a, target, ok, n, err, kcount); break;
    case 2: hipLaunchKernelGGL((dp::k_dproduct_ate<2>), blocks(2), dim3(64), 0, s, a, target, ok, n, err, kcount); break;
    case 3: hipLaunchKernelGGL((dp::k_dproduct_ate<3>), blocks(1), dim3(64), 0, s, a, target, ok, n, err, kcount); break;
    case 4: hipLaunchKernelGGL((dp::k_dproduct_ate<4>), blocks(1), dim3(64), 0, s, a, target, ok, n, err, kcount); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_dtate(const uint32_t* g1, const uint32_t* g2, uint32_t* out, size_t n, unsigned long long* err, uint32_t mark_word, uint32_t mark, bool short_loop, hipStream_t s) {
  if (n == 0) return hipSuccess;
  if (short_loop) hipLaunchKernelGGL(dp::k_dtate<true>, dim3((unsigned)((n + dp::GPW - 1) / dp::GPW)), dim3(64), 0, s, g1, g2, out, n, err, mark_word, mark);
  else hipLaunchKernelGGL(dp::k_dtate<false>, dim3((unsigned)((n + dp::GPW - 1) / dp::GPW)), dim3(64), 0, s, g1, g2, out, n, err, mark_word, mark);
  return hipGetLastError();
}

}  // namespace zkt
