// Batched group kernels: one point operation per lane (rows a7, a8, a16 of SURVEY §8).
//   impl_affine_add!        curves/macros.rs:34-163
//   impl_scalar_mul_point!  curves/macros.rs:1-32
//   Neg                     bls12_381/g1_point.rs:177-195, g2_point.rs:91-107
// Points arrive and leave as canonical affine structs (include/zkt.h); inside a lane the
// point is Jacobian over Montgomery residues and is normalised once at the end.
#include <mutex>
#include "abi.h"
#include "zkt_internal.h"

namespace zkt {

static inline unsigned nblk(size_t n, int tpb) { return (unsigned)((n + tpb - 1) / tpb); }

template <class F>
__global__ void __launch_bounds__(64) k_group_add(const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n) {   // out may alias a
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  constexpr int W = PtIO<F>::WORDS;
  Aff<F> p = PtIO<F>::ld(a + i * W), q = PtIO<F>::ld(b + i * W);
  PtIO<F>::st(out + i * W, jac_to_aff(jac_add_aff(jac_from_aff(p), q)));
}
template <class F>
__global__ void __launch_bounds__(64) k_group_neg(const uint32_t* __restrict__ a, uint32_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  constexpr int W = PtIO<F>::WORDS;
  Aff<F> p = PtIO<F>::ld(a + i * W);
  if (!p.inf) p.y = F::neg(p.y);
  PtIO<F>::st(out + i * W, p);
}
template <class F>
__global__ void __launch_bounds__(64) k_group_mul(const uint32_t* __restrict__ pts, int pt_stride, const uint32_t* __restrict__ scalars, int kw, int k_stride,
                                                  uint32_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  constexpr int W = PtIO<F>::WORDS;
  Aff<F> p = PtIO<F>::ld(pts + i * (size_t)pt_stride);   // pt_stride = W, or 0 for one fixed base (g * y_i, crs.rs:85-135)
  const uint32_t* k = scalars + i * (size_t)k_stride;   // k_stride = kw, or 0 for one scalar for every point (gg * x, bulletproofs.rs:44)
  PtIO<F>::st(out + i * W, jac_to_aff(scalar_mul_aff<F>(p, k, kw)));     // NAF double-and-add, curve.h
}

template <class F> static hipError_t add_t(const uint32_t* a, const uint32_t* b, uint32_t* o, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_group_add<F>, dim3(nblk(n, 64)), dim3(64), 0, s, a, b, o, n); return hipGetLastError(); }
template <class F> static hipError_t neg_t(const uint32_t* a, uint32_t* o, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_group_neg<F>, dim3(nblk(n, 64)), dim3(64), 0, s, a, o, n); return hipGetLastError(); }
template <class F> static hipError_t mul_t(const uint32_t* p, bool fixed, const uint32_t* k, int kw, bool fixed_k, uint32_t* o, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_group_mul<F>, dim3(nblk(n, 64)), dim3(64), 0, s, p, fixed ? 0 : PtIO<F>::WORDS, k, kw, fixed_k ? 0 : kw, o, n); return hipGetLastError(); }

hipError_t launch_group_add(int grp, const uint32_t* a, const uint32_t* b, uint32_t* o, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  switch (grp) { case G_G1: return add_t<FqOps>(a, b, o, n, s); case G_G2: return add_t<Fq2Ops>(a, b, o, n, s); case G_SECP: return add_t<SpOps>(a, b, o, n, s); }
  return hipErrorInvalidValue;
}
hipError_t launch_group_neg(int grp, const uint32_t* a, uint32_t* o, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  switch (grp) { case G_G1: return neg_t<FqOps>(a, o, n, s); case G_G2: return neg_t<Fq2Ops>(a, o, n, s); case G_SECP: return neg_t<SpOps>(a, o, n, s); }
  return hipErrorInvalidValue;
}
hipError_t launch_group_mul(int grp, const uint32_t* p, const uint32_t* k, int kw, uint32_t* o, size_t n, hipStream_t s, bool fixed_base, bool fixed_scalar) {
  if (n == 0) return hipSuccess;
  switch (grp) { case G_G1: return mul_t<FqOps>(p, fixed_base, k, kw, fixed_scalar, o, n, s); case G_G2: return mul_t<Fq2Ops>(p, fixed_base, k, kw, fixed_scalar, o, n, s); case G_SECP: return mul_t<SpOps>(p, fixed_base, k, kw, fixed_scalar, o, n, s); }
  return hipErrorInvalidValue;
}
// ---- predicates (row a16) ----------------------------------------------------------------------------------------------------
//   RationalPoint::is_rational_point   g1_point.rs:97-113, g2_point.rs:70-82, secp256k1/affine_point.rs:92-104: y^2 == x^3 + b, false at infinity
//   order-r membership                 r P == infinity — what every G1Point / G2Point the reference builds satisfies (g * k,
//                                      get_random_point g1_point.rs:83-88) and what the verification entry points assume
template <class F> struct CurveB;
template <> struct CurveB<FqOps> { __device__ static Fq b() { uint32_t w[12] = {4}; return fp_from_words<FqC>(w); } };               // y^2 = x^3 + 4
template <> struct CurveB<Fq2Ops> { __device__ static Fq2 b() { const Fq f = CurveB<FqOps>::b(); return Fq2{f, f}; } };             // y^2 = x^3 + 4(1+u)
template <> struct CurveB<SpOps> { __device__ static SpE b() { uint32_t w[8] = {7}; return fp_from_words<SpC>(w); } };              // y^2 = x^3 + 7
template <class F, int PRED>
__global__ void __launch_bounds__(64) k_group_pred(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ order, int order_words, uint32_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  Aff<F> p = PtIO<F>::ld(pts + i * PtIO<F>::WORDS);
  if (PRED == 0) {
    out[i] = !p.inf && F::eq(F::sqr(p.y), F::add(F::mul(F::sqr(p.x), p.x), CurveB<F>::b()));
  } else {
    uint32_t k[SCALAR_MAX_LIMBS];
    for (int j = 0; j < order_words; ++j) k[j] = order[j];
    out[i] = p.inf || jac_is_inf(scalar_mul_aff<F>(p, k, order_words));
  }
}
hipError_t launch_group_pred(int grp, int pred, const uint32_t* pts, const uint32_t* order, int order_words, uint32_t* out, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  dim3 g(nblk(n, 64)), t(64);
#define ZKT_PRED(F) if (pred == 0) hipLaunchKernelGGL((k_group_pred<F, 0>), g, t, 0, s, pts, order, order_words, out, n); \
                    else hipLaunchKernelGGL((k_group_pred<F, 1>), g, t, 0, s, pts, order, order_words, out, n)
  switch (grp) { case G_G1: ZKT_PRED(FqOps); break; case G_G2: ZKT_PRED(Fq2Ops); break; case G_SECP: ZKT_PRED(SpOps); break; default: return hipErrorInvalidValue; }
#undef ZKT_PRED
  return hipGetLastError();
}

// P on E and in G1 for up to four point arrays in one launch (what the 63-step verification kernels need of their G1 arguments, pairing.h g1_in_subgroup): out[k * n + i].
// Its own kernel because it needs ~110 registers: four waves per SIMD here, against the one wave of a 512-register verification kernel in which the same 127-doubling
// chain issued at less than half the rate.  A point at infinity reads as fitting (the verification kernels report it as the reference's panic themselves).
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) k_g1_fits(G1Fits f, uint32_t* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  const int k = blockIdx.y;
  if (i >= n) return;
  const Aff<FqOps> p = PtIO<FqOps>::ld(f.pts[k] + i * f.stride[k]);
  out[(size_t)k * n + i] = p.inf || (g1_on_curve(p.x, p.y) && g1_in_subgroup(p.x, p.y));
}
hipError_t launch_g1_fits(const G1Fits& f, int K, uint32_t* out, size_t n, hipStream_t s) {
  if (n == 0 || K == 0) return hipSuccess;
  if (K < 0 || K > 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_g1_fits, dim3(nblk(n, 64), (unsigned)K), dim3(64), 0, s, f, out, n);
  return hipGetLastError();
}

// Several independent batched scalar multiplications in ONE launch.  A 255-step double-and-add costs ~4 ms of latency however few
// points it covers, so callers that issue many small independent ones per step (every level of the inner-product argument,
// bulletproofs.rs:36-47) pay that latency once instead of six times.
template <class F>
__global__ void __launch_bounds__(64) k_group_mul_segs(MulSegs segs, int kw, size_t total) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= total) return;
  int sidx = 0;
  while (sidx + 1 < segs.n && i >= segs.s[sidx].count) { i -= segs.s[sidx].count; ++sidx; }
  const MulSeg g = segs.s[sidx];
  constexpr int W = PtIO<F>::WORDS;
  Aff<F> p = PtIO<F>::ld(g.pts + i * (size_t)g.pt_stride);
  PtIO<F>::st(g.out + i * W, jac_to_aff(scalar_mul_aff<F>(p, g.k + i * (size_t)g.k_stride, kw)));
}
hipError_t launch_group_mul_segs(int grp, const MulSegs& segs, int kw, hipStream_t s) {
  size_t total = 0;
  for (int k = 0; k < segs.n; ++k) total += segs.s[k].count;
  if (total == 0) return hipSuccess;
  switch (grp) {
    case G_G1: hipLaunchKernelGGL(k_group_mul_segs<FqOps>, dim3(nblk(total, 64)), dim3(64), 0, s, segs, kw, total); break;
    case G_G2: hipLaunchKernelGGL(k_group_mul_segs<Fq2Ops>, dim3(nblk(total, 64)), dim3(64), 0, s, segs, kw, total); break;
    case G_SECP: hipLaunchKernelGGL(k_group_mul_segs<SpOps>, dim3(nblk(total, 64)), dim3(64), 0, s, segs, kw, total); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---- multiples of a group's generator: a comb table instead of 255 doublings per product ------------------------------------------
// Every fixed-base multiplication on this path is generator * scalar (CRS::new crs.rs:85-135, pinocchio/crs.rs:86-140, hash_to_g2point
// g2_point.rs:84-88).  table[w * 15 + d - 1] = d * 16^w * G (w < 64, d = 1..15: 960 affine points in internal coordinates, none at infinity since
// d * 16^w < r), built once per process and group by one launch of 960 double-and-add lanes.  A product is then at most 64 mixed additions and no doubling
// — ~660 field multiplications instead of ~2,700 — and the result is the same group element, hence the same canonical affine bytes.
template <class F> struct RawXY;
template <class C> struct RawXY<PrimeOps<C>> {
  static constexpr int CW = C::N;
  __device__ static Fp<C> ld(const uint32_t* p) { return ld_raw<C>(p); }
  __device__ static void st(uint32_t* p, const Fp<C>& a) { st_raw<C>(p, a); }
};
template <> struct RawXY<Fq2Ops> {
  static constexpr int CW = 2 * FqC::N;
  __device__ static Fq2 ld(const uint32_t* p) { Fq2 r; r.c0 = ld_raw<FqC>(p); r.c1 = ld_raw<FqC>(p + FqC::N); return r; }
  __device__ static void st(uint32_t* p, const Fq2& a) { st_raw<FqC>(p, a.c0); st_raw<FqC>(p + FqC::N, a.c1); }
};
static constexpr int COMB_ENTRIES = 64 * 15;
template <class F>
__global__ void __launch_bounds__(64) k_generator_table(const uint32_t* __restrict__ gen_abi, uint32_t* __restrict__ table) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= COMB_ENTRIES) return;
  constexpr int CW = RawXY<F>::CW;
  const int w = t / 15; const uint32_t d = (uint32_t)(t % 15) + 1u;
  uint32_t k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  k[w >> 3] = d << ((w & 7) * 4);                                  // d * 16^w: the digit sits inside one 32-bit word
  const Aff<F> a = jac_to_aff(scalar_mul_aff<F>(PtIO<F>::ld(gen_abi), k, 8));
  RawXY<F>::st(table + (size_t)t * 2 * CW, a.x); RawXY<F>::st(table + (size_t)t * 2 * CW + CW, a.y);
}
template <class F>
__global__ void __launch_bounds__(64) k_generator_mul(const uint32_t* __restrict__ table, const uint32_t* __restrict__ scalars, uint32_t* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  constexpr int CW = RawXY<F>::CW, W = PtIO<F>::WORDS;
  uint32_t k[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) k[j] = scalars[i * 8 + j];
  Jac<F> acc = jac_inf<F>();
#pragma unroll 1
  for (int w = 0; w < 64; ++w) {
    uint32_t kw = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) kw = (j == (w >> 3)) ? k[j] : kw;   // no dynamically indexed register array
    const uint32_t d = (kw >> ((w & 7) * 4)) & 15u;
    if (d) {
      const uint32_t* e = table + ((size_t)w * 15 + d - 1) * 2 * CW;
      Aff<F> q; q.x = RawXY<F>::ld(e); q.y = RawXY<F>::ld(e + CW); q.inf = false;
      acc = jac_add_aff(acc, q);
    }
  }
  PtIO<F>::st(out + i * W, jac_to_aff(acc));
}
namespace {
struct GeneratorTable { std::mutex mu; uint32_t* dev = nullptr; };
GeneratorTable g_gen_table[2];                                     // G_G1, G_G2
}
// zkt_shutdown: the comb tables live on the device zkt_init chose; a later zkt_init may pick another one, so they are rebuilt on first use
void group_release_device_state() {
  for (GeneratorTable& T : g_gen_table) { std::lock_guard<std::mutex> lk(T.mu); if (T.dev) { (void)hipFree(T.dev); T.dev = nullptr; } }
}
hipError_t launch_generator_mul(int grp, const uint32_t* gen_abi, const uint32_t* k, uint32_t* out, size_t n, hipStream_t s) {
  if (grp != G_G1 && grp != G_G2) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  GeneratorTable& T = g_gen_table[grp == G_G1 ? 0 : 1];
  {
    std::lock_guard<std::mutex> lk(T.mu);
    if (!T.dev) {                                                  // first use in this process: build on the caller's stream and wait, so that later callers on other streams find it complete
      const size_t bytes = (size_t)COMB_ENTRIES * 2 * (grp == G_G1 ? RawXY<FqOps>::CW : RawXY<Fq2Ops>::CW) * 4;
      uint32_t* mem = nullptr; hipError_t e;
      if ((e = hipMalloc((void**)&mem, bytes)) != hipSuccess) return e;
      if (grp == G_G1) hipLaunchKernelGGL(k_generator_table<FqOps>, dim3(COMB_ENTRIES / 64), dim3(64), 0, s, gen_abi, mem);
      else hipLaunchKernelGGL(k_generator_table<Fq2Ops>, dim3(COMB_ENTRIES / 64), dim3(64), 0, s, gen_abi, mem);
      if ((e = hipGetLastError()) != hipSuccess || (e = hipStreamSynchronize(s)) != hipSuccess) { (void)hipFree(mem); return e; }
      T.dev = mem;
    }
  }
  if (grp == G_G1) hipLaunchKernelGGL(k_generator_mul<FqOps>, dim3(nblk(n, 64)), dim3(64), 0, s, (const uint32_t*)T.dev, k, out, n);
  else hipLaunchKernelGGL(k_generator_mul<Fq2Ops>, dim3(nblk(n, 64)), dim3(64), 0, s, (const uint32_t*)T.dev, k, out, n);
  return hipGetLastError();
}

// ---- fixed-base products (zkt_internal.h) ------------------------------------------------------------------------------------
template <class F>
__global__ void __launch_bounds__(64) k_fixed_table(FixedTables ft) {
  const int w = threadIdx.x;
  constexpr int W = PtIO<F>::WORDS;
  const uint32_t* point = ft.point[blockIdx.x]; uint32_t* table = ft.table[blockIdx.x];
  Jac<F> j = jac_from_aff(PtIO<F>::ld(point));
  for (int d = 0; d < 4 * w; ++d) j = jac_dbl(j);               // lane 63: 252 doublings — the latency of the launch, paid once per point
  PtIO<F>::st(table + (size_t)w * W, jac_to_aff(j));
}
template <class F> __device__ inline void fixed_mul_wave(const FixedMul& m);
template <class F>
__global__ void __launch_bounds__(64) k_fixed_muls(FixedMuls f) { fixed_mul_wave<F>(f.m[blockIdx.x]); }
template <class F>
__global__ void __launch_bounds__(64) k_fixed_muls_batch(const uint32_t* __restrict__ tables, const uint32_t* __restrict__ k, uint32_t* __restrict__ out, size_t n, int n_pts) {
  constexpr int W = PtIO<F>::WORDS;
  const size_t i = blockIdx.x / (unsigned)n_pts; const int j = blockIdx.x % (unsigned)n_pts;
  fixed_mul_wave<F>(FixedMul{tables + (size_t)j * 64 * W, k + (i * n_pts + j) * 8, out + ((size_t)j * n + i) * W});
}
template <class F> __device__ inline void fixed_mul_wave(const FixedMul& m) {
  typedef typename F::E E;
  constexpr int W = PtIO<F>::WORDS, EW = sizeof(E) / 4, JW = 3 * EW;
  __shared__ uint32_t lds[32 * JW];
  const int w = threadIdx.x;
  const uint32_t d = (m.k[w >> 3] >> ((w & 7) * 4)) & 15u;      // w-th 4-bit digit of the scalar (8 canonical 32-bit words)
  const Aff<F> t = PtIO<F>::ld(m.table + (size_t)w * W);
  Jac<F> acc = jac_inf<F>();
  for (int b = 3; b >= 0; --b) { acc = jac_dbl(acc); if ((d >> b) & 1) acc = jac_add_aff(acc, t); }
  auto stj = [](uint32_t* p, const Jac<F>& a) { const uint32_t *x = (const uint32_t*)&a.X, *y = (const uint32_t*)&a.Y, *z = (const uint32_t*)&a.Z;
    for (int i = 0; i < EW; ++i) { p[i] = x[i]; p[EW + i] = y[i]; p[2 * EW + i] = z[i]; } };
  auto ldj = [](const uint32_t* p) { Jac<F> a; uint32_t *x = (uint32_t*)&a.X, *y = (uint32_t*)&a.Y, *z = (uint32_t*)&a.Z;
    for (int i = 0; i < EW; ++i) { x[i] = p[i]; y[i] = p[EW + i]; z[i] = p[2 * EW + i]; } return a; };
  for (int h = 32; h >= 1; h >>= 1) {
    if (w >= h && w < 2 * h) stj(lds + (w - h) * JW, acc);
    __syncthreads();
    if (w < h) acc = jac_add<F>(acc, ldj(lds + w * JW));
    __syncthreads();
  }
  if (w == 0) PtIO<F>::st(m.out, jac_to_aff(acc));
}
hipError_t launch_fixed_tables(int grp, const FixedTables& t, hipStream_t s) {
  if (t.n < 0 || t.n > 12) return hipErrorInvalidValue;
  if (t.n == 0) return hipSuccess;
  switch (grp) {                                                            // Bulletproofs' secp256k1 generators, a Groth16 key's statement points, a Pinocchio key's io points (G1 and G2)
    case G_SECP: hipLaunchKernelGGL(k_fixed_table<SpOps>, dim3((unsigned)t.n), dim3(64), 0, s, t); break;
    case G_G1: hipLaunchKernelGGL(k_fixed_table<FqOps>, dim3((unsigned)t.n), dim3(64), 0, s, t); break;
    case G_G2: hipLaunchKernelGGL(k_fixed_table<Fq2Ops>, dim3((unsigned)t.n), dim3(64), 0, s, t); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_fixed_muls_batch(int grp, const uint32_t* tables, const uint32_t* k, uint32_t* out, size_t n, int n_pts, hipStream_t s) {
  if ((grp != G_G1 && grp != G_G2) || n_pts < 1 || n * (size_t)n_pts >= (size_t(1) << 31)) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (grp == G_G1) hipLaunchKernelGGL(k_fixed_muls_batch<FqOps>, dim3((unsigned)(n * n_pts)), dim3(64), 0, s, tables, k, out, n, n_pts);
  else hipLaunchKernelGGL(k_fixed_muls_batch<Fq2Ops>, dim3((unsigned)(n * n_pts)), dim3(64), 0, s, tables, k, out, n, n_pts);
  return hipGetLastError();
}
// ---- statement sums of a large verification batch from 8-bit window tables (Groth16 verifier.rs:41-45) ---------------------------------------------------
// S_i = sum_j stmt[i][j] * U_j for 65,536 proofs is 196,608 scalar multiplications by the SAME n_stmt points: a 255-step chain each (1.2 M multiply-adds) when done
// as variable-base multiplications.  The key's cache entry instead carries, per statement point, the 32 x 256 Jacobian multiples d * 256^w * U_j (1.4 MB per point,
// built once per key: k_stmt_wide_tables); a proof's lane then adds one table entry per scalar byte — 32 n_stmt additions and ONE
// inversion for the affine S_i.  Entry 0 of a window is the point at infinity, which the complete addition absorbs.
static constexpr int WIDE_WINDOWS = 32, WIDE_ENTRIES = 256, WIDE_JW = 3 * FqC::N;      // words per Jacobian entry in the kernels' own limb form
__device__ inline void st_jac_words(uint32_t* p, const Jac<FqOps>& a) {
#pragma unroll
  for (int i = 0; i < FqC::N; ++i) { p[i] = a.X.v[i]; p[FqC::N + i] = a.Y.v[i]; p[2 * FqC::N + i] = a.Z.v[i]; }
}
__device__ inline Jac<FqOps> ld_jac_words(const uint32_t* p) {
  Jac<FqOps> a;
#pragma unroll
  for (int i = 0; i < FqC::N; ++i) { a.X.v[i] = p[i]; a.Y.v[i] = p[FqC::N + i]; a.Z.v[i] = p[2 * FqC::N + i]; }
  return a;
}
// One lane per (point, window, high nibble): 256^w U_j and 16 * 256^w U_j are entries 2w and 2w+1 of the point's 16^k table (k_fixed_table, built for the key's small
// batches anyway), so the lane forms hi * (16 * base) in <= 4 doublings and additions and walks its sixteen entries by mixed additions — a chain of <= 23 group operations
// where one lane per (point, window) ran 8w doublings and 255 additions (8.3 ms of the 21 a key's entry cost; profiles/r04_protocols_kernel_stats.csv).
__global__ void __launch_bounds__(64) k_stmt_wide_tables(const uint32_t* __restrict__ tab16, int n_pts, uint32_t* __restrict__ tables) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= n_pts * WIDE_WINDOWS * 16) return;
  const int hi = t & 15, w = (t >> 4) % WIDE_WINDOWS, j = (t >> 4) / WIDE_WINDOWS;
  const uint32_t* tj = tab16 + (size_t)j * 64 * ABI_G1_WORDS;
  const Aff<FqOps> base = PtIO<FqOps>::ld(tj + (size_t)(2 * w) * ABI_G1_WORDS), b16 = PtIO<FqOps>::ld(tj + (size_t)(2 * w + 1) * ABI_G1_WORDS);
  Jac<FqOps> acc = jac_inf<FqOps>();
  for (int b = 3; b >= 0; --b) { acc = jac_dbl(acc); if ((hi >> b) & 1) acc = jac_add_aff(acc, b16); }
  uint32_t* out = tables + (((size_t)j * WIDE_WINDOWS + w) * WIDE_ENTRIES + 16 * hi) * WIDE_JW;
  for (int lo = 0; lo < 16; ++lo) { st_jac_words(out + (size_t)lo * WIDE_JW, acc); acc = jac_add_aff(acc, base); }
}
__global__ void __launch_bounds__(64) k_stmt_sums_wide(const uint32_t* __restrict__ tables, const uint32_t* __restrict__ stmt, int n_stmt, uint32_t* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  Jac<FqOps> acc = jac_inf<FqOps>();
  for (int j = 0; j < n_stmt; ++j) {
    const uint32_t* k = stmt + (i * n_stmt + j) * 8;                  // canonical Fr, eight 32-bit words
    for (int w = 0; w < WIDE_WINDOWS; ++w) {
      const uint32_t d = (k[w >> 2] >> ((w & 3) * 8)) & 255u;
      if (d) acc = jac_add<FqOps>(acc, ld_jac_words(tables + (((size_t)j * WIDE_WINDOWS + w) * WIDE_ENTRIES + d) * WIDE_JW));
    }
  }
  PtIO<FqOps>::st(out + i * ABI_G1_WORDS, jac_to_aff(acc));
}
size_t stmt_wide_table_words(int n_pts) { return (size_t)n_pts * WIDE_WINDOWS * WIDE_ENTRIES * WIDE_JW; }
hipError_t launch_stmt_wide_tables(const uint32_t* tab16, int n_pts, uint32_t* tables, hipStream_t s) {      // tab16: the points' 64-entry tables of launch_fixed_tables, [point][64] affine
  if (n_pts < 1 || n_pts > 12 || !tab16) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_stmt_wide_tables, dim3((unsigned)(n_pts * WIDE_WINDOWS * 16 / 64)), dim3(64), 0, s, tab16, n_pts, tables);
  return hipGetLastError();
}
hipError_t launch_stmt_sums_wide(const uint32_t* tables, const uint32_t* stmt, int n_stmt, uint32_t* out, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_stmt_sums_wide, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, tables, stmt, n_stmt, out, n);
  return hipGetLastError();
}
hipError_t launch_fixed_muls(int grp, const FixedMuls& f, hipStream_t s) {
  if (grp != G_SECP || f.n < 0 || f.n > 16) return hipErrorInvalidValue;
  if (f.n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_fixed_muls<SpOps>, dim3((unsigned)f.n), dim3(64), 0, s, f);
  return hipGetLastError();
}

// out[j] = in[j][0] + ... + in[j][cnt-1] (zkt_internal.h: PointSums), one lane per output
template <class F>
__global__ void __launch_bounds__(64) k_point_sums(PointSums p) {
  const int j = threadIdx.x;
  if (j >= p.n) return;
  Jac<F> acc = jac_inf<F>();
  for (int k = 0; k < p.cnt[j]; ++k) acc = jac_add_aff(acc, PtIO<F>::ld(p.in[j][k]));
  PtIO<F>::st(p.out[j], jac_to_aff(acc));
}
hipError_t launch_point_sums(int grp, const PointSums& p, hipStream_t s) {
  if (grp != G_SECP || p.n < 0 || p.n > 12) return hipErrorInvalidValue;
  if (p.n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_point_sums<SpOps>, dim3(1), dim3(64), 0, s, p);
  return hipGetLastError();
}

// ---- sum of n affine points ------------------------------------------------------------------------------------------------
// the G2 / secp256k1 sums behind eval_with_g2_hidings (polynomial.rs:283-293) and (AffinePoints * PrimeFieldElems).sum()
// (secp256k1/affine_points.rs:25-31,123-144).  Two launches: every lane adds a strided share into a Jacobian accumulator (mixed
// additions, no inversion), a block folds its 64 accumulators through LDS, and one block folds the <= 64 block results and
// normalises.  (The earlier log2(n) rounds of affine pairwise additions cost one inversion latency per round.)
template <class F> struct JacRaw;
template <class C> struct JacRaw<PrimeOps<C>> {
  static constexpr int EW = C::N;
  __device__ static void st(uint32_t* p, const Fp<C>& a) { st_raw<C>(p, a); }
  __device__ static Fp<C> ld(const uint32_t* p) { return ld_raw<C>(p); }
};
template <> struct JacRaw<Fq2Ops> {
  static constexpr int EW = 2 * FqC::N;
  __device__ static void st(uint32_t* p, const Fq2& a) { st_raw<FqC>(p, a.c0); st_raw<FqC>(p + FqC::N, a.c1); }
  __device__ static Fq2 ld(const uint32_t* p) { Fq2 r; r.c0 = ld_raw<FqC>(p); r.c1 = ld_raw<FqC>(p + FqC::N); return r; }
};
template <class F> __device__ inline void st_jac(uint32_t* p, const Jac<F>& a) { constexpr int E = JacRaw<F>::EW; JacRaw<F>::st(p, a.X); JacRaw<F>::st(p + E, a.Y); JacRaw<F>::st(p + 2 * E, a.Z); }
template <class F> __device__ inline Jac<F> ld_jac(const uint32_t* p) { constexpr int E = JacRaw<F>::EW; return Jac<F>{JacRaw<F>::ld(p), JacRaw<F>::ld(p + E), JacRaw<F>::ld(p + 2 * E)}; }
template <class F> __device__ inline Jac<F> block_jac_sum(Jac<F> v, uint32_t* lds) {      // 64 lanes -> lane 0
  constexpr int JW = 3 * JacRaw<F>::EW;
  const int lane = threadIdx.x;
  for (int d = 32; d >= 1; d >>= 1) {
    if (lane >= d && lane < 2 * d) st_jac<F>(lds + (lane - d) * JW, v);
    __syncthreads();
    if (lane < d) v = jac_add(v, ld_jac<F>(lds + lane * JW));
    __syncthreads();
  }
  return v;
}
static constexpr int SUM_BLOCKS = 64;
template <class F>
__global__ void __launch_bounds__(64) k_group_sum_partials(const uint32_t* __restrict__ pts, size_t n, uint32_t* __restrict__ partials) {
  constexpr int W = PtIO<F>::WORDS, JW = 3 * JacRaw<F>::EW;
  __shared__ uint32_t lds[32 * JW];
  Jac<F> acc = jac_inf<F>();
  for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < n; i += (size_t)gridDim.x * 64) acc = jac_add_aff(acc, PtIO<F>::ld(pts + i * W));
  acc = block_jac_sum<F>(acc, lds);
  if (threadIdx.x == 0) st_jac<F>(partials + (size_t)blockIdx.x * JW, acc);
}
template <class F>
__global__ void __launch_bounds__(64) k_group_sum_finish(const uint32_t* __restrict__ partials, int count, uint32_t* __restrict__ out) {
  constexpr int JW = 3 * JacRaw<F>::EW;
  __shared__ uint32_t lds[32 * JW];
  Jac<F> acc = (int)threadIdx.x < count ? ld_jac<F>(partials + (size_t)threadIdx.x * JW) : jac_inf<F>();
  acc = block_jac_sum<F>(acc, lds);
  if (threadIdx.x == 0) PtIO<F>::st(out, jac_to_aff(acc));
}
template <class F> static hipError_t sum_t(uint32_t* pts, size_t n, uint32_t* scratch, hipStream_t s) {
  const int nb = (int)(n < (size_t)SUM_BLOCKS * 64 ? (n + 63) / 64 : SUM_BLOCKS);
  hipLaunchKernelGGL(k_group_sum_partials<F>, dim3(nb), dim3(64), 0, s, (const uint32_t*)pts, n, scratch);
  hipLaunchKernelGGL(k_group_sum_finish<F>, dim3(1), dim3(64), 0, s, (const uint32_t*)scratch, nb, pts);     // result in pts[0]
  return hipGetLastError();
}
// `pts` is clobbered (result in pts[0]); `scratch` holds SUM_BLOCKS Jacobian partials (zkt_group_sum_scratch_words()).
size_t group_sum_scratch_words(int grp) { return (size_t)SUM_BLOCKS * 3 * (grp == G_G1 ? FqC::N : grp == G_G2 ? 2 * FqC::N : SpC::N); }
hipError_t launch_group_sum_inplace(int grp, uint32_t* pts, size_t n, hipStream_t s) {
  if (n <= 1) return hipSuccess;
  // block partials: one small buffer per calling thread (the ABI is thread-safe; a thread issues its sums on one stream, in order)
  static thread_local uint32_t* scratch = nullptr;
  if (!scratch && hipMalloc((void**)&scratch, group_sum_scratch_words(G_G2) * 4) != hipSuccess) return hipErrorOutOfMemory;
  switch (grp) { case G_G1: return sum_t<FqOps>(pts, n, scratch, s); case G_G2: return sum_t<Fq2Ops>(pts, n, scratch, s); case G_SECP: return sum_t<SpOps>(pts, n, scratch, s); }
  return hipErrorInvalidValue;
}

}  // namespace zkt
