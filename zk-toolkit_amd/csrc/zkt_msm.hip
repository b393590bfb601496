// G1 multi-scalar multiplication  sum_i s_i * P_i  for gfx950 (row a9 of SURVEY §8).
//   Polynomial::eval_with_g1_hidings   src/building_block/field/polynomial.rs:271-281
// The reference runs n double-and-add scalar multiplications and n affine additions
// strictly sequentially.  The group element is the same whatever the summation order,
// and the result leaves as a canonical affine point, so this is bit-compatible.
//
// MI355X-first design:
//  * bases are device-resident (a CRS is uploaded once and reused by every proof) and,
//    because HBM is 288 GB, every base is stored together with its window multiples
//    2^(c*w) * P_i  (nwin * n affine points, 112 B each for G1).  All windows then share ONE set
//    of 2^(c-1) buckets: there is no per-window bucket reduction and no final Horner
//    chain of 256 serial doublings — on this machine a serial Fq multiply costs ~1.1 us
//    per lane, so serial chains, not FLOPs, are what must be designed away.
//  * signed c-bit digits (scalars used as-is, 256 bits: macros.rs:10-21), counting sort of
//    (bucket, point) pairs, one bucket per lane accumulating in XYZZ coordinates with
//    complete mixed additions (P+P, P+(-P), infinity: macros.rs:43-63), gather of 112-byte
//    affine points, then a two-level bucket reduction built from short trees.
//  * one-shot calls with host pointers use the DIRECT form instead (msm_plan_direct): no table, every window owns its
//    buckets, all windows are reduced side by side (grid.y) and joined by one wave — the serial doubling chain is paid once
//    per call instead of a table build that costs 40x a resident MSM.
#include <cstdlib>
#include "abi.h"
#include "zkt_internal.h"

namespace zkt {

// The pipeline is generic over the group: G1 over Fq (FqOps), G2 over Fq2 (Fq2Ops, row f-1: eval_with_g2_hidings,
// polynomial.rs:283-293) and secp256k1 (SpOps: (AffinePoints * PrimeFieldElems).sum(), secp256k1/affine_points.rs:25-31,123-144).
// Coordinates are stored raw (Montgomery) with CW words each: affine point = 2*CW, XYZZ = 4*CW, Jacobian partial = 3*CW.
// Kernels of the sort and reduce stages run beside the VALU-saturating accumulate waves of a neighbouring MSM; raise
// their wave priority so their (few, latency-bound) instructions are not queued behind them.
#ifndef ZKT_SIDE_PRIO
#define ZKT_SIDE_PRIO __builtin_amdgcn_s_setprio(3)
#endif
template <class F> struct Coord;
template <class C> struct Coord<PrimeOps<C>> {
  static constexpr int CW = C::N;
  __device__ static Fp<C> ld(const uint32_t* p) { return ld_raw<C>(p); }
  __device__ static void st(uint32_t* p, const Fp<C>& a) { st_raw<C>(p, a); }
};
template <> struct Coord<Fq2Ops> {
  static constexpr int CW = 2 * FqC::N;
  __device__ static Fq2 ld(const uint32_t* p) { Fq2 r; r.c0 = ld_raw<FqC>(p); r.c1 = ld_raw<FqC>(p + FqC::N); return r; }
  __device__ static void st(uint32_t* p, const Fq2& a) { st_raw<FqC>(p, a.c0); st_raw<FqC>(p + FqC::N, a.c1); }
};
template <class F> __device__ inline Xyzz<F> ld_xy(const uint32_t* p) {
  constexpr int CW = Coord<F>::CW; Xyzz<F> r;
  r.X = Coord<F>::ld(p); r.Y = Coord<F>::ld(p + CW); r.ZZ = Coord<F>::ld(p + 2 * CW); r.ZZZ = Coord<F>::ld(p + 3 * CW); return r;
}
template <class F> __device__ inline void st_xy(uint32_t* p, const Xyzz<F>& a) {
  constexpr int CW = Coord<F>::CW;
  Coord<F>::st(p, a.X); Coord<F>::st(p + CW, a.Y); Coord<F>::st(p + 2 * CW, a.ZZ); Coord<F>::st(p + 3 * CW, a.ZZZ);
}
static int coord_words(int grp) { return grp == G_G1 ? FqC::N : grp == G_G2 ? 2 * FqC::N : SpC::N; }
// dimensions of the atomic-free partition sort (k_part_*, below): partitions of PART_SUB buckets, tiles of PART_TILE scalars, chunks of PART_CHUNK records
static constexpr int PART_LO = 9, PART_SUB = 1 << PART_LO, PART_TPB = 256, PART_TILE = 4 * PART_TPB, PART_CHUNK = 8192, PART_MAXP = 2048;
struct PartDims { uint32_t P, ntiles, maxblk; };
static PartDims part_dims(size_t n, size_t nbuckets, int nwin) {
  PartDims d; d.P = (uint32_t)((nbuckets + PART_SUB - 1) / PART_SUB); d.ntiles = (uint32_t)((n + PART_TILE - 1) / PART_TILE);
  d.maxblk = d.P + (uint32_t)((size_t)nwin * n / PART_CHUNK) + 1; return d;
}
static size_t part_ws_bytes(size_t n, size_t nbuckets, int nwin) {
  const PartDims d = part_dims(n, nbuckets, nwin);
  return 1024 + (size_t)nwin * n * 8 + 256 + 2 * ((size_t)d.P * d.ntiles + 1) * 4 + ((size_t)d.P + 1) * 4 + ((size_t)d.P * d.ntiles / 2048 + 2) * 4 + 256 + (size_t)d.maxblk * PART_SUB * 4;
}

// ---------------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------------
#if defined(ZKT_MSM_PART_G1)
static constexpr uint32_t MSM_CHUNK_MAX = 128;
static uint32_t pick_chunk(size_t entries, int grp);
MsmPlan msm_plan(size_t n, int grp) {
  MsmPlan p; p.n = n; p.grp = grp;
  const size_t XYW = 4 * (size_t)coord_words(grp);
  int lg = 0; while ((size_t(1) << (lg + 1)) <= (n ? n : 1)) ++lg;   // floor(log2 n)
  // Window width.  2^20 terms: c = 20, 2^19 buckets ~ n/2, ~2*nwin ~ 26 points per bucket.  Below 2^19 terms the MSM is latency-bound and the width is MEASURED, not derived
  // (profiles/r04_msm_window_sweep.txt: single and pipelined time of resident G1 and G2 MSMs over widths and sizes 2^4 .. 2^19).  Round 3's c = floor(log2 n) met widths whose
  // TOP window holds a few bits of a 255-bit scalar (256 + c is not a multiple of c: the last window is a remainder; three bits at c = 12, 14, 18): every scalar then lands in one
  // of 8-16 NEIGHBOURING buckets of that window, which (i) one tile of k_merge_partials summed one after the other (0.84 ms of a 1.6 ms MSM at 2^14 terms — fixed there: tiles
  // are strided now) and (ii) serialise the counting sort's atomics on 8-16 counters (k_digits 1.2-1.5 ms at 2^18 terms, c = 18).  The plan picks widths that measured well on
  // both counts: 16 from 2^14 terms on (windows end exactly at bit 256: no remainder window), 13 / 11 / 10 below.  c = 17 -> 16 at 2^17 terms: 1.24 -> 1.14 ms single,
  // 2^18 terms 2.90 -> 1.62 ms, G2 at 2^18 terms 6.2 -> 4.2 ms, a rank's share of a sharded Groth16 proof 6.3 -> 5.1 ms; 2^19 terms: c = 20 as well (2.17 / 1.45 ms single /
  // pipelined against 3.27 / 1.69 at c = 19).
  static const int forced_c = [] { const char* e = getenv("ZKT_MSM_C"); return e ? atoi(e) : 0; }();
  int c = lg >= 19 ? 20 : lg >= 14 ? 16 : lg >= 11 ? 13 : lg == 10 ? 11 : 10;      // (2^19 terms: c = 20 as well — 2.17 / 1.45 ms single / pipelined against 3.27 / 1.69 at c = 19)
  if (forced_c) c = forced_c;
  if (c < 4) c = 4;
  if (c > 20) c = 20;
  p.c = c; p.nwin = (256 + 1 + c - 1) / c; p.nbuckets = size_t(1) << (c - 1);
  size_t ent = (size_t)p.nwin * n;
  // G2: pair-tree rounds in affine coordinates ahead of the XYZZ accumulate (zkt_msm_affine.hip).  OFF by default: measured inside the product the rounds LOSE to the
  // lane-pair XYZZ kernel (2^20-term pipelined G2 MSM 8.06 ms without, 8.75 / 9.05 / 9.59 ms with 1 / 2 / 3 rounds; profiles/r04_batched_affine_go_no_go.md) — kept as a
  // tested alternative behind ZKT_G2_AFFINE_ROUNDS (1..4) and ZKT_G2_AFFINE_MIN_ENTRIES for A/B measurements on other shapes.
  static const int aff_rounds_cfg = [] { const char* e = getenv("ZKT_G2_AFFINE_ROUNDS"); int r = e ? atoi(e) : 0; return r < 0 ? 0 : r > MSM_AFFINE_MAX_ROUNDS ? MSM_AFFINE_MAX_ROUNDS : r; }();
  static const size_t aff_min_cfg = [] { const char* e = getenv("ZKT_G2_AFFINE_MIN_ENTRIES"); return e ? (size_t)strtoull(e, nullptr, 10) : (size_t(1) << 22); }();
  p.aff_rounds = (grp == G_G2 && ent >= aff_min_cfg) ? aff_rounds_cfg : 0;
  p.aff_off = 0;
  p.chunk = pick_chunk(ent >> p.aff_rounds, grp);
  size_t b = 0;
  b += (p.nbuckets + 1) * 4 * 3;          // counts, offsets, cursor
  b += 2 * ent * 4 + 256;                  // entries, slots
  b += p.nbuckets * XYW * 4;               // bucket sums
  b += (2048 + 64 * 4) * XYW * 4;          // row/col sums, bit classes (WB_SPLIT = 4 parts each)
  b += (1024 + 3 * 1025) * 4 + 2 * (p.nbuckets + 1) * 4;          // scan scratch, size bins, task counts/offsets
  b += (p.nbuckets + ent / p.chunk + 2) * (8 + XYW * 4);          // task list + partial sums of split buckets
  b += 64 * 64 * XYW * 4 + 1024;                                  // block sums of hot buckets (k_merge_hot), hot list
  b += part_ws_bytes(n, p.nbuckets, p.nwin);
  b += 4096;
  if (p.aff_rounds) { b = (b + 255) & ~(size_t)255; p.aff_off = b; b += msm_affine_ws_bytes(ent, p.nbuckets, p.aff_rounds, coord_words(grp)) + 512; }
  p.ws_bytes = b;
  p.direct = 0; p.half = p.nbuckets;
  return p;
}
// Task granularity (entries per lane-task of the accumulate kernel) as a function of the problem: large MSMs keep long chains (128: the
// partial sums of split buckets cost a full addition each), small ones are cut until the grid fills the chip — at 2^17 terms a chunk of 128
// left 65,536 one-bucket tasks = ONE wave per SIMD, each lane a serial chain of 32-55 additions at the single-wave multiply-add rate
// (1.37 ms for an eighth of the work of a 2^20-term MSM, profiles/r02_msm_latency_kernel_trace.txt).
static uint32_t pick_chunk(size_t entries, int grp) {
  static const int forced = [] { const char* e = getenv("ZKT_MSM_CHUNK"); return e ? atoi(e) : 0; }();
  if (forced >= 2 && forced <= (int)MSM_CHUNK_MAX) return (uint32_t)forced;
  // lanes to fill: 256 CUs x 4 SIMDs x 2 waves x 64 lanes (G1, secp256k1), x 1.5 so that the short tail tasks have something to hide under;
  // the G2 pair kernel runs one wave per SIMD and two lanes per task
  const size_t tasks = grp == G_G2 ? (size_t)65536 : (size_t)196608;
  size_t c = (entries + tasks - 1) / tasks;
  if (c < 8) c = 8;
  if (c > MSM_CHUNK_MAX) c = MSM_CHUNK_MAX;
  return (uint32_t)c;
}
MsmPlan msm_plan_direct(size_t n, int grp) {
  MsmPlan p; p.n = n; p.grp = grp;
  const size_t XYW = 4 * (size_t)coord_words(grp);
  int lg = 0; while ((size_t(1) << (lg + 1)) <= (n ? n : 1)) ++lg;
  int c = lg - 3;                   // every window has its own buckets: fewer, fuller buckets than the shared form
  if (c < 9) c = 9;                 // nwin <= 29: k_join_windows folds at most 32 windows in one wave
  if (c > 16) c = 16;
  p.c = c; p.nwin = (256 + 1 + c - 1) / c; p.half = size_t(1) << (c - 1); p.nbuckets = (size_t)p.nwin * p.half; p.direct = 1;
  p.aff_rounds = 0; p.aff_off = 0;
  size_t ent = (size_t)p.nwin * n;
  p.chunk = pick_chunk(ent, grp);
  size_t b = 0;
  b += (p.nbuckets + 1) * 4 * 3 + 2 * ent * 4 + 256 + p.nbuckets * XYW * 4;
  b += (size_t)p.nwin * (2048 + 64 * 4 + 1) * XYW * 4;           // per-window row/col sums, bit classes (4 parts each), window results
  b += (1024 + 3 * 1025) * 4 + 2 * (p.nbuckets + 1) * 4;
  b += (p.nbuckets + ent / p.chunk + 2) * (8 + XYW * 4);
  b += 64 * 64 * XYW * 4 + 1024;
  b += part_ws_bytes(n, p.nbuckets, p.nwin);
  b += 8192;
  p.ws_bytes = b;
  return p;
}
#endif

// ---------------------------------------------------------------------------------
// layout conversion and window-multiple precomputation
// ---------------------------------------------------------------------------------
template <class F>
__global__ void __launch_bounds__(64) k_to_kernel_layout(const uint32_t* __restrict__ abi, uint32_t* __restrict__ mont,
                                                         uint8_t* __restrict__ inf, size_t n) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  constexpr int CW = Coord<F>::CW;
  Aff<F> p = PtIO<F>::ld(abi + i * PtIO<F>::WORDS);
  Coord<F>::st(mont + i * 2 * CW, p.x); Coord<F>::st(mont + i * 2 * CW + CW, p.y);
  inf[i] = p.inf ? 1 : 0;
}
// This file is compiled three times (Makefile): -DZKT_MSM_PART_G1 -DZKT_INLINE_MUL instantiates the G1 kernels with the field
// multiply inlined (the headline path) and defines the public launch_msm_* dispatchers; -DZKT_MSM_PART_SECP -DZKT_INLINE_MUL does the
// same for secp256k1; -DZKT_MSM_PART_OTHER instantiates G2 with a called multiply (an inlined Fq2 XYZZ add would be ~200 KB of code
// and minutes of compile time).
#if defined(ZKT_MSM_PART_G1)
#define MSM_DISPATCH(grp, CALL) switch (grp) { case G_G1: { typedef FqOps F; CALL; } break; default: return hipErrorInvalidValue; }
#define PART(x) x##_g1
#elif defined(ZKT_MSM_PART_SECP)     // secp256k1, multiply inlined like G1 (8-limb field: small code; a called multiply is latency bound at the low
#define MSM_DISPATCH(grp, CALL) switch (grp) { case G_SECP: { typedef SpOps F; CALL; } break; default: return hipErrorInvalidValue; }   // occupancy of 2^17-term MSMs)
#define PART(x) x##_secp
#else
#define MSM_DISPATCH(grp, CALL) switch (grp) { case G_G2: { typedef Fq2Ops F; CALL; } break; default: return hipErrorInvalidValue; }
#define PART(x) x##_other
#endif
hipError_t PART(launch_msm_to_kernel_layout)(int grp, const uint32_t* abi, uint32_t* mont, uint8_t* inf, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  MSM_DISPATCH(grp, hipLaunchKernelGGL(k_to_kernel_layout<F>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, abi, mont, inf, n));
  return hipGetLastError();
}

// table[w*n + i] = 2^(c*w) * P_i (affine, Montgomery); infinity flags likewise.
// One lane walks one point through all windows in Jacobian coordinates (255 doublings) and normalises the nwin-1 results with ONE
// inversion (Montgomery's trick along the lane's own chain): X_w, Y_w wait in their table slots, Z_w and prod_{v<w} Z_v in `tmp`
// (two coordinates per entry, freed after the build).  An inversion per window made the build 3x longer than the doublings alone.
template <class F>
__global__ void __launch_bounds__(64) k_precompute(uint32_t* __restrict__ table, uint8_t* __restrict__ inf, size_t n, int c, int nwin, uint32_t* __restrict__ tmp) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n || nwin < 2) return;
  constexpr int CW = Coord<F>::CW, PW = 2 * CW;
  typedef typename F::E E;
  Aff<F> a; a.x = Coord<F>::ld(table + i * PW); a.y = Coord<F>::ld(table + i * PW + CW); a.inf = inf[i] != 0;
  Jac<F> j = jac_from_aff(a);
  E acc = F::one();
  for (int w = 1; w < nwin; ++w) {
    for (int d = 0; d < c; ++d) j = jac_dbl(j);
    const size_t o = (size_t)w * n + i, t = (size_t)(w - 1) * n + i;
    const bool isinf = jac_is_inf(j);
    const E z = isinf ? F::one() : j.Z;
    Coord<F>::st(table + o * PW, j.X); Coord<F>::st(table + o * PW + CW, j.Y);
    Coord<F>::st(tmp + t * PW, z); Coord<F>::st(tmp + t * PW + CW, acc);
    inf[o] = isinf ? 1 : 0;
    acc = F::mul(acc, z);
  }
  E inv = F::inv(acc);                                      // 1 / (Z_1 ... Z_{nwin-1}), no factor is zero
  for (int w = nwin - 1; w >= 1; --w) {
    const size_t o = (size_t)w * n + i, t = (size_t)(w - 1) * n + i;
    const E z = Coord<F>::ld(tmp + t * PW), pre = Coord<F>::ld(tmp + t * PW + CW);
    const E zi = F::mul(inv, pre);                          // 1 / Z_w
    inv = F::mul(inv, z);
    const E zi2 = F::sqr(zi);
    E x = F::mul(Coord<F>::ld(table + o * PW), zi2), y = F::mul(Coord<F>::ld(table + o * PW + CW), F::mul(zi2, zi));
    if (inf[o]) { x = F::zero(); y = F::zero(); }
    Coord<F>::st(table + o * PW, x); Coord<F>::st(table + o * PW + CW, y);
  }
}

// ---------------------------------------------------------------------------------
// signed-digit decomposition, counting sort by bucket
// ---------------------------------------------------------------------------------
__device__ inline uint32_t window_bits(const uint32_t* k, int w, int c) {
  int o = w * c, word = o >> 5, sh = o & 31;
  if (word >= 8) return 0;
  uint64_t v = k[word];
  if (word + 1 < 8) v |= (uint64_t)k[word + 1] << 32;
  return (uint32_t)(v >> sh) & ((1u << c) - 1);
}

// One atomic per (scalar, window) on the bucket counter; the returned value is the entry's slot inside its bucket.  The COUNT pass keeps it
// (slot[w*n + i], a coalesced 4-byte store per window), so the SCATTER pass is atomic-free: recompute the digit, read slot and offsets[bucket],
// write the entry (13.6 M atomics at 2^20 terms were 0.5 ms of the scatter pass).
// Skewed inputs (a witness full of 0/1) send most lanes of a wave to the SAME counter, which would serialise a million
// atomics on one address: up to two rounds of wave-level aggregation elect a leader for the most common bucket id among
// the active lanes (one atomicAdd of the population count, ranks by prefix popcount); the rest go individually.
static constexpr uint32_t NO_SLOT = 0xffffffffu;          // zero digit / infinity: no entry
template <bool SCATTER>
static __global__ void __launch_bounds__(256) k_digits(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf, size_t n, int c, int nwin, uint32_t win_buckets,
                                                uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets,
                                                uint32_t* __restrict__ slot, uint32_t* __restrict__ entries) {
  ZKT_SIDE_PRIO;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < n;
  uint32_t k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (live) {
    const uint4* sp = reinterpret_cast<const uint4*>(scalars + i * 8);
    uint4 lo = sp[0], hi = sp[1];
    k[0] = lo.x; k[1] = lo.y; k[2] = lo.z; k[3] = lo.w; k[4] = hi.x; k[5] = hi.y; k[6] = hi.z; k[7] = hi.w;
  }
  const uint32_t half = 1u << (c - 1);
  const unsigned lane = threadIdx.x & 63;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  uint32_t carry = 0;
  for (int w = 0; w < nwin; ++w) {                      // wave-uniform trip count: the ballots below need every lane here
    uint32_t raw = window_bits(k, w, c) + carry;
    uint32_t neg = raw > half;
    uint32_t mag = neg ? (1u << c) - raw : raw;         // |digit| in [0, 2^(c-1)]
    carry = neg;
    // resident form (win_buckets = 0): entry = index into the window-multiple table, one bucket set for all windows;
    // direct form: entry = the base itself, window w owns buckets [w * win_buckets, (w+1) * win_buckets)
    size_t src = win_buckets ? i : (size_t)w * n + i;
    const uint32_t b = mag - 1 + (uint32_t)w * win_buckets;
    if (SCATTER) {
      if (live) { const uint32_t pos = slot[(size_t)w * n + i]; if (pos != NO_SLOT) entries[offsets[b] + pos] = (uint32_t)src | (neg << 31); }
      continue;
    }
    bool todo = live && mag != 0 && !inf[live ? src : 0];   // infinity and zero digits contribute nothing
    uint32_t pos = NO_SLOT;
    for (int round = 0; round < 2; ++round) {
      unsigned long long act = __ballot(todo);
      if (act == 0) break;
      int leader = __ffsll((long long)act) - 1;
      uint32_t lb = __shfl(b, leader);
      unsigned long long same = __ballot(todo && b == lb);
      int cnt = __popcll(same);
      if (cnt < 4) break;                               // nothing worth aggregating: fall through to individual atomics
      uint32_t base = 0;
      if ((int)lane == leader) base = atomicAdd(&counts[lb], (uint32_t)cnt);
      base = __shfl(base, leader);
      if (todo && b == lb) { pos = base + (uint32_t)__popcll(same & lt_mask); todo = false; }
    }
    if (todo) pos = atomicAdd(&counts[b], 1u);
    if (live) slot[(size_t)w * n + i] = pos;
  }
}

// exclusive scan of counts[0..m) into offsets[0..m]: per-block sums, scan of the block sums, per-block scan.
static constexpr int SCAN_TPB = 256, SCAN_ITEMS = 8, SCAN_TILE = SCAN_TPB * SCAN_ITEMS;
__device__ inline uint32_t block_excl_scan_256(uint32_t v, uint32_t* lds /*256*/, uint32_t& total) {
  const int t = threadIdx.x;
  lds[t] = v; __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    uint32_t x = t >= d ? lds[t - d] : 0; __syncthreads();
    lds[t] += x; __syncthreads();
  }
  total = lds[255];
  uint32_t excl = t ? lds[t - 1] : 0;
  __syncthreads();
  return excl;
}
static __global__ void __launch_bounds__(256) k_scan_sums(const uint32_t* __restrict__ in, size_t m, uint32_t* __restrict__ blocksum) {
  ZKT_SIDE_PRIO;
  __shared__ uint32_t lds[256];
  size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < m) s += in[base + k];
  uint32_t tot; (void)block_excl_scan_256(s, lds, tot);
  if (threadIdx.x == 0) blocksum[blockIdx.x] = tot;
}
static __global__ void __launch_bounds__(256) k_scan_top(uint32_t* __restrict__ blocksum, int nblk, uint32_t* __restrict__ grand_total) {
  ZKT_SIDE_PRIO;
  __shared__ uint32_t lds[256];
  uint32_t run = 0;
  for (int base = 0; base < nblk; base += 256) {          // nblk <= 2^19/2048 = 256 in practice
    int i = base + threadIdx.x;
    uint32_t v = i < nblk ? blocksum[i] : 0, tot;
    uint32_t ex = block_excl_scan_256(v, lds, tot);
    if (i < nblk) blocksum[i] = run + ex;
    run += tot;
  }
  if (threadIdx.x == 0) *grand_total = run;
}
static __global__ void __launch_bounds__(256) k_scan_final(const uint32_t* __restrict__ in, size_t m, const uint32_t* __restrict__ blocksum,
                                                    uint32_t* __restrict__ out) {
  ZKT_SIDE_PRIO;
  __shared__ uint32_t lds[256];
  size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS], s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) { v[k] = base + k < m ? in[base + k] : 0; s += v[k]; }
  uint32_t tot; uint32_t run = block_excl_scan_256(s, lds, tot) + blocksum[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) { if (base + k < m) out[base + k] = run; run += v[k]; }
}
// (Round 4 measured ONE block scanning a small set's counters — and the task count fused into that scan — to save kernel nodes of a small MSM's graph: the one-block scan of
//  65,536 counters takes 40 us against 18 us for the three kernels below plus their launch gaps, and the 2^17-term MSM, the range proof and the small Groth16 keys did not move
//  (1.28 ms, 3.3-3.6 ms, 2.5 ms).  Taken out again: a node of a replayed graph does not cost what DESIGN §9 (round 3) assumed.)
// out[0..m) = exclusive scan of in, out[m] = total.  scratch: >= ceil(m/2048) words
static void launch_scan(const uint32_t* in, uint32_t* out, size_t m, uint32_t* scratch, hipStream_t s) {
  int nblk = (int)((m + SCAN_TILE - 1) / SCAN_TILE);
  hipLaunchKernelGGL(k_scan_sums, dim3(nblk), dim3(256), 0, s, in, m, scratch);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(256), 0, s, scratch, nblk, out + m);
  hipLaunchKernelGGL(k_scan_final, dim3(nblk), dim3(256), 0, s, in, m, (const uint32_t*)scratch, out);
}

// ---------------------------------------------------------------------------------
// counting sort by bucket WITHOUT global atomics (large MSMs): two-level radix partition through LDS histograms
// ---------------------------------------------------------------------------------
// k_digits pays one returning global atomic per (scalar, window) — 13.6 M of them at 2^20 terms, executed at the memory side at ~25 G/s whatever the
// occupancy: 0.55 ms for the count pass plus 0.22-0.35 ms for the scatter, and two sorts side by side in a Groth16 proof slow each other to 1.9 + 2.6 ms.
// Here the bucket id is split into a partition (its high bits) and a position inside the partition (its low PART_LO bits):
//   pass A  every tile of PART_TILE scalars (4 per thread of a 256-thread block: 1,024 blocks at 2^20 terms — with 16 per thread the pass had one block per CU and took
//           140 us instead of 57; a block small enough to find room beside the accumulate waves of a neighbouring MSM — 1024-thread blocks waited 1.8 ms for a
//           free CU in the pipeline) histograms its digits by PARTITION in LDS                 -> tilehist[partition][tile]
//   scan    (partition-major) gives every (partition, tile) its segment of the record array
//   pass B  the tiles recompute their digits and write (entry, low bits) records into their segments     (ranks from LDS atomics; ONE 8-byte store per record:
//           the pass is bound by the NUMBER of scattered stores — 316 us with a 4-byte and a 2-byte store per record, 180 us with one)
//   pass C  every partition is cut into chunks of PART_CHUNK records; a block histograms the low bits of its chunk in LDS (C1), one block per partition
//           turns the chunk histograms into running offsets and the bucket totals into counts[] / offsets[] (C1b), and the chunk blocks place their
//           entries (C2).  A skewed input (a witness of 0/1: a million entries in one partition) is simply more chunks.
// (Round 3 also measured both scatter passes with their runs staged through LDS — records grouped by partition / bucket in a 48-56 KB stage and written out in order, to
//  cure the 7x write amplification the counters show (592 + 369 MB written for 82 + 54 MB of payload).  Pass C2 took the same 165 us staged as direct, pass B 893 us
//  instead of 313 (its 4096-scalar tile only fits the stage in eight partition ranges, each re-deriving the digits): the partial-sector writes are absorbed by the memory
//  side, the passes are bound by their LDS atomics and latency.  Taken out again.)
// Every pass streams: ~310 MB at 2^20 terms instead of 13.6 M scattered read-modify-writes.  counts / offsets / entries come out exactly as from k_digits
// (the order of the entries inside a bucket is arbitrary in both), so everything downstream is unchanged.
template <bool SCATTER>
static __global__ void __launch_bounds__(PART_TPB) k_part_tiles(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf, size_t n, int c, int nwin, uint32_t win_buckets,
                                                    uint32_t P, uint32_t ntiles, uint32_t* __restrict__ tilehist, const uint32_t* __restrict__ tileoff,
                                                    uint2* __restrict__ rec) {
  ZKT_SIDE_PRIO;
  __shared__ uint32_t h[PART_MAXP];
  for (uint32_t p = threadIdx.x; p < P; p += PART_TPB) h[p] = SCATTER ? tileoff[(size_t)p * ntiles + blockIdx.x] : 0u;
  __syncthreads();
  const uint32_t half = 1u << (c - 1);
  // two scalars per trip: their loads, and the LDS atomic -> store chains of their digits, are in flight together (one scalar at a time the pass is a latency chain)
  for (int j = 0; j < PART_TILE / PART_TPB; j += 2) {
    const size_t i0 = (size_t)blockIdx.x * PART_TILE + j * PART_TPB + threadIdx.x, i1 = i0 + PART_TPB;
    const bool v0 = i0 < n, v1 = i1 < n;
    uint32_t k0[8] = {}, k1[8] = {};
    if (v0) { const uint4* sp = reinterpret_cast<const uint4*>(scalars + i0 * 8); const uint4 lo = sp[0], hi = sp[1];
      k0[0] = lo.x; k0[1] = lo.y; k0[2] = lo.z; k0[3] = lo.w; k0[4] = hi.x; k0[5] = hi.y; k0[6] = hi.z; k0[7] = hi.w; }
    if (v1) { const uint4* sp = reinterpret_cast<const uint4*>(scalars + i1 * 8); const uint4 lo = sp[0], hi = sp[1];
      k1[0] = lo.x; k1[1] = lo.y; k1[2] = lo.z; k1[3] = lo.w; k1[4] = hi.x; k1[5] = hi.y; k1[6] = hi.z; k1[7] = hi.w; }
    uint32_t carry0 = 0, carry1 = 0;
    for (int w = 0; w < nwin; ++w) {
      const uint32_t raw0 = window_bits(k0, w, c) + carry0, raw1 = window_bits(k1, w, c) + carry1;
      const uint32_t neg0 = raw0 > half, neg1 = raw1 > half;
      const uint32_t mag0 = neg0 ? (1u << c) - raw0 : raw0, mag1 = neg1 ? (1u << c) - raw1 : raw1;
      carry0 = neg0; carry1 = neg1;
      const size_t src0 = win_buckets ? i0 : (size_t)w * n + i0, src1 = win_buckets ? i1 : (size_t)w * n + i1;
      const bool a0 = v0 && mag0 != 0 && !inf[src0], a1 = v1 && mag1 != 0 && !inf[src1];       // infinity and zero digits contribute nothing
      const uint32_t b0 = mag0 - 1 + (uint32_t)w * win_buckets, b1 = mag1 - 1 + (uint32_t)w * win_buckets;
      uint32_t pos0 = 0, pos1 = 0;
      if (a0) pos0 = atomicAdd(&h[b0 >> PART_LO], 1u);
      if (a1) pos1 = atomicAdd(&h[b1 >> PART_LO], 1u);
      if (SCATTER) {
        // ONE 8-byte store per record (entry, low bits): the pass is bound by the number of scattered stores, not by their bytes (two stores per record: 285 us)
        if (a0) rec[pos0] = make_uint2((uint32_t)src0 | (neg0 << 31), b0 & (PART_SUB - 1));
        if (a1) rec[pos1] = make_uint2((uint32_t)src1 | (neg1 << 31), b1 & (PART_SUB - 1));
      }
    }
  }
  if (!SCATTER) {
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < P; p += PART_TPB) tilehist[(size_t)p * ntiles + blockIdx.x] = h[p];
  }
}
// chunk blocks of every partition: blkoff[p] = first block of partition p, blkoff[P] = blocks in use (one block: P <= PART_MAXP)
static __global__ void __launch_bounds__(256) k_part_blocks(const uint32_t* __restrict__ tileoff, uint32_t P, uint32_t ntiles, uint32_t* __restrict__ blkoff) {
  ZKT_SIDE_PRIO;
  __shared__ uint32_t lds[256];
  constexpr int ITEMS = PART_MAXP / 256;
  uint32_t v[ITEMS], sum = 0;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const uint32_t p = threadIdx.x * ITEMS + k;
    uint32_t cnt = 0;
    if (p < P) { const uint32_t size = tileoff[(size_t)(p + 1) * ntiles] - tileoff[(size_t)p * ntiles]; cnt = (size + PART_CHUNK - 1) / PART_CHUNK; }
    v[k] = cnt; sum += cnt;
  }
  uint32_t tot; uint32_t run = block_excl_scan_256(sum, lds, tot);
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) { const uint32_t p = threadIdx.x * ITEMS + k; if (p < P) blkoff[p] = run; run += v[k]; }
  if (threadIdx.x == 0) blkoff[P] = tot;
}
// which (partition, chunk) block `blk` is: the last p with blkoff[p] <= blk
__device__ inline uint32_t part_of_block(const uint32_t* __restrict__ blkoff, uint32_t P, uint32_t blk) {
  uint32_t lo = 0, hi = P;                          // invariant: blkoff[lo] <= blk < blkoff[hi]
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (blkoff[mid] <= blk) lo = mid; else hi = mid; }
  return lo;
}
template <bool PLACE>
static __global__ void __launch_bounds__(256) k_part_chunks(const uint32_t* __restrict__ tileoff, uint32_t P, uint32_t ntiles, const uint32_t* __restrict__ blkoff,
                                                     const uint2* __restrict__ rec, uint32_t* __restrict__ subhist,
                                                     const uint32_t* __restrict__ offsets, size_t nbuckets, uint32_t* __restrict__ entries) {
  ZKT_SIDE_PRIO;
  __shared__ uint32_t h[PART_SUB];
  const uint32_t blk = blockIdx.x;
  if (blk >= blkoff[P]) return;                     // block-uniform
  const uint32_t p = part_of_block(blkoff, P, blk), chunk = blk - blkoff[p];
  const uint32_t pbeg = tileoff[(size_t)p * ntiles], pend = tileoff[(size_t)(p + 1) * ntiles];
  const uint32_t beg = pbeg + chunk * PART_CHUNK, end = beg + PART_CHUNK < pend ? beg + PART_CHUNK : pend;
  for (int t = threadIdx.x; t < PART_SUB; t += 256) {
    const size_t b = (size_t)p * PART_SUB + t;
    h[t] = PLACE ? (b < nbuckets ? offsets[b] + subhist[(size_t)blk * PART_SUB + t] : 0u) : 0u;
  }
  __syncthreads();
  // four records per lane and trip: the loads of a trip are in flight together (the loop is a chain of load -> LDS atomic -> store otherwise: 165 us for 136 MB)
  uint32_t r = beg + threadIdx.x;
  for (; r + 3 * 256 < end; r += 4 * 256) {
    const uint2 r0 = rec[r], r1 = rec[r + 256], r2 = rec[r + 512], r3 = rec[r + 768];
    const uint32_t p0 = atomicAdd(&h[r0.y], 1u), p1 = atomicAdd(&h[r1.y], 1u), p2 = atomicAdd(&h[r2.y], 1u), p3 = atomicAdd(&h[r3.y], 1u);
    if (PLACE) { entries[p0] = r0.x; entries[p1] = r1.x; entries[p2] = r2.x; entries[p3] = r3.x; }
  }
  for (; r < end; r += 256) {
    const uint2 rr = rec[r];
    const uint32_t pos = atomicAdd(&h[rr.y], 1u);
    if (PLACE) entries[pos] = rr.x;
  }
  if (!PLACE) {
    __syncthreads();
    for (int t = threadIdx.x; t < PART_SUB; t += 256) subhist[(size_t)blk * PART_SUB + t] = h[t];
  }
}
// one block per partition: chunk histograms -> running offsets (in place); bucket totals -> counts[], offsets[] (and offsets[nbuckets] = all entries)
static __global__ void __launch_bounds__(PART_SUB) k_part_offsets(const uint32_t* __restrict__ tileoff, uint32_t P, uint32_t ntiles, const uint32_t* __restrict__ blkoff,
                                                           uint32_t* __restrict__ subhist, size_t nbuckets, uint32_t* __restrict__ counts, uint32_t* __restrict__ offsets) {
  ZKT_SIDE_PRIO;
  __shared__ uint32_t sc[PART_SUB];
  const uint32_t p = blockIdx.x, t = threadIdx.x;
  uint32_t run = 0;
  for (uint32_t cblk = blkoff[p]; cblk < blkoff[p + 1]; ++cblk) { const uint32_t v = subhist[(size_t)cblk * PART_SUB + t]; subhist[(size_t)cblk * PART_SUB + t] = run; run += v; }
  sc[t] = run; __syncthreads();
  for (int d = 1; d < PART_SUB; d <<= 1) { const uint32_t x = t >= (uint32_t)d ? sc[t - d] : 0u; __syncthreads(); sc[t] += x; __syncthreads(); }
  const size_t b = (size_t)p * PART_SUB + t;
  if (b < nbuckets) { counts[b] = run; offsets[b] = tileoff[(size_t)p * ntiles] + sc[t] - run; }
  if (p == P - 1 && t == 0) offsets[nbuckets] = tileoff[(size_t)P * ntiles];
}

// Work list.  A bucket of cnt entries is cut into nt = ceil(cnt / chunk) equal pieces (the first cnt % nt of them one entry longer), one
// piece per lane ("task"), so one lane never runs an unbounded list: with random scalars at 2^20 terms every bucket (~26 entries) is a single
// task, while skewed inputs (many equal scalars: a real witness is full of 0/1) and small MSMs (chunk from pick_chunk) spread their buckets
// over many lanes, whose partial sums are merged afterwards.  Tasks are ordered by size (largest first) by a counting sort, so the 64 lanes
// of a wave run equally long lists.
static constexpr uint32_t CHUNK = 128;           // the largest chunk (MsmPlan::chunk <= CHUNK): one size bin per possible task size
static constexpr int SIZE_BINS = CHUNK + 1;      // bin k holds tasks of size CHUNK - k
__device__ inline uint32_t ntasks_of(uint32_t c, uint32_t chunk) { return c <= chunk ? 1u : (c + chunk - 1) / chunk; }
__device__ inline uint32_t task_size_of(uint32_t c, uint32_t nt) { return (c + nt - 1) / nt; }      // the larger of the two piece sizes
static __global__ void __launch_bounds__(256) k_task_count(const uint32_t* __restrict__ counts, size_t m, uint32_t chunk, uint32_t* __restrict__ ntask, uint32_t* __restrict__ hist) {
  ZKT_SIDE_PRIO;
  __shared__ uint32_t h[SIZE_BINS];
  for (int i = threadIdx.x; i < SIZE_BINS; i += 256) h[i] = 0;
  __syncthreads();
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < m) {
    uint32_t c = counts[i], nt = ntasks_of(c, chunk);
    ntask[i] = nt;
    atomicAdd(&h[CHUNK - task_size_of(c, nt)], nt);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < SIZE_BINS; k += 256) if (h[k]) atomicAdd(&hist[k], h[k]);
}
// Buckets cut into more than HOT_NT pieces (a witness full of one value; the carry window of a plan whose windows end exactly at the scalars' top bit: half of
// all scalars then meet in bucket 0) are listed in hot[1..], hot[0] counting them: their partial sums are merged by 64 blocks each (k_merge_hot) instead
// of one wave walking thousands of partials.  More than HOT_CAP of them: the list is ignored and k_merge_partials does them all.
static constexpr uint32_t HOT_NT = 512, HOT_CAP = 64, HOT_FAN = 64;
static __global__ void __launch_bounds__(256) k_task_scatter(const uint32_t* __restrict__ counts, size_t m, uint32_t chunk, const uint32_t* __restrict__ size_hist,
                                                      uint32_t* __restrict__ bincur, uint2* __restrict__ order, uint32_t* __restrict__ hot) {
  ZKT_SIDE_PRIO;
  // rank inside the block with LDS atomics, then ONE global atomic per (block, non-empty bin): ~50 distinct sizes
  // are shared by 2^19 buckets, so per-element global atomics would serialise.  The exclusive scan of the SIZE_BINS-entry
  // histogram (bin offsets in the size-ordered task list) is recomputed by every block: 129 values, cheaper than three more launches.
  __shared__ uint32_t h[SIZE_BINS], base[SIZE_BINS], binoff[SIZE_BINS + 1], scan[256];
  __shared__ uint32_t hot_n, hot_b[256], hot_pos[256], hot_nt[256];
  for (int i = threadIdx.x; i < SIZE_BINS; i += 256) h[i] = 0;
  if (threadIdx.x == 0) hot_n = 0;
  {
    static_assert(SIZE_BINS <= 256, "one histogram bin per thread");
    uint32_t v = threadIdx.x < SIZE_BINS ? size_hist[threadIdx.x] : 0, tot;
    uint32_t ex = block_excl_scan_256(v, scan, tot);
    if (threadIdx.x < SIZE_BINS) binoff[threadIdx.x] = ex;
  }
  __syncthreads();
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t bin = 0, rank = 0, nt = 1;
  if (i < m) { uint32_t c = counts[i]; nt = ntasks_of(c, chunk); bin = CHUNK - task_size_of(c, nt); rank = atomicAdd(&h[bin], nt); }
  __syncthreads();
  for (int k = threadIdx.x; k < SIZE_BINS; k += 256) if (h[k]) base[k] = binoff[k] + atomicAdd(&bincur[k], h[k]);
  __syncthreads();
  if (i < m) {
    const uint32_t pos = base[bin] + rank;
    if (nt <= 16) { for (uint32_t k = 0; k < nt; ++k) order[pos + k] = make_uint2((uint32_t)i, k); }
    else { const uint32_t slot = atomicAdd(&hot_n, 1u); hot_b[slot] = (uint32_t)i; hot_pos[slot] = pos; hot_nt[slot] = nt; }      // a hot bucket: the whole block writes its tasks
    if (nt > HOT_NT) { const uint32_t g = atomicAdd(&hot[0], 1u); if (g < HOT_CAP) hot[1 + g] = (uint32_t)i; }
  }
  __syncthreads();
  for (uint32_t hb = 0; hb < hot_n; ++hb) {
    const uint32_t b = hot_b[hb], pos = hot_pos[hb], cnt = hot_nt[hb];
    for (uint32_t k = threadIdx.x; k < cnt; k += 256) order[pos + k] = make_uint2(b, k);
  }
}

// ---------------------------------------------------------------------------------
// bucket accumulation: one task (bucket chunk) per lane — the dominant kernel
// ---------------------------------------------------------------------------------
#ifndef ZKT_ACC_ATTR
#define ZKT_ACC_ATTR
#endif
template <class F>
__global__ void __launch_bounds__(64) ZKT_ACC_ATTR k_accumulate(const uint32_t* __restrict__ table, const uint32_t* __restrict__ entries,
                                                   const uint32_t* __restrict__ offsets, const uint2* __restrict__ order,
                                                   const uint32_t* __restrict__ task_off, size_t nbuckets,
                                                   uint32_t* __restrict__ sums, uint32_t* __restrict__ partial) {
  size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (t >= task_off[nbuckets]) return;
  const uint2 tk = order[t];
  const size_t b = tk.x;
  const uint32_t t0 = task_off[b], nt = task_off[b + 1] - t0;
  const uint32_t off = offsets[b], cnt = offsets[b + 1] - off;
  uint32_t beg = off, end = off + cnt;
  if (nt != 1) {                                   // piece tk.y of nt equal pieces: the first cnt % nt pieces hold one entry more
    const uint32_t q = cnt / nt, r = cnt - q * nt;
    beg = off + tk.y * q + (tk.y < r ? tk.y : r); end = beg + q + (tk.y < r ? 1u : 0u);
  }
  constexpr int CW = Coord<F>::CW, XYW = 4 * CW;
  Xyzz<F> acc = xyzz_inf<F>();
  // software-pipelined gather: the next point (and the entry after it) are in flight while this one is added, so the
  // random-access latency of the table hides under ~6000 VALU instructions instead of stalling the two waves of a SIMD
  typedef typename F::E E;
  uint32_t ent = beg < end ? entries[beg] : 0, ent_next = beg + 1 < end ? entries[beg + 1] : 0;      // empty bucket: harmless load of entry 0
  const uint32_t* p = table + (size_t)(ent & 0x7fffffffu) * (2 * CW);
#ifndef ZKT_ACC_PREFETCH_MAX_CW
#define ZKT_ACC_PREFETCH_MAX_CW 16
#endif
  if constexpr (CW <= ZKT_ACC_PREFETCH_MAX_CW) {
    E nx = Coord<F>::ld(p), ny = Coord<F>::ld(p + CW);
    for (uint32_t e = beg; e < end; ++e) {
      E x = nx, y = ny;
      const bool negate = ent >> 31;
      if (e + 1 < end) {
        ent = ent_next;
        ent_next = e + 2 < end ? entries[e + 2] : 0;
        p = table + (size_t)(ent & 0x7fffffffu) * (2 * CW);
        nx = Coord<F>::ld(p); ny = Coord<F>::ld(p + CW);
      }
      if (negate) y = F::neg(y);
      acc = xyzz_add_aff<F>(acc, x, y);
    }
  } else {                                        // wide coordinates (Fq2): the add is long enough to cover the gather; keep the registers
    for (uint32_t e = beg; e < end; ++e) {
      E x = Coord<F>::ld(p), y = Coord<F>::ld(p + CW);
      const bool negate = ent >> 31;
      ent = ent_next;
      ent_next = e + 2 < end ? entries[e + 2] : 0;
      p = table + (size_t)(ent & 0x7fffffffu) * (2 * CW);
      if (negate) y = F::neg(y);
      acc = xyzz_add_aff<F>(acc, x, y);
    }
  }
  st_xy<F>(nt == 1 ? sums + b * XYW : partial + (size_t)(t0 + tk.y) * XYW, acc);
}

// ---------------------------------------------------------------------------------
// bucket reduction  sum_b (b+1) S_b, b = hi*NLO + lo:
//   = sum_lo (lo+1) C_lo + NLO * sum_hi hi * R_hi,   C_lo = sum_hi S, R_hi = sum_lo S
// four lanes per point operation: msm_reduce_coop.h
// ---------------------------------------------------------------------------------
}  // namespace zkt
#include "msm_reduce_coop.h"
namespace zkt {

// direct form: total = sum_w 2^(c w) W_w.  One wave: lane w doubles its window result c*w times (the longest lane does the
// c*(nwin-1) <= 256 doublings a serial Horner would), then an LDS tree; lane 0 writes the Jacobian sum and its affine normalisation.
template <class F>
__global__ void __launch_bounds__(64) k_join_windows(const uint32_t* __restrict__ win_jac, int nwin, int c, uint32_t* __restrict__ out_jac, uint32_t* __restrict__ out_abi) {
  ZKT_SIDE_PRIO;
  constexpr int CW = Coord<F>::CW, XYW = 4 * CW, JW = 3 * CW;
  __shared__ uint32_t lds[16 * JW];
  const int t = threadIdx.x;
  auto ldj = [](const uint32_t* p) { return Jac<F>{Coord<F>::ld(p), Coord<F>::ld(p + CW), Coord<F>::ld(p + 2 * CW)}; };
  auto stj = [](uint32_t* p, const Jac<F>& a) { Coord<F>::st(p, a.X); Coord<F>::st(p + CW, a.Y); Coord<F>::st(p + 2 * CW, a.Z); };
  Jac<F> v = jac_inf<F>();
  if (t < nwin) {
    v = ldj(win_jac + (size_t)t * XYW);
    for (int d = 0; d < c * t; ++d) v = jac_dbl(v);
  }
  for (int d = 16; d >= 1; d >>= 1) {                  // nwin <= 32
    if (t >= d && t < 2 * d) stj(lds + (t - d) * JW, v);
    __syncthreads();
    if (t < d) v = jac_add(v, ldj(lds + t * JW));
    __syncthreads();
  }
  if (t == 0) {
    stj(out_jac, v);
    if (out_abi) PtIO<F>::st(out_abi, jac_to_aff(v));
  }
}

namespace {
struct MsmWs {   // workspace carve-up (one per in-flight MSM)
  uint32_t *zero_begin, *counts, *cursor, *size_hist, *size_off, *size_cur, *hot, *zero_end;   // [zero_begin, zero_end) is cleared per MSM
  uint32_t *offsets, *entries, *slot, *sums, *colsum, *rowsum, *clsA, *clsB, *win_jac, *scan_tmp, *ntask, *task_off, *partial, *hot_part;
  uint32_t *tilehist, *tileoff, *blkoff, *subhist, *part_scan; uint2* rec;   // the partition sort; rec = (entry, low bits of the bucket id) per digit
  uint2* order; size_t max_tasks;
};
MsmWs carve(const MsmPlan& P, void* workspace) {
  const size_t B = P.nbuckets, XYW = 4 * (size_t)coord_words(P.grp);
  uint8_t* ws = (uint8_t*)workspace;
  MsmWs w;
  w.zero_begin = (uint32_t*)ws;
  w.counts = (uint32_t*)ws; ws += (B + 1) * 4;
  w.cursor = (uint32_t*)ws; ws += (B + 1) * 4;
  w.size_hist = (uint32_t*)ws; ws += (SIZE_BINS + 1) * 4;
  w.size_off = (uint32_t*)ws; ws += (SIZE_BINS + 1) * 4;
  w.size_cur = (uint32_t*)ws; ws += (SIZE_BINS + 1) * 4;
  w.hot = (uint32_t*)ws; ws += (HOT_CAP + 1) * 4;
  w.zero_end = (uint32_t*)ws;
  w.offsets = (uint32_t*)ws; ws += (B + 1) * 4;
  ws = (uint8_t*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  w.entries = (uint32_t*)ws; ws += (size_t)P.nwin * P.n * 4;
  ws = (uint8_t*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  w.slot = (uint32_t*)ws; ws += (size_t)P.nwin * P.n * 4;
  ws = (uint8_t*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  w.sums = (uint32_t*)ws; ws += B * XYW * 4;
  const size_t nw = P.direct ? (size_t)P.nwin : 1;                // the direct form reduces every window side by side
  w.colsum = (uint32_t*)ws; ws += nw * 1024 * XYW * 4;
  w.rowsum = (uint32_t*)ws; ws += nw * 1024 * XYW * 4;
  w.clsA = (uint32_t*)ws; ws += nw * 32 * WB_SPLIT * XYW * 4;
  w.clsB = (uint32_t*)ws; ws += nw * 32 * WB_SPLIT * XYW * 4;
  w.win_jac = (uint32_t*)ws; ws += nw * XYW * 4;
  w.scan_tmp = (uint32_t*)ws; ws += 1024 * 4;
  w.ntask = (uint32_t*)ws; ws += (B + 1) * 4;
  w.task_off = (uint32_t*)ws; ws += (B + 1) * 4;
  w.max_tasks = B + (size_t)P.nwin * P.n / P.chunk + 1;
  ws = (uint8_t*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  w.order = (uint2*)ws; ws += w.max_tasks * 8;
  w.partial = (uint32_t*)ws; ws += w.max_tasks * XYW * 4;
  w.hot_part = (uint32_t*)ws; ws += (size_t)HOT_CAP * HOT_FAN * XYW * 4;
  {
    const PartDims d = part_dims(P.n, P.nbuckets, P.nwin);
    ws = (uint8_t*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    w.rec = (uint2*)ws; ws += (((size_t)P.nwin * P.n * 8) + 255) & ~(size_t)255;
    w.tilehist = (uint32_t*)ws; ws += ((size_t)d.P * d.ntiles + 1) * 4;
    w.tileoff = (uint32_t*)ws; ws += ((size_t)d.P * d.ntiles + 1) * 4;
    w.blkoff = (uint32_t*)ws; ws += ((size_t)d.P + 1) * 4;
    w.part_scan = (uint32_t*)ws; ws += ((size_t)d.P * d.ntiles / 2048 + 2) * 4;
    ws = (uint8_t*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    w.subhist = (uint32_t*)ws; ws += (size_t)d.maxblk * PART_SUB * 4;
  }
  return w;
}
}  // namespace

// The pipeline clears its counters with a kernel of its own rather than hipMemsetAsync: inside a captured graph (zkt_api.cpp, msm_submit_locked) a memset becomes a
// runtime-owned node, and the pipeline's launches should be the same objects whether they are issued or replayed.
static __global__ void __launch_bounds__(256) k_zero_words(uint32_t* __restrict__ p, int head_words, size_t quads, int tail_words) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < (size_t)head_words) p[i] = 0u;                                              // words before the first 16-byte boundary
  uint4* q = reinterpret_cast<uint4*>(p + head_words);
  if (i < quads) q[i] = uint4{0u, 0u, 0u, 0u};
  if (i < (size_t)tail_words) p[(size_t)head_words + quads * 4 + i] = 0u;
}
// ptr 4-byte aligned (the carve puts `offsets` at 4 mod 16), bytes a multiple of 4.  Never a hipMemsetAsync: a captured graph must hold kernel nodes only
// (zkt_api.cpp, msm_submit_locked: a runtime-owned memset node faults on the first replay after any later hipFree), and a zero-term MSM is captured like any other.
static hipError_t zero_async(void* ptr, size_t bytes, hipStream_t s) {
  if (bytes == 0) return hipSuccess;
  if (((uintptr_t)ptr & 3u) || (bytes & 3u)) return hipErrorInvalidValue;
  size_t words = bytes / 4;
  int head = (int)(((16u - ((uintptr_t)ptr & 15u)) & 15u) / 4);
  if ((size_t)head > words) head = (int)words;
  words -= (size_t)head;
  const size_t quads = words / 4; const int tail = (int)(words % 4);
  size_t items = quads > (size_t)tail ? quads : (size_t)tail;
  if (items < (size_t)head) items = (size_t)head;
  hipLaunchKernelGGL(k_zero_words, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, (uint32_t*)ptr, head, quads, tail);
  return hipGetLastError();
}
// stage 1 (atomic/memory bound): signed digits, counting sort by bucket, bucket order by population
hipError_t PART(launch_msm_sort)(const MsmPlan& P, const uint8_t* inf, const uint32_t* scalars, void* workspace, hipStream_t s) {
  const size_t B = P.nbuckets, n = P.n;
  MsmWs w = carve(P, workspace);
  hipError_t e;
  if ((e = zero_async(w.zero_begin, (uint8_t*)w.zero_end - (uint8_t*)w.zero_begin, s)) != hipSuccess) return e;
  const PartDims pd = part_dims(n, B, P.nwin);
  static const int force_sort = [] { const char* e = getenv("ZKT_MSM_SORT"); return e ? atoi(e) : 0; }();      // 1: atomics (k_digits), 2: partition sort, 0: by size
  const bool partition = force_sort == 2 || (force_sort != 1 && (size_t)P.nwin * n >= (size_t(1) << 22));        // below ~2^18 terms the atomics are as fast and take fewer launches
  if (n && partition && pd.P <= PART_MAXP) {
    // large MSMs: the atomic-free two-level partition (k_part_*); the record array lives in the slot buffer
    const uint32_t wb = P.direct ? (uint32_t)P.half : 0u;
    hipLaunchKernelGGL(k_part_tiles<false>, dim3(pd.ntiles), dim3(PART_TPB), 0, s, scalars, inf, n, P.c, P.nwin, wb, pd.P, pd.ntiles, w.tilehist, (const uint32_t*)nullptr, (uint2*)nullptr);
    launch_scan(w.tilehist, w.tileoff, (size_t)pd.P * pd.ntiles, w.part_scan, s);
    hipLaunchKernelGGL(k_part_tiles<true>, dim3(pd.ntiles), dim3(PART_TPB), 0, s, scalars, inf, n, P.c, P.nwin, wb, pd.P, pd.ntiles, (uint32_t*)nullptr, (const uint32_t*)w.tileoff, w.rec);
    hipLaunchKernelGGL(k_part_blocks, dim3(1), dim3(256), 0, s, (const uint32_t*)w.tileoff, pd.P, pd.ntiles, w.blkoff);
    hipLaunchKernelGGL(k_part_chunks<false>, dim3(pd.maxblk), dim3(256), 0, s, (const uint32_t*)w.tileoff, pd.P, pd.ntiles, (const uint32_t*)w.blkoff, (const uint2*)w.rec,
                       w.subhist, (const uint32_t*)nullptr, B, (uint32_t*)nullptr);
    hipLaunchKernelGGL(k_part_offsets, dim3(pd.P), dim3(PART_SUB), 0, s, (const uint32_t*)w.tileoff, pd.P, pd.ntiles, (const uint32_t*)w.blkoff, w.subhist, B, w.counts, w.offsets);
    hipLaunchKernelGGL(k_part_chunks<true>, dim3(pd.maxblk), dim3(256), 0, s, (const uint32_t*)w.tileoff, pd.P, pd.ntiles, (const uint32_t*)w.blkoff, (const uint2*)w.rec,
                       w.subhist, (const uint32_t*)w.offsets, B, w.entries);
  } else if (n) {
    const unsigned g = (unsigned)((n + 255) / 256);
    const uint32_t wb = P.direct ? (uint32_t)P.half : 0u;
    hipLaunchKernelGGL(k_digits<false>, dim3(g), dim3(256), 0, s, scalars, inf, n, P.c, P.nwin, wb, w.counts, (const uint32_t*)nullptr, w.slot, (uint32_t*)nullptr);
    launch_scan(w.counts, w.offsets, B, w.scan_tmp, s);
    hipLaunchKernelGGL(k_digits<true>, dim3(g), dim3(256), 0, s, scalars, inf, n, P.c, P.nwin, wb, (uint32_t*)nullptr, (const uint32_t*)w.offsets, w.slot, w.entries);
  } else {
    if ((e = zero_async(w.offsets, (B + 1) * 4, s)) != hipSuccess) return e;
  }
  const uint32_t* task_counts = w.counts;
  if (P.aff_rounds) {                      // the accumulate kernel sees what the affine rounds leave: ceil(cnt / 2^R) points per bucket
    const MsmAffineWs aw = msm_affine_carve((uint8_t*)workspace + P.aff_off, (size_t)P.nwin * n, B, P.aff_rounds, coord_words(P.grp));
    if ((e = launch_msm_affine_final_layer(w.offsets, B, P.aff_rounds, aw, s)) != hipSuccess) return e;
    task_counts = aw.cntR;
  }
  hipLaunchKernelGGL(k_task_count, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, task_counts, B, P.chunk, w.ntask, w.size_hist);
  launch_scan(w.ntask, w.task_off, B, w.scan_tmp, s);
  hipLaunchKernelGGL(k_task_scatter, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, task_counts, B, P.chunk, (const uint32_t*)w.size_hist, w.size_cur, w.order, w.hot);
  return hipGetLastError();
}
// stage 2 (VALU bound, the dominant kernel): one bucket per lane
hipError_t PART(launch_msm_accumulate)(const MsmPlan& P, const uint32_t* table, void* workspace, hipStream_t s) {
  MsmWs w = carve(P, workspace);
#if defined(ZKT_MSM_PART_OTHER) && !defined(ZKT_G2_ONE_LANE)
  if (P.grp == G_G2 && P.aff_rounds) {      // affine pair-tree rounds, then the XYZZ accumulate over their last layer (zkt_msm_affine.hip)
    const size_t ent = (size_t)P.nwin * P.n;
    const MsmAffineWs aw = msm_affine_carve((uint8_t*)workspace + P.aff_off, ent, P.nbuckets, P.aff_rounds, coord_words(P.grp));
    hipError_t e = launch_msm_affine_rounds_g2(table, (const uint32_t*)w.entries, (const uint32_t*)w.offsets, P.nbuckets, ent, P.aff_rounds, aw, s);
    if (e != hipSuccess) return e;
    return ::zkt_launch_accumulate_g2_pair_direct(aw.pts[P.aff_rounds], aw.inf[P.aff_rounds], aw.offR, aw.cntR, (const void*)w.order, (const uint32_t*)w.task_off, P.nbuckets,
                                                  w.sums, w.partial, w.max_tasks, s);
  }
  if (P.grp == G_G2)        // two lanes per task: zkt_msm_g2pair.hip
    return ::zkt_launch_accumulate_g2_pair(table, (const uint32_t*)w.entries, (const uint32_t*)w.offsets, (const void*)w.order, (const uint32_t*)w.task_off, P.nbuckets,
                                           w.sums, w.partial, w.max_tasks, s);
#endif
  MSM_DISPATCH(P.grp, hipLaunchKernelGGL(k_accumulate<F>, dim3((unsigned)((w.max_tasks + 63) / 64)), dim3(64), 0, s, table, (const uint32_t*)w.entries,
                                         (const uint32_t*)w.offsets, (const uint2*)w.order, (const uint32_t*)w.task_off, P.nbuckets, w.sums, w.partial));
  return hipGetLastError();
}
// stage 3 (latency bound): sum_b (b+1) S_b, b = hi*NLO + lo  ->  Jacobian partial (+ affine point if out_abi)
hipError_t PART(launch_msm_reduce)(const MsmPlan& P, void* workspace, uint32_t* dev_result_jac, uint32_t* dev_out_abi, hipStream_t s) {
  const size_t B = P.direct ? P.half : P.nbuckets;          // buckets of one reduction (per window in the direct form)
  MsmWs w = carve(P, workspace);
  const size_t NLO = B < 1024 ? B : 1024, NHI = B / NLO;
  int lo_bits = 0; while ((size_t(1) << lo_bits) < NLO) ++lo_bits;
  int hi_bits = 0; while ((size_t(1) << hi_bits) < NHI) ++hi_bits;
  const int nbA = lo_bits + 1, nbB = NHI > 1 ? hi_bits : 0;     // weights lo+1 in [1,NLO]; hi in [0,NHI)
  // pieces per row (k_marginals): a column block sums NHI points, a row piece NLO / RS — cut the rows until both carry chains of the same length
  // (at 2^16 buckets: 64 x 1024, RS = 16 gives 64-point pieces, one per lane + the tree, instead of 8 per lane), within the 1024-entry row buffer
  int RS = 1;
  while (NHI > 1 && (size_t)(2 * RS) * NHI <= 1024 && NLO / (size_t)(2 * RS) >= 64 && NLO / (size_t)(2 * RS) >= NHI) RS *= 2;
  const size_t NROW = (size_t)RS * NHI;
  if (P.direct) {                                          // every window reduced side by side (grid.y), then joined
    const unsigned ny = (unsigned)P.nwin;
    MSM_DISPATCH(P.grp,
      hipLaunchKernelGGL(k_merge_hot<F>, dim3(HOT_FAN, HOT_CAP), dim3(RED_TPB), 0, s, (const uint32_t*)w.task_off, (const uint32_t*)w.partial, (const uint32_t*)w.hot, w.hot_part);
      hipLaunchKernelGGL(k_merge_partials<F>, dim3((unsigned)((P.nbuckets + RED_NG - 1) / RED_NG < 4096 ? (P.nbuckets + RED_NG - 1) / RED_NG : 4096)), dim3(RED_TPB), 0, s, (const uint32_t*)w.task_off, P.nbuckets, (const uint32_t*)w.partial, (const uint32_t*)w.hot, (const uint32_t*)w.hot_part, w.sums);
      hipLaunchKernelGGL(k_marginals<F>, dim3((unsigned)(NLO + (NHI > 1 ? NROW : 0)), ny), dim3(RED_TPB), 0, s, (const uint32_t*)w.sums, NLO, NHI, RS, w.colsum, w.rowsum);
      hipLaunchKernelGGL(k_weight_bits<F>, dim3((unsigned)(nbA + nbB) * WB_SPLIT, ny), dim3(WB_TPB), 0, s, (const uint32_t*)w.colsum, NLO, nbA, (const uint32_t*)w.rowsum, NHI, RS, w.clsA, w.clsB);
      hipLaunchKernelGGL(k_combine<F>, dim3(1, ny), dim3(CMB_TPB), 0, s, (const uint32_t*)w.clsA, nbA, (const uint32_t*)w.clsB, nbB, lo_bits, w.win_jac, (uint32_t*)nullptr);
      hipLaunchKernelGGL(k_join_windows<F>, dim3(1), dim3(64), 0, s, (const uint32_t*)w.win_jac, P.nwin, P.c, dev_result_jac, dev_out_abi));
    return hipGetLastError();
  }
  MSM_DISPATCH(P.grp,
    hipLaunchKernelGGL(k_merge_hot<F>, dim3(HOT_FAN, HOT_CAP), dim3(RED_TPB), 0, s, (const uint32_t*)w.task_off, (const uint32_t*)w.partial, (const uint32_t*)w.hot, w.hot_part);
    hipLaunchKernelGGL(k_merge_partials<F>, dim3((unsigned)((B + RED_NG - 1) / RED_NG < 4096 ? (B + RED_NG - 1) / RED_NG : 4096)), dim3(RED_TPB), 0, s, (const uint32_t*)w.task_off, B, (const uint32_t*)w.partial, (const uint32_t*)w.hot, (const uint32_t*)w.hot_part, w.sums);
    hipLaunchKernelGGL(k_marginals<F>, dim3((unsigned)(NLO + (NHI > 1 ? NROW : 0))), dim3(RED_TPB), 0, s, (const uint32_t*)w.sums, NLO, NHI, RS, w.colsum, w.rowsum);
    hipLaunchKernelGGL(k_weight_bits<F>, dim3((unsigned)(nbA + nbB) * WB_SPLIT), dim3(WB_TPB), 0, s, (const uint32_t*)w.colsum, NLO, nbA, (const uint32_t*)w.rowsum, NHI, RS, w.clsA, w.clsB);
    hipLaunchKernelGGL(k_combine<F>, dim3(1), dim3(CMB_TPB), 0, s, (const uint32_t*)w.clsA, nbA, (const uint32_t*)w.clsB, nbB, lo_bits, dev_result_jac, dev_out_abi));
  return hipGetLastError();
}

// window-multiple table build
hipError_t PART(launch_msm_precompute)(int grp, uint32_t* table, uint8_t* inf, size_t n, int c, int nwin, uint32_t* tmp, hipStream_t s) {
  if (n == 0) return hipSuccess;
  MSM_DISPATCH(grp, hipLaunchKernelGGL(k_precompute<F>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, table, inf, n, c, nwin, tmp));
  return hipGetLastError();
}

// combine step of a sharded MSM + affine normalisation: sum of `count` Jacobian partials (3*CW words each, `stride` words apart).
// One wave: lane t sums partials t, t+64, ..., then an LDS tree (6 levels) — 8 partials cost 3 additions of latency, not 8.
template <class F>
__global__ void __launch_bounds__(64) k_jac_sum_to_affine(const uint32_t* __restrict__ parts, size_t count, size_t stride, uint32_t* __restrict__ out_abi) {
  constexpr int CW = Coord<F>::CW, JW = 3 * CW;
  __shared__ uint32_t lds[32 * JW];
  const int t = threadIdx.x;
  auto ldj = [](const uint32_t* p) { return Jac<F>{Coord<F>::ld(p), Coord<F>::ld(p + CW), Coord<F>::ld(p + 2 * CW)}; };
  auto stj = [](uint32_t* p, const Jac<F>& a) { Coord<F>::st(p, a.X); Coord<F>::st(p + CW, a.Y); Coord<F>::st(p + 2 * CW, a.Z); };
  Jac<F> v = jac_inf<F>();
  for (size_t i = t; i < count; i += 64) v = jac_add<F>(v, ldj(parts + i * stride));
  for (int d = 32; d >= 1; d >>= 1) {
    if (t >= d && t < 2 * d) stj(lds + (t - d) * JW, v);
    __syncthreads();
    if (t < d && (size_t)(t + d) < count) v = jac_add<F>(v, ldj(lds + t * JW));
    __syncthreads();
  }
  if (t == 0) PtIO<F>::st(out_abi, jac_to_aff(v));
}
// out = a + b on Jacobian partials (3 coordinates each): two partial sums of ONE result that were accumulated over different base sets
template <class F>
__global__ void __launch_bounds__(64) k_jac_add(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t* __restrict__ out) {
  if (threadIdx.x) return;
  constexpr int CW = Coord<F>::CW;
  const Jac<F> x{Coord<F>::ld(a), Coord<F>::ld(a + CW), Coord<F>::ld(a + 2 * CW)}, y{Coord<F>::ld(b), Coord<F>::ld(b + CW), Coord<F>::ld(b + 2 * CW)};
  const Jac<F> r = jac_add<F>(x, y);
  Coord<F>::st(out, r.X); Coord<F>::st(out + CW, r.Y); Coord<F>::st(out + 2 * CW, r.Z);
}
hipError_t PART(launch_msm_jac_add)(int grp, const uint32_t* a, const uint32_t* b, uint32_t* out, hipStream_t s) {
  MSM_DISPATCH(grp, hipLaunchKernelGGL(k_jac_add<F>, dim3(1), dim3(64), 0, s, a, b, out));
  return hipGetLastError();
}
hipError_t PART(launch_msm_jac_sum_to_affine)(int grp, const uint32_t* parts, size_t count, size_t stride, uint32_t* out_abi, hipStream_t s) {
  MSM_DISPATCH(grp, hipLaunchKernelGGL(k_jac_sum_to_affine<F>, dim3(1), dim3(64), 0, s, parts, count, stride, out_abi));
  return hipGetLastError();
}


#if defined(ZKT_MSM_PART_G1)
// public entry points: G1 lives in this object, secp256k1 in the PART_SECP object, G2 in the PART_OTHER object
#define ZKT_MSM_FWD(SUF) \
  hipError_t launch_msm_to_kernel_layout##SUF(int, const uint32_t*, uint32_t*, uint8_t*, size_t, hipStream_t); \
  hipError_t launch_msm_precompute##SUF(int, uint32_t*, uint8_t*, size_t, int, int, uint32_t*, hipStream_t); \
  hipError_t launch_msm_sort##SUF(const MsmPlan&, const uint8_t*, const uint32_t*, void*, hipStream_t); \
  hipError_t launch_msm_accumulate##SUF(const MsmPlan&, const uint32_t*, void*, hipStream_t); \
  hipError_t launch_msm_reduce##SUF(const MsmPlan&, void*, uint32_t*, uint32_t*, hipStream_t); \
  hipError_t launch_msm_jac_add##SUF(int, const uint32_t*, const uint32_t*, uint32_t*, hipStream_t); \
  hipError_t launch_msm_jac_sum_to_affine##SUF(int, const uint32_t*, size_t, size_t, uint32_t*, hipStream_t);
ZKT_MSM_FWD(_other) ZKT_MSM_FWD(_secp)
#define ZKT_MSM_BY_GROUP(grp, NAME, ...) ((grp) == G_G1 ? NAME##_g1(__VA_ARGS__) : (grp) == G_SECP ? NAME##_secp(__VA_ARGS__) : NAME##_other(__VA_ARGS__))
hipError_t launch_msm_to_kernel_layout(int grp, const uint32_t* a, uint32_t* t, uint8_t* i, size_t n, hipStream_t s) { return ZKT_MSM_BY_GROUP(grp, launch_msm_to_kernel_layout, grp, a, t, i, n, s); }
hipError_t launch_msm_precompute(int grp, uint32_t* t, uint8_t* i, size_t n, int c, int nw, uint32_t* tmp, hipStream_t s) { return ZKT_MSM_BY_GROUP(grp, launch_msm_precompute, grp, t, i, n, c, nw, tmp, s); }
hipError_t launch_msm_sort(const MsmPlan& P, const uint8_t* i, const uint32_t* k, void* w, hipStream_t s) { return ZKT_MSM_BY_GROUP(P.grp, launch_msm_sort, P, i, k, w, s); }
hipError_t launch_msm_accumulate(const MsmPlan& P, const uint32_t* t, void* w, hipStream_t s) { return ZKT_MSM_BY_GROUP(P.grp, launch_msm_accumulate, P, t, w, s); }
hipError_t launch_msm_reduce(const MsmPlan& P, void* w, uint32_t* j, uint32_t* o, hipStream_t s) { return ZKT_MSM_BY_GROUP(P.grp, launch_msm_reduce, P, w, j, o, s); }
hipError_t launch_msm_jac_add(int grp, const uint32_t* a, const uint32_t* b, uint32_t* o, hipStream_t s) { return ZKT_MSM_BY_GROUP(grp, launch_msm_jac_add, grp, a, b, o, s); }
hipError_t launch_msm_jac_sum_to_affine(int grp, const uint32_t* p, size_t c, size_t st, uint32_t* o, hipStream_t s) { return ZKT_MSM_BY_GROUP(grp, launch_msm_jac_sum_to_affine, grp, p, c, st, o, s); }
#endif

}  // namespace zkt
