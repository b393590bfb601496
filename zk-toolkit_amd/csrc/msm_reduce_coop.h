// Bucket reduction of the MSM with FOUR LANES PER POINT OPERATION (included by zkt_msm.hip; row a9 / f-1 of SURVEY §8).
//
// The reduce stage of an MSM (merge of split buckets -> two-level marginals -> bit classes -> combine) is a chain of short trees: with one
// point operation per lane every tree level is a full XYZZ addition (12M + 2S in sequence, ~25 us on one lane of this machine, where a
// dependent v_mad_u64_u32 chain is what bounds a lane) while half of the lanes idle at every level.  Round 2 measured the consequence:
// 1.25 ms of latency-bound tail on a 4.6 ms G1 MSM, 1.4 ms of 2.0 ms at 2^17 terms, and for G2 — where one lane holding a whole Fq2 XYZZ
// addition spilled 590-635 VGPRs — 6.6 ms of a 24.6 ms Groth16 proof.
//
// Here a GROUP of four adjacent lanes performs one point operation.  The addition formulas (add-2008-s) have four independent products at
// three of their four dependency levels, so a group runs an addition as FOUR rounds of ONE field multiplication per lane (a doubling as
// three) instead of fourteen in sequence.  Operands live in numbered LDS slots of the group; a round is the same code for every lane —
//     c = slot[a] * slot[b]  ->  slot[d]
// with the slot numbers read from a per-lane table, so the multiplications (nine tenths of the work) run without divergence between the roles;
// the few differences the formulas need (P = U2 - U1, R = S2 - S1, Q - X3) are made by the one lane that consumes them.  A lane holds two operands and one product at a time: ~120 VGPRs for Fq2, nothing spills.
// The exceptional cases of the group law (macros.rs:43-63: an operand at infinity, P + P, P + (-P)) are detected by every lane of the group
// from the same LDS values: infinity operands are a four-lane copy, P + (-P) a four-lane store, P + P the cooperative doubling.
#pragma once

namespace zkt {

template <class F> struct Coop {
  typedef typename F::E E;
  static constexpr int CW = Coord<F>::CW, SW = (CW + 3) & ~3;        // slot words: 16-byte aligned rows
  enum Slot { AX = 0, AY, AZZ, AZZZ, BX, BY, BZZ, BZZZ, T0, T1, T2, T3, T4, T5, T6, T7, NSLOT };
  static constexpr int GW = NSLOT * SW;                                // LDS words per group
  __device__ static E ld(const uint32_t* g, int slot) { return Coord<F>::ld(g + slot * SW); }
  __device__ static void st(uint32_t* g, int slot, const E& v) { Coord<F>::st(g + slot * SW, v); }
};
__device__ inline void coop_sync() { __syncthreads(); }
// lane r of a group picks entry r of a four-entry table (compile-time constants: two v_cndmask)
__device__ inline int pick4(int r, int a, int b, int c, int d) { return r == 0 ? a : r == 1 ? b : r == 2 ? c : d; }

template <class F> __device__ inline void coop_set_inf(uint32_t* g, int r) {         // A <- infinity (X = Y = 1, ZZ = ZZZ = 0), lane r writes coordinate r
  Coop<F>::st(g, Coop<F>::AX + r, r < 2 ? F::one() : F::zero());
}
template <class F> __device__ inline void coop_init(uint32_t* g, int r) { coop_set_inf<F>(g, r); }      // A = infinity (then coop_sync)

// flag of lane `src` (0..3) of the caller's group, for every lane of the group
__device__ inline bool group_flag(bool f, int src) { return __shfl((int)f, (int)(threadIdx.x & ~3u) + src) != 0; }

// A <- 2 A (dbl-2008-s-1), three rounds.  Every lane of the BLOCK must call it (it contains barriers); `active` = this group has a doubling
// to do (divergence between the groups of a wave is fine: an idle group only keeps the barriers).
template <class F> __device__ inline void coop_dbl(uint32_t* g, int r, bool active) {
  typedef Coop<F> K; typedef typename F::E E;
  bool inf = false;
  if (active) {
    // round 1: V = U^2 with U = 2Y -> T0 (U -> T5);  XX = X^2 -> T1 (M = 3 XX -> T6);  lanes 2, 3 test ZZ and Y for zero meanwhile
    if (r < 2) {
      E x = K::ld(g, r == 0 ? K::AY : K::AX);
      if (r == 0) { x = F::add(x, x); K::st(g, K::T5, x); }
      const E c = F::mul(x, x);
      K::st(g, K::T0 + r, c);
      if (r == 1) K::st(g, K::T6, F::add(F::add(c, c), c));
    } else inf = F::is_zero(K::ld(g, r == 2 ? K::AZZ : K::AY));            // infinity, or a point of order two (y = 0, macros.rs:61-63)
    inf = group_flag(inf, 2) || group_flag(inf, 3);
  }
  coop_sync();
  if (active) {
    // round 2: W = U V -> T2, S = X V -> T3, MM = M^2 -> T4, ZZ3 = V ZZ -> AZZ (an infinite A is rewritten at the end)
    K::st(g, pick4(r, K::T2, K::T3, K::T4, K::AZZ), F::mul(K::ld(g, pick4(r, K::T5, K::AX, K::T6, K::T0)), K::ld(g, pick4(r, K::T0, K::T0, K::T6, K::AZZ))));
  }
  coop_sync();
  if (active && r < 3) {
    // round 3: X3 = MM - 2S -> AX;  ZZZ3 = W ZZZ -> AZZZ, WY = W Y -> T7, Y3a = M (S - X3) -> T1
    E b = K::ld(g, pick4(r, K::AZZZ, K::AY, K::T3, K::T3));
    if (r == 2) { const E x3 = F::sub2(K::ld(g, K::T4), F::zero(), b); K::st(g, K::AX, x3); b = F::sub(b, x3); }
    K::st(g, pick4(r, K::AZZZ, K::T7, K::T1, K::T1), F::mul(K::ld(g, pick4(r, K::T2, K::T2, K::T6, K::T6)), b));
  }
  coop_sync();
  if (active) {
    if (inf) coop_set_inf<F>(g, r);
    else if (r == 1) K::st(g, K::AY, F::sub(K::ld(g, K::T1), K::ld(g, K::T7)));                // Y3 = M (S - X3) - W Y
  }
  coop_sync();
}

// A <- A + B for the group's XYZZ points in slots A*, B* (B is clobbered), four rounds.  Same calling rule.
template <class F> __device__ inline void coop_add(uint32_t* g, int r, bool active) {
  typedef Coop<F> K; typedef typename F::E E;
  bool infA = false, infB = false, eqx = false, eqy = false, ok = false;
  if (active) {
    // round 1: U1 = X1 ZZ2 -> T0, U2 = X2 ZZ1 -> T1, S1 = Y1 ZZZ2 -> T2, S2 = Y2 ZZZ1 -> T3; lanes 0 / 1 test ZZ1 / ZZ2 for zero on the way
    const E b = K::ld(g, pick4(r, K::BZZ, K::AZZ, K::BZZZ, K::AZZZ));
    const bool z = r < 2 && F::is_zero(b);
    infB = group_flag(z, 0); infA = group_flag(z, 1);
    K::st(g, K::T0 + r, F::mul(K::ld(g, pick4(r, K::AX, K::BX, K::AY, K::BY)), b));
  }
  coop_sync();
  if (active) {
    // round 2: ZZ12 = ZZ1 ZZ2 -> T4, ZZZ12 = ZZZ1 ZZZ2 -> T5, PP = P^2 -> T6, RR = R^2 -> T7 with P = U2 - U1 -> T1, R = S2 - S1 -> T3
    E a = K::ld(g, pick4(r, K::AZZ, K::AZZZ, K::T1, K::T3)), b;
    bool z = false;
    if (r >= 2) { a = F::sub(a, K::ld(g, r == 2 ? K::T0 : K::T2)); z = F::is_zero(a); K::st(g, r == 2 ? K::T1 : K::T3, a); b = a; }
    else b = K::ld(g, r == 0 ? K::BZZ : K::BZZZ);
    eqx = group_flag(z, 2); eqy = group_flag(z, 3);
    K::st(g, K::T4 + r, F::mul(a, b));
    ok = !infA && !infB && !eqx;                                         // the generic case: results overwrite A (and the dead B slots) from here on
  }
  coop_sync();
  if (ok && r < 3) {
    // round 3: PPP = P PP -> BX, Q = U1 PP -> BY, ZZ3 = ZZ12 PP -> AZZ
    K::st(g, pick4(r, K::BX, K::BY, K::AZZ, K::AZZ), F::mul(K::ld(g, pick4(r, K::T1, K::T0, K::T4, K::T4)), K::ld(g, K::T6)));
  }
  coop_sync();
  if (ok && r < 3) {
    // round 4: X3 = RR - PPP - 2Q -> AX;  ZZZ3 = ZZZ12 PPP -> AZZZ, T = S1 PPP -> BZZ, Y3a = R (Q - X3) -> BZZZ
    E b = K::ld(g, pick4(r, K::BX, K::BX, K::BY, K::BY));
    if (r == 2) { const E x3 = F::sub2(K::ld(g, K::T7), K::ld(g, K::BX), b); K::st(g, K::AX, x3); b = F::sub(b, x3); }
    K::st(g, pick4(r, K::AZZZ, K::BZZ, K::BZZZ, K::BZZZ), F::mul(K::ld(g, pick4(r, K::T5, K::T2, K::T3, K::T3)), b));
  }
  coop_sync();
  if (active) {
    if (ok) { if (r == 1) K::st(g, K::AY, F::sub(K::ld(g, K::BZZZ), K::ld(g, K::BZZ))); }       // Y3 = R (Q - X3) - S1 PPP
    else if (infB) {}                                                                          // A + infinity = A
    else if (infA) K::st(g, K::AX + r, K::ld(g, K::BX + r));                                   // infinity + B = B: lane r copies coordinate r
    else if (!eqy) coop_set_inf<F>(g, r);                                                      // P + (-P) = infinity (macros.rs:53-56)
  }
  coop_sync();
  // P + P: the explicit doubling branch of macros.rs:57-108 (repeated bases do occur: bulletproofs.rs:231-246 uses gg = [g, g]).  The rounds of the
  // doubling are always walked for their barriers; a group without this case skips their work.
  coop_dbl<F>(g, r, active && !infA && !infB && eqx && eqy);
}

// lane r of the group moves coordinate r of an XYZZ point between global memory (4 * CW words, the layout of ld_xy / st_xy) and the group's slots
template <class F> __device__ inline void coop_load(uint32_t* g, int r, int first_slot, const uint32_t* p) {
  Coop<F>::st(g, first_slot + r, Coord<F>::ld(p + r * Coord<F>::CW));
}
template <class F> __device__ inline void coop_store(const uint32_t* g, int r, uint32_t* p) { Coord<F>::st(p + r * Coord<F>::CW, Coop<F>::ld(g, Coop<F>::AX + r)); }

// tree over the NG groups of a block: A of group 0 <- sum of all A.  NG a power of two.
template <class F, int NG> __device__ inline void coop_block_tree(uint32_t* lds, int grp, int r) {
  typedef Coop<F> K;
  uint32_t* g = lds + grp * K::GW;
#pragma unroll 1
  for (int k = NG / 2; k >= 1; k >>= 1) {
    const bool act = grp < k;
    if (act) K::st(g, K::BX + r, K::ld(lds + (grp + k) * K::GW, K::AX + r));
    coop_sync();
    coop_add<F>(g, r, act);
  }
}

// ---------------------------------------------------------------------------------
// the reduce kernels
// ---------------------------------------------------------------------------------
static constexpr int RED_TPB = 64, RED_NG = RED_TPB / 4;             // one wave = 16 groups per block
// two waves per SIMD for every reduce kernel: the one-lane bulk phase of the prime-field kernels would take 280 registers (one wave per SIMD, the
// 2048 marginal blocks of a 2^20-term MSM in two rounds) and the Fq2 kernels 257 — capped at 256 they keep a few dwords in scratch instead
#ifndef ZKT_RED_ATTR
#define ZKT_RED_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif

// Where the one-lane XYZZ addition fits the register file (the prime-field groups) a LONG list is first summed one point per lane — every lane busy,
// no exchange — and only the 4 * NG lane sums (each group's own four) go through the groups: the cooperative form pays ~15 % in idle lane-rounds and LDS traffic, which is
// the wrong trade while there is a point for every lane (marginals of a 2^20-term MSM: 512 points per block).  For Fq2 the one-lane addition
// spills several hundred registers, so G2 stays cooperative throughout.
template <class F> struct CoopBulk { static constexpr bool serial = false; };
template <class C> struct CoopBulk<PrimeOps<C>> { static constexpr bool serial = true; };

// sum of `count` points in[(first + j * stride) * XYW], j < count, by the NG groups of the block (4 * NG lanes) -> A of group 0
template <class F, int NG> __device__ inline void coop_block_sum(uint32_t* lds, int grp, int r, const uint32_t* __restrict__ in, size_t first, size_t stride, size_t count) {
  typedef Coop<F> K; constexpr int XYW = 4 * Coord<F>::CW;
  uint32_t* g = lds + grp * K::GW;
  bool bulk = false;
  if constexpr (CoopBulk<F>::serial) bulk = count > (size_t)4 * NG;     // block-uniform
  if (bulk) {
    if constexpr (CoopBulk<F>::serial) {
      Xyzz<F> acc = xyzz_inf<F>();
      for (size_t j = threadIdx.x; j < count; j += 4 * NG) acc = xyzz_add<F>(acc, ld_xy<F>(in + (first + j * stride) * XYW));
      // the four lane sums of a group are that group's first four points: lane 0's goes to A, the others to B one after the other
#pragma unroll 1
      for (int k = 0; k < 4; ++k) {
        if (r == k) { const int base = k == 0 ? K::AX : K::BX; K::st(g, base, acc.X); K::st(g, base + 1, acc.Y); K::st(g, base + 2, acc.ZZ); K::st(g, base + 3, acc.ZZZ); }
        if (k) { coop_sync(); coop_add<F>(g, r, true); }
      }
    }
  } else {
    coop_init<F>(g, r);
    coop_sync();
#pragma unroll 1
    for (size_t j0 = 0; j0 < count; j0 += NG) {                        // block-uniform trip count
      const size_t j = j0 + grp; const bool act = j < count;
      if (act) coop_load<F>(g, r, K::BX, in + (first + j * stride) * XYW);
      coop_sync();
      coop_add<F>(g, r, act);
    }
  }
  coop_block_tree<F, NG>(lds, grp, r);
}

// hot buckets (k_task_scatter's list): the HOT_FAN blocks of bucket h sum its partials k = bx * NG + grp (mod HOT_FAN * NG) -> hot_part[h][bx]
template <class F>
__global__ void __launch_bounds__(RED_TPB) ZKT_RED_ATTR k_merge_hot(const uint32_t* __restrict__ task_off, const uint32_t* __restrict__ partial, const uint32_t* __restrict__ hot,
                                                       uint32_t* __restrict__ hot_part) {
  ZKT_SIDE_PRIO;
  typedef Coop<F> K; constexpr int XYW = 4 * Coord<F>::CW;
  __shared__ __attribute__((aligned(16))) uint32_t lds[RED_NG * K::GW];
  const uint32_t hc = hot[0], h = blockIdx.y;
  if (hc > HOT_CAP || h >= hc) return;                                 // block-uniform
  const int grp = threadIdx.x >> 2, r = threadIdx.x & 3;
  const uint32_t b = hot[1 + h], t0 = task_off[b], nt = task_off[b + 1] - t0;
  const size_t stride = (size_t)HOT_FAN * RED_NG;
  const size_t mine = (size_t)blockIdx.x * RED_NG < nt ? ((size_t)nt - blockIdx.x * RED_NG + stride - 1) / stride : 0;      // partials k = bx*NG + i*stride + grp < nt
  // (rows of NG consecutive partials, one per group; the last row may be ragged)
  uint32_t* g = lds + grp * K::GW;
  coop_init<F>(g, r);
  coop_sync();
#pragma unroll 1
  for (size_t i = 0; i < mine; ++i) {
    const size_t k = (size_t)blockIdx.x * RED_NG + i * stride + grp; const bool act = k < nt;
    if (act) coop_load<F>(g, r, K::BX, partial + ((size_t)t0 + k) * XYW);
    coop_sync();
    coop_add<F>(g, r, act);
  }
  coop_block_tree<F, RED_NG>(lds, grp, r);
  if (grp == 0) coop_store<F>(g, r, hot_part + ((size_t)h * HOT_FAN + blockIdx.x) * XYW);
}

// split buckets: sums[b] = sum of the bucket's partials.  A block takes tiles of RED_NG buckets (strided, below): a bucket cut into a few pieces is summed by its own
// group, piece after piece (small MSMs cut every bucket); a bucket with many pieces takes the whole block (groups in parallel + the tree), and a
// listed hot bucket is the sum of the HOT_FAN block sums k_merge_hot left.
template <class F>
__global__ void __launch_bounds__(RED_TPB) ZKT_RED_ATTR k_merge_partials(const uint32_t* __restrict__ task_off, size_t nbuckets, const uint32_t* __restrict__ partial,
                                                            const uint32_t* __restrict__ hot, const uint32_t* __restrict__ hot_part, uint32_t* __restrict__ sums) {
  ZKT_SIDE_PRIO;
  typedef Coop<F> K; constexpr int XYW = 4 * Coord<F>::CW;
  __shared__ __attribute__((aligned(16))) uint32_t lds[RED_NG * K::GW];
  static_assert(RED_TPB == 64 && HOT_CAP <= 64, "one wave per block: the ballots below cover the block");
  constexpr uint32_t MERGE_GROUP_MAX = 8;
  const int grp = threadIdx.x >> 2, r = threadIdx.x & 3;
  uint32_t* g = lds + grp * K::GW;
  const uint32_t hc = hot[0];
  const uint32_t my_hot = threadIdx.x < hc && hc <= HOT_CAP ? hot[1 + threadIdx.x] : 0xffffffffu;
  // A tile is RED_NG buckets `ntiles` apart, NOT RED_NG neighbours: buckets with many pieces come in runs of neighbours (the few digits a narrow top window can hold, small
  // witness values) and a tile works through its many-piece buckets one after the other — sixteen neighbours in one tile were 0.84 ms of a 1.6 ms MSM (2^14 terms at c = 14,
  // profiles/r04_msm_window_sweep.txt), spread over sixteen tiles they are one bucket's time.
  const size_t ntiles = (nbuckets + RED_NG - 1) / RED_NG;
  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const size_t mine = tile + (size_t)grp * ntiles;
    const uint32_t my_t0 = mine < nbuckets ? task_off[mine] : 0u, my_nt = mine < nbuckets ? task_off[mine + 1] - my_t0 : 1u;
    const unsigned long long few = __ballot(my_nt > 1 && my_nt <= MERGE_GROUP_MAX);
    if (few) {                                                          // wave-uniform
      const bool mineFew = my_nt > 1 && my_nt <= MERGE_GROUP_MAX;
      uint32_t most = mineFew ? my_nt : 0u;                             // the longest piece list among the tile's groups: the loop's trip count
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(most, d); most = most > o ? most : o; }
      if (mineFew) coop_load<F>(g, r, K::AX, partial + (size_t)my_t0 * XYW);
#pragma unroll 1
      for (uint32_t k = 1; k < most; ++k) {
        const bool act = mineFew && k < my_nt;
        if (act) coop_load<F>(g, r, K::BX, partial + ((size_t)my_t0 + k) * XYW);
        coop_sync();
        coop_add<F>(g, r, act);
      }
      if (mineFew) coop_store<F>(g, r, sums + mine * XYW);
      coop_sync();
    }
    unsigned long long todo = __ballot(my_nt > MERGE_GROUP_MAX && r == 0);
    while (todo) {                                                      // wave-uniform: one bucket with many pieces at a time, the whole block on it
      const int l = __ffsll((long long)todo) - 1; todo &= todo - 1;
      const size_t b = tile + (size_t)(l >> 2) * ntiles;
      const uint32_t t0 = task_off[b], nt = task_off[b + 1] - t0;
      const unsigned long long listed = __ballot(my_hot == (uint32_t)b);
      // a listed hot bucket: the HOT_FAN block sums of k_merge_hot; any other: its partials
      coop_block_sum<F, RED_NG>(lds, grp, r, listed ? hot_part : partial, listed ? (size_t)(__ffsll((long long)listed) - 1) * HOT_FAN : (size_t)t0, 1, listed ? (size_t)HOT_FAN : (size_t)nt);
      if (grp == 0) coop_store<F>(lds, r, sums + b * XYW);
      coop_sync();
    }
  }
}

// Both marginals of the NHI x NLO bucket matrix in ONE launch: blocks [0,NLO) produce the column sums C_lo = sum_hi S[hi][lo];
// blocks [NLO, NLO + RS*NHI) the row sums, every row cut into RS pieces (rowsum[RS*hi + piece]) so that row and column blocks carry chains of
// the same length — with NLO = 1024, NHI = 512 and RS = 2 all 2048 blocks sum 512 points: 32 per group + the tree.
template <class F>
__global__ void __launch_bounds__(RED_TPB) ZKT_RED_ATTR k_marginals(const uint32_t* __restrict__ in, size_t NLO, size_t NHI, int RS,
                                                       uint32_t* __restrict__ colsum, uint32_t* __restrict__ rowsum) {
  ZKT_SIDE_PRIO;
  typedef Coop<F> K; constexpr int XYW = 4 * Coord<F>::CW;
  __shared__ __attribute__((aligned(16))) uint32_t lds[RED_NG * K::GW];
  const int grp = threadIdx.x >> 2, r = threadIdx.x & 3;
  in += (size_t)blockIdx.y * NLO * NHI * XYW; colsum += (size_t)blockIdx.y * 1024 * XYW; rowsum += (size_t)blockIdx.y * 1024 * XYW;   // grid.y = window (direct form)
  const bool is_col = blockIdx.x < NLO;
  const size_t o = is_col ? blockIdx.x : blockIdx.x - NLO;
  const size_t piece = NLO / RS;
  const size_t count = is_col ? NHI : piece, stride_j = is_col ? NLO : 1;
  const size_t first = is_col ? o : (o / RS) * NLO + (o % RS) * piece;
  coop_block_sum<F, RED_NG>(lds, grp, r, in, first, stride_j, count);
  if (grp == 0) coop_store<F>(lds, r, (is_col ? colsum : rowsum) + o * XYW);
}

// Bit classes of both weighted sums: class `bit` of A = sum of the column sums whose weight lo + 1 has that bit, class `bit` of B = sum of the row pieces
// whose weight hi = index / RS has it.  The members of a class are ENUMERATED (the q-th weight with bit `bit` set is q with a one inserted at
// that position), so no lane walks the non-members; every class is cut into WB_SPLIT blocks (cls[(bit * WB_SPLIT + part)]), which k_combine adds up.
static constexpr int WB_TPB = 128, WB_NG = WB_TPB / 4, WB_SPLIT = 4;
__device__ inline uint32_t insert_one(uint32_t q, int bit) { return ((q >> bit) << (bit + 1)) | (1u << bit) | (q & ((1u << bit) - 1u)); }
template <class F>
__global__ void __launch_bounds__(WB_TPB) ZKT_RED_ATTR k_weight_bits(const uint32_t* __restrict__ colsum, size_t NLO, int nbA,
                                                        const uint32_t* __restrict__ rowsum, size_t NHI, int RS, uint32_t* __restrict__ clsA, uint32_t* __restrict__ clsB) {
  ZKT_SIDE_PRIO;
  typedef Coop<F> K; constexpr int XYW = 4 * Coord<F>::CW;
  __shared__ __attribute__((aligned(16))) uint32_t lds[WB_NG * K::GW];
  const int grp = threadIdx.x >> 2, r = threadIdx.x & 3;
  colsum += (size_t)blockIdx.y * 1024 * XYW; rowsum += (size_t)blockIdx.y * 1024 * XYW;
  clsA += (size_t)blockIdx.y * 32 * WB_SPLIT * XYW; clsB += (size_t)blockIdx.y * 32 * WB_SPLIT * XYW;
  const int cls = blockIdx.x / WB_SPLIT, part = blockIdx.x % WB_SPLIT;
  const bool isA = cls < nbA;
  const int bit = isA ? cls : cls - nbA;
  // members of the class: A, bit < lo_bits: weights insert_one(q, bit), q < NLO/2 (entry = weight - 1); bit == lo_bits: the single weight NLO.
  //                       B: rows hi = insert_one(q, bit), q < NHI/2, all RS pieces of each
  const bool topA = isA && ((size_t)1 << bit) == NLO;
  const size_t members = isA ? (topA ? 1 : NLO / 2) : (NHI / 2) * (size_t)RS;
  uint32_t* g = lds + grp * K::GW;
  coop_init<F>(g, r);
  coop_sync();
  const size_t per = (members + WB_SPLIT - 1) / WB_SPLIT, lo = (size_t)part * per, hi = lo + per < members ? lo + per : members;
#pragma unroll 1
  for (size_t m0 = lo; m0 < hi; m0 += WB_NG) {
    const size_t m = m0 + grp; const bool act = m < hi;
    if (act) {
      size_t e;
      if (isA) e = topA ? NLO - 1 : (size_t)insert_one((uint32_t)m, bit) - 1;
      else e = (size_t)insert_one((uint32_t)(m / RS), bit) * RS + m % RS;
      coop_load<F>(g, r, K::BX, (isA ? colsum : rowsum) + e * XYW);
    }
    coop_sync();
    coop_add<F>(g, r, act);
  }
  coop_block_tree<F, WB_NG>(lds, grp, r);
  if (grp == 0) coop_store<F>(lds, r, (isA ? clsA : clsB) + ((size_t)bit * WB_SPLIT + part) * XYW);
}

// total = sum_{b<=shift} 2^b A_b + 2^shift sum_b 2^b B_b = sum_t 2^t D_t with D_t = A_t (t<=shift) (+) B_{t-shift} (t>=shift).
// One block of 32 groups: group t sums the WB_SPLIT parts of its classes, doubles D_t t times (<= 19 three-round doublings instead of a 40-step
// serial Horner), then the tree; lane 0 writes the Jacobian sum and its affine normalisation (the one inversion of an MSM).
static constexpr int CMB_TPB = 128, CMB_NG = CMB_TPB / 4;
template <class F>
__global__ void __launch_bounds__(CMB_TPB) k_combine(const uint32_t* __restrict__ clsA, int nbA, const uint32_t* __restrict__ clsB, int nbB, int shift,
                                                     uint32_t* __restrict__ out_jac, uint32_t* __restrict__ out_abi) {
  ZKT_SIDE_PRIO;
  typedef Coop<F> K; constexpr int CW = Coord<F>::CW, XYW = 4 * CW;
  __shared__ __attribute__((aligned(16))) uint32_t lds[CMB_NG * K::GW];
  const int t = threadIdx.x >> 2, r = threadIdx.x & 3;
  clsA += (size_t)blockIdx.y * 32 * WB_SPLIT * XYW; clsB += (size_t)blockIdx.y * 32 * WB_SPLIT * XYW; out_jac += (size_t)blockIdx.y * XYW;      // window results are XYW apart
  uint32_t* g = lds + t * K::GW;
  coop_init<F>(g, r);
  coop_sync();
  const bool hasA = t < nbA, hasB = t >= shift && t - shift < nbB;
#pragma unroll 1
  for (int k = 0; k < 2 * WB_SPLIT; ++k) {                              // D_t: the parts of A_t, then the parts of B_{t - shift}
    const bool act = k < WB_SPLIT ? hasA : hasB;
    if (act) coop_load<F>(g, r, K::BX, (k < WB_SPLIT ? clsA + ((size_t)t * WB_SPLIT + k) * XYW : clsB + ((size_t)(t - shift) * WB_SPLIT + (k - WB_SPLIT)) * XYW));
    coop_sync();
    coop_add<F>(g, r, act);
  }
  const int top = nbA > shift + nbB ? nbA : shift + nbB;                // classes in use: doublings run to top - 1
#pragma unroll 1
  for (int d = 0; d + 1 < top; ++d) coop_dbl<F>(g, r, d < t);
  coop_block_tree<F, CMB_NG>(lds, t, r);
  if (threadIdx.x == 0) {
    Xyzz<F> v; v.X = K::ld(lds, K::AX); v.Y = K::ld(lds, K::AY); v.ZZ = K::ld(lds, K::AZZ); v.ZZZ = K::ld(lds, K::AZZZ);
    Jac<F> j = xyzz_to_jac<F>(v);
    Coord<F>::st(out_jac, j.X); Coord<F>::st(out_jac + CW, j.Y); Coord<F>::st(out_jac + 2 * CW, j.Z);
    if (out_abi) PtIO<F>::st(out_abi, xyzz_to_aff<F>(v));
  }
}

}  // namespace zkt
