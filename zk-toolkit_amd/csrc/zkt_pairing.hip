// Pairing kernels other than the plain batched Tate pairing (zkt_tate.hip): raw Miller values / Weil (row a14), the fused
// Groth16 verification (f-2) and pairing-product equalities (f-4).  The algorithm and why it is bit-identical to the reference are in pairing.h.
#include <cstdlib>
#include <mutex>
#include <atomic>
#include "abi.h"
#include "zkt_internal.h"

namespace zkt {

// raw Miller values / Weil (row a14): which = 0 calc_g1_g2, 1 calc_g2_g1, 2 weil (pairing.rs:54-55,75-84)
__global__ void __launch_bounds__(64) k_miller_exact(int which, const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g2,
                                                     uint32_t* __restrict__ out, size_t n, unsigned long long* err) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  Aff<FqOps> p = PtIO<FqOps>::ld(g1 + i * ABI_G1_WORDS);
  Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(g2 + i * ABI_G2_WORDS);
  if (p.inf || q.inf) { atomicMin(err, (unsigned long long)i); return; }
  Fq12 r; bool bad = false;
  if (which == 0) r = miller_g1_g2_exact(p.x, p.y, q.x, q.y, bad);
  else if (which == 1) r = miller_g2_g1_exact(q.x, q.y, p.x, p.y);
  else r = fq12_mul(miller_g1_g2_exact(p.x, p.y, q.x, q.y, bad), fq12_inv(miller_g2_g1_exact(q.x, q.y, p.x, p.y)));
  if (bad) { atomicMin(err, (unsigned long long)i); return; }          // the reference's panic on a multiple of P at infinity
  st_fq12(out + i * 144, r);
}
hipError_t launch_miller_exact(int which, const uint32_t* g1, const uint32_t* g2, uint32_t* out, size_t n, unsigned long long* err, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_miller_exact, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, which, g1, g2, out, n, err);
  return hipGetLastError();
}

// ---- preconditions of the 127-step loop for the small-batch kernels -----------------------------------------------------------------
// The lane-distributed kernels (zkt_dpairing.hip) run the 127-step loop on trust; whether every pair of an element fits it (points on their
// curves, Q in G2: pairing_args_fit_short_loop) is decided by this one-lane-per-element kernel on a side stream at the same time — the G2
// test alone is 1.8 ms on a single lane, about a third of the pairing it guards.  Elements that fail are redone by the 255-step kernels.
__global__ void __launch_bounds__(64) k_short_loop_guards(PairArgs a, int K, uint32_t* __restrict__ flags, size_t n) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  uint32_t fits = 1;
  for (int k = 0; k < K && fits; ++k) {
    Aff<FqOps> p = PtIO<FqOps>::ld(a.g1[k] + i * a.s1[k]);
    Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(a.g2[k] + i * a.s2[k]);
    if (p.inf || q.inf) continue;                                  // the reference's panic: reported by the main kernel, nothing to redo
    fits = pairing_args_fit_short_loop<1>(&p.x, &p.y, &q.x, &q.y) ? 1u : 0u;
  }
  flags[i] = fits;
}
// the same for the 63-step loop of the lane-distributed kernels (k_dproduct_ate): P on E and in G1 (the chain on P is gone, so its membership is tested here, 127
// doublings on E), Q on E'; Q in G2 comes out of the loop itself
// One lane per (element, pair), four lanes per element: the membership test of a P is a 127-doubling chain on one lane (1.7 ms), and with the K of them in one lane
// this kernel, not the pairing beside it, was the critical path of a single verification.  p_skip: bit k = pair k's P needs no test (a sum of multiples of key points that
// were tested with the key).
__global__ void __launch_bounds__(64) k_ate_guards(PairArgs a, int K, uint32_t* __restrict__ flags, size_t n, uint32_t p_skip) {
  const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x, i = t >> 2;
  const int k = (int)(t & 3);
  bool bad = false;
  if (i < n && k < K) {
    Aff<FqOps> p = PtIO<FqOps>::ld(a.g1[k] + i * a.s1[k]);
    Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(a.g2[k] + i * a.s2[k]);
    if (!p.inf && !q.inf)                                          // the reference's panic on infinity: reported by the main kernel, nothing to redo
      bad = !(g2_on_curve(q.x, q.y) && (((p_skip >> k) & 1) || (g1_on_curve(p.x, p.y) && g1_in_subgroup(p.x, p.y))));
  }
  const unsigned long long b = __ballot(bad);
  if (i < n && k == 0) flags[i] = ((b >> (threadIdx.x & ~3)) & 15ull) ? 0u : 1u;
}
static bool small_ate() { static const bool on = [] { const char* e = getenv("ZKT_PRODUCT_LOOP"); return !(e && atoi(e) == 127); }(); return on; }
hipError_t launch_ate_guards(const PairArgs& a, int K, uint32_t* flags, size_t n, hipStream_t s, uint32_t p_skip) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_ate_guards, dim3((unsigned)((4 * n + 63) / 64)), dim3(64), 0, s, a, K, flags, n, p_skip);
  return hipGetLastError();
}
hipError_t launch_short_loop_guards(const PairArgs& a, int K, uint32_t* flags, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_short_loop_guards, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, a, K, flags, n);
  return hipGetLastError();
}
namespace {
struct GuardStreams {
  // ONE side stream for the whole library, on purpose: the runtime sizes a queue's scratch for every wave slot of the device (8192) times the largest
  // per-lane frame it has seen (capped at 6 GiB), and keeps it — 2.4 GiB for a queue that ran k_short_loop_guards, 5-6 GiB for one that ran a pairing kernel.  The
  // process's scratch pool is ~30 GB; spreading pairing-family kernels over more queues exhausts it and the runtime aborts the process
  // (HSA_STATUS_ERROR_OUT_OF_RESOURCES in its queue-error callback), which is how a second side stream and two verifier streams ended the GPU suite.
  std::mutex mu; hipStream_t side = nullptr; hipEvent_t ev[32] = {}; unsigned next = 0;
  hipError_t event(hipEvent_t* e) {                                // a small ring: a wait captures the event's record at the time of the call
    std::lock_guard<std::mutex> lk(mu);
    hipError_t rc;
    if (!side && (rc = hipStreamCreateWithFlags(&side, hipStreamNonBlocking)) != hipSuccess) return rc;
    hipEvent_t& x = ev[next++ % 32];
    if (!x && (rc = hipEventCreateWithFlags(&x, hipEventDisableTiming)) != hipSuccess) return rc;
    *e = x; return hipSuccess;
  }
  void release() {                                                 // zkt_shutdown: stream and events belong to the device of that zkt_init
    std::lock_guard<std::mutex> lk(mu);
    if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); side = nullptr; }
    for (hipEvent_t& x : ev) if (x) { (void)hipEventDestroy(x); x = nullptr; }
    next = 0;
  }
} g_guard;
}  // namespace
void pairing_release_device_state() { g_guard.release(); }
hipError_t guard_fork(hipStream_t s, hipStream_t* side) {
  hipEvent_t e; hipError_t rc;
  if ((rc = g_guard.event(&e)) != hipSuccess) return rc;
  *side = g_guard.side;
  if ((rc = hipEventRecord(e, s)) != hipSuccess) return rc;
  return hipStreamWaitEvent(*side, e, 0);
}
hipError_t guard_join(hipStream_t s, hipStream_t side) {
  hipEvent_t e; hipError_t rc;
  if ((rc = g_guard.event(&e)) != hipSuccess) return rc;
  if ((rc = hipEventRecord(e, side)) != hipSuccess) return rc;
  return hipStreamWaitEvent(s, e, 0);
}
// ok[i] <- OK_REDO where the guards say the element does not fit the 127-step loop
static constexpr uint32_t OK_REDO = 2;
// ok[i] <- OK_EXACT: a G1 argument outside the order-r subgroup, or a point off its curve.  There e(-P,Q) = e(P,Q)^-1 is not available and two addition chains
// for the same multiple are different functions, so "lhs == rhs as a product == 1" is not the reference's decision by construction: k_product_exact_marked
// evaluates both sides the reference's way (verifier.rs:36-53, signature.rs:34-39).  Round 3 failed such elements closed; zkt_verify_set_fail_closed(1) brings that back.
static constexpr uint32_t OK_EXACT = 3;
__global__ void __launch_bounds__(256) k_product_resolve(const uint32_t* __restrict__ flags, uint32_t* __restrict__ ok, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n && !flags[i]) ok[i] = OK_REDO;
}

// Groth16 verification, one proof per lane (verifier.rs:30-54 / SURVEY §8 f-2):
//   e(A,B) == alpha_beta * e(S,gamma) * e(C,delta)   <=>   tate-product(A,B; -S,gamma; -C,delta) == alpha_beta
// (e(-P,Q) = e(P,Q)^-1 exactly).  S_i = sum_j stmt[i][j] * uvw_stmt[j] is formed here too.  ok[i] = 1 / 0.
// G2 points shared by a whole batch (stride 0: gamma and delta of a verifying key): curve equation and subgroup membership once per launch.
// good[0] bit j <- points[j] is in G2.  One wave; lanes beyond `count` idle.
__global__ void __launch_bounds__(64) k_shared_g2_guards(const uint32_t* __restrict__ p0, const uint32_t* __restrict__ p1, uint32_t* __restrict__ good) {
  const int j = threadIdx.x;
  bool ok = false;
  if (j < 2) {
    Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(j == 0 ? p0 : p1);
    ok = !q.inf && g2_on_curve(q.x, q.y) && g2_in_subgroup(q.x, q.y);
  }
  const unsigned long long b = __ballot(ok);
  if (j == 0) good[0] = (uint32_t)(b & 3u);
}
// ok[i] = OK_REDO: the preconditions of the 127-step loop (pairing.h) do not hold for element i; the 255-step kernel, launched behind with
// only_redo = 1, decides it.  Honest batches never see that second pass (one load per lane).
template <bool SHORT>
__global__ void __launch_bounds__(64) k_groth16_verify(const uint32_t* __restrict__ A, const uint32_t* __restrict__ B, const uint32_t* __restrict__ C,
                                                       const uint32_t* __restrict__ uvw_stmt, const uint32_t* __restrict__ stmt, int n_stmt,
                                                       const uint32_t* __restrict__ gamma, const uint32_t* __restrict__ delta,
                                                       const uint32_t* __restrict__ alpha_beta, uint32_t* __restrict__ ok, size_t n, unsigned long long* err, int only_redo,
                                                       const uint32_t* __restrict__ shared_good, const uint32_t* __restrict__ S_pre) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  if (only_redo && ok[i] != OK_REDO) return;
  Aff<FqOps> S;
  if (S_pre) S = PtIO<FqOps>::ld(S_pre + i * ABI_G1_WORDS);            // the statement sums came from one batched scalar multiplication (launch_groth16_verify)
  else {
    Jac<FqOps> acc = jac_inf<FqOps>();
    for (int j = 0; j < n_stmt; ++j) {                                 // verifier.rs:41-45
      Aff<FqOps> u = PtIO<FqOps>::ld(uvw_stmt + (size_t)j * ABI_G1_WORDS);
      acc = jac_add(acc, scalar_mul_aff<FqOps>(u, stmt + ((size_t)i * n_stmt + j) * 8, 8));
    }
    S = jac_to_aff(acc);
  }
  Aff<FqOps> a = PtIO<FqOps>::ld(A + i * ABI_G1_WORDS), c = PtIO<FqOps>::ld(C + i * ABI_G1_WORDS);
  Aff<Fq2Ops> b = PtIO<Fq2Ops>::ld(B + i * ABI_G2_WORDS), g = PtIO<Fq2Ops>::ld(gamma), d = PtIO<Fq2Ops>::ld(delta);
  if (a.inf || b.inf || c.inf || S.inf || g.inf || d.inf) { atomicMin(err, (unsigned long long)i); ok[i] = 0; return; }   // tate() with infinity panics
  Fq xp[3] = {a.x, S.x, c.x}, yp[3] = {a.y, fp_neg(S.y), fp_neg(c.y)};
  Fq2 xq[3] = {b.x, g.x, d.x}, yq[3] = {b.y, g.y, d.y};
  bool in_g1;
  Fq12 e;
  if constexpr (SHORT) {
    if (!pairing_args_fit_short_loop<3>(xp, yp, xq, yq, shared_good ? (shared_good[0] & 3u) << 1 : 0u)) { ok[i] = OK_REDO; return; }      // gamma, delta: pairs 1, 2
    e = final_exponentiation_t<true>(miller_g1_g2_multi_short<3>(xp, yp, xq, yq, in_g1));      // the Tate product itself: comparable with alpha_beta
  } else {
    for (int k = 0; k < 3; ++k) if (!g1_on_curve(xp[k], yp[k]) || !g2_on_curve(xq[k], yq[k])) { ok[i] = OK_EXACT; return; }      // off its curve: only the reference's own chain gives the reference's value
    e = final_exponentiation(miller_g1_g2_multi<3>(xp, yp, xq, yq, in_g1));
  }
  if (!in_g1) { ok[i] = OK_EXACT; return; }       // a G1 argument outside the order-r subgroup: e(-P,Q) = e(P,Q)^-1 is not available — both sides evaluated the reference's way behind
  uint32_t got[144]; st_fq12(got, e);
  uint32_t diff = 0;
  for (int k = 0; k < 144; ++k) diff |= got[k] ^ alpha_beta[k];
  ok[i] = diff == 0;
}
// S_i = sum_j terms[j*n + i] for the small-batch verification path (the terms stmt[i][j] * uvw_stmt[j] come from one batched
// scalar multiplication, one lane per term, instead of n_stmt serial ones in the proof's lane)
__global__ void __launch_bounds__(64) k_stmt_sums(const uint32_t* __restrict__ terms, int n_stmt, uint32_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  Jac<FqOps> acc = jac_inf<FqOps>();
  for (int j = 0; j < n_stmt; ++j) acc = jac_add_aff(acc, PtIO<FqOps>::ld(terms + ((size_t)j * n + i) * ABI_G1_WORDS));
  PtIO<FqOps>::st(out + i * ABI_G1_WORDS, jac_to_aff(acc));
}
// ---- the deciding entry points on the 63-step loop (pairing.h, "optimal ate") ---------------------------------------------------------
// What a verifying key contributes, prepared ONCE per key (the host caches it by the key's bytes, zkt_protocols.hip ate_key_begin):
//   words [0, T)      the 68 line triples of gamma          T = ATE_LINES * ATE_LINE_WORDS
//   words [T, 2T)     ... of delta
//   words [2T, +144)  final_exponentiation(f_{|x|,beta}(alpha)) — the ate counterpart of the key's alpha_beta (compared in the ABI's Fq12 layout)
//   word  2T + 144    bit 0 gamma in G2, bit 1 delta in G2, bit 2 alpha in G1 and beta on E', bit 3 every statement point in G1, bit 4 beta in G2 (k_key_ab, which writes the pairing)
// The host uses the 63-step kernel for a key only when all five bits are set AND the key's alpha_beta equals tate(alpha, beta) (one small-batch pairing).
__global__ void __launch_bounds__(64) k_ate_key_prep(const uint32_t* __restrict__ alpha, const uint32_t* __restrict__ beta, const uint32_t* __restrict__ gamma,
                                                     const uint32_t* __restrict__ delta, const uint32_t* __restrict__ uvw_stmt, int n_stmt, uint32_t* __restrict__ key) {
  constexpr size_t T = (size_t)ATE_LINES * ATE_LINE_WORDS;
  const int j = threadIdx.x;
  if (blockIdx.x == 0) {
    if (j < 2) {
      Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(j == 0 ? gamma : delta);
      if (!q.inf && g2_on_curve(q.x, q.y) && ate_line_table(q.x, q.y, key + j * T)) atomicOr(key + 2 * T + 144, 1u << j);
    }
  } else if (blockIdx.x == 1) {
    if (j == 0) {       // alpha on E and in G1, beta on E'; beta in G2 and the pairing itself come from k_key_ab (bit 4), one lane group instead of this one lane
      Aff<FqOps> p = PtIO<FqOps>::ld(alpha); Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(beta);
      if (!p.inf && !q.inf && g1_on_curve(p.x, p.y) && g1_in_subgroup(p.x, p.y) && g2_on_curve(q.x, q.y)) atomicOr(key + 2 * T + 144, 4u);
    }
  } else {
    bool ok = true;
    if (j < n_stmt) {
      Aff<FqOps> p = PtIO<FqOps>::ld(uvw_stmt + (size_t)j * ABI_G1_WORDS);
      ok = p.inf || (g1_on_curve(p.x, p.y) && g1_in_subgroup(p.x, p.y));      // infinity is the neutral element: a statement point may be it
    }
    if (__ballot(!ok) == 0 && j == 0 && n_stmt <= 64) atomicOr(key + 2 * T + 144, 8u);
  }
}
// (the verdict word is zeroed by the caller; the pairing a(beta, alpha), bit 4 and tate(alpha, beta) come from launch_key_ab, zkt_dpairing.hip)
hipError_t launch_ate_key_prep(const uint32_t* alpha, const uint32_t* beta, const uint32_t* gamma, const uint32_t* delta, const uint32_t* uvw_stmt, int n_stmt,
                               uint32_t* key, hipStream_t s) {
  hipLaunchKernelGGL(k_ate_key_prep, dim3(3), dim3(64), 0, s, alpha, beta, gamma, delta, uvw_stmt, n_stmt, key);
  return hipGetLastError();
}
// e(A,B) == alpha_beta e(S,gamma) e(C,delta)  <=>  a(B,A) a(gamma,-S) a(delta,-C) == a(beta,alpha): B runs its chain in the lane (its G2 test comes with it),
// gamma and delta bring their tables, S is a sum of multiples of statement points tested with the key.  A and C are tested here (127 doublings each).
// Whatever does not fit — a point off its curve or outside its group — is marked OK_REDO and decided by the older kernels behind this one.
__global__ void __launch_bounds__(64) k_groth16_verify_ate(const uint32_t* __restrict__ A, const uint32_t* __restrict__ B, const uint32_t* __restrict__ C,
                                                           const uint32_t* __restrict__ S_pre, const uint32_t* __restrict__ key,
                                                           uint32_t* __restrict__ ok, size_t n, unsigned long long* err, const uint32_t* __restrict__ fits) {
  constexpr size_t T = (size_t)ATE_LINES * ATE_LINE_WORDS;
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  Aff<FqOps> S = PtIO<FqOps>::ld(S_pre + i * ABI_G1_WORDS), a = PtIO<FqOps>::ld(A + i * ABI_G1_WORDS), c = PtIO<FqOps>::ld(C + i * ABI_G1_WORDS);
  Aff<Fq2Ops> b = PtIO<Fq2Ops>::ld(B + i * ABI_G2_WORDS);
  if (a.inf || b.inf || c.inf || S.inf) { atomicMin(err, (unsigned long long)i); ok[i] = 0; return; }   // tate() with infinity panics (gamma, delta: tested with the key)
  if (!fits[i] || !fits[n + i] || !g2_on_curve(b.x, b.y)) { ok[i] = OK_REDO; return; }       // fits: A, C on E and in G1 (k_g1_fits, run at four waves per SIMD before this kernel)
  Fq xp[3] = {a.x, S.x, c.x}, yp[3] = {a.y, fp_neg(S.y), fp_neg(c.y)};
  const uint32_t* tabs[2] = {key, key + T};
  bool in_g2;
  Fq12 f = miller_ate_multi<1, 2>(xp, yp, &b.x, &b.y, tabs, in_g2);
  if (!in_g2) { ok[i] = OK_REDO; return; }
  uint32_t got[144]; st_fq12(got, final_exponentiation_3h(f));      // three times the exact exponent: decisions only (pairing.h)
  uint32_t diff = 0;
  for (int k = 0; k < 144; ++k) diff |= got[k] ^ key[2 * T + k];
  ok[i] = diff == 0;
}
// G1 points shared by a whole batch (stride 0: the generator in signature verification): curve equation and subgroup membership once per launch.
__global__ void __launch_bounds__(64) k_shared_g1_guards(PairArgs a, int K, uint32_t* __restrict__ good) {
  const int k = threadIdx.x;
  bool ok = false;
  if (k < K && a.s1[k] == 0) {
    Aff<FqOps> p = PtIO<FqOps>::ld(a.g1[k]);
    ok = !p.inf && g1_on_curve(p.x, p.y) && g1_in_subgroup(p.x, p.y);
  }
  const unsigned long long b = __ballot(ok);
  if (k == 0) good[0] = (uint32_t)(b & 15u);
}
// prod_k tate(+-P_k, Q_k) == 1 decided on the 63-step loop: every Q_k runs its chain in the lane and is tested where it ends, every P_k is tested first
// (bit k of p_good[0]: P_k is shared by the batch and was tested by k_shared_g1_guards).  ok[i] = OK_REDO leaves the element to the kernels behind.
template <int K>
__global__ void __launch_bounds__(64) k_pairing_product_check_ate(PairArgs a, uint32_t* __restrict__ ok, size_t n, unsigned long long* err, const uint32_t* __restrict__ p_good, uint32_t p_trusted,
                                                                  const uint32_t* __restrict__ fits) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  Fq xp[K], yp[K]; Fq2 xq[K], yq[K];
  bool inf = false;
  for (int k = 0; k < K; ++k) {
    Aff<FqOps> p = PtIO<FqOps>::ld(a.g1[k] + i * a.s1[k]);
    Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(a.g2[k] + i * a.s2[k]);
    inf = inf || p.inf || q.inf;
    xp[k] = p.x; yp[k] = a.neg[k] ? fp_neg(p.y) : p.y; xq[k] = q.x; yq[k] = q.y;
  }
  if (inf) { atomicMin(err, (unsigned long long)i); ok[i] = 0; return; }
  const uint32_t known = p_trusted | (p_good ? p_good[0] : 0u);      // p_trusted: slots whose P is a constant of the library (the G1 generator in signature verification)
  for (int k = 0; k < K; ++k) {
    const bool p_ok = ((known >> k) & 1) || fits[(size_t)k * n + i] != 0;      // fits: k_g1_fits, run before this kernel for the slots that are not shared
    if (!p_ok || !g2_on_curve(xq[k], yq[k])) { ok[i] = OK_REDO; return; }
  }
  bool in_g2;
  const uint32_t* none[1] = {nullptr};
  Fq12 f = miller_ate_multi<K, 0>(xp, yp, xq, yq, none, in_g2);
  if (!in_g2) { ok[i] = OK_REDO; return; }
  uint32_t got[144]; st_fq12(got, final_exponentiation_3h(f));      // three times the exact exponent: decisions only (pairing.h)
  uint32_t diff = got[132] ^ 1u;
  for (int k = 0; k < 144; ++k) if (k != 132) diff |= got[k];
  ok[i] = diff == 0;
}

// ---- elements marked OK_EXACT: the reference's own evaluation ------------------------------------------------------------------------
// Verifier::verify (verifier.rs:36-53): lhs = tate(A, B); rhs = alpha_beta * tate(sum, gamma) * tate(C, delta); lhs == rhs.  Signer::verify (signature.rs:34-39):
// tate(g1, sig) == tate(pk, H(m)).  Every pairing through the reference's chain (miller_g1_g2_exact: binary chain, vertical lines, its panics) and the exact final
// exponentiation, the pairs with neg = 0 multiplied into the left side, those with neg = 1 (taken WITHOUT the negation) and the target into the right side, then
// Fq12 equality on canonical words.  A panic of the reference (a multiple of P at infinity, a vanishing denominator) is the batch's error index.
// 17 KB of scratch per lane: launched only when a read-back says an element is marked (finish_exact).
__global__ void __launch_bounds__(64) k_product_exact_marked(PairArgs a, int K, const uint8_t* __restrict__ kcount, const uint32_t* __restrict__ target,
                                                             uint32_t* __restrict__ ok, size_t n, unsigned long long* err) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n || ok[i] != OK_EXACT) return;
  Fq12 side[2] = {fq12_one(), target ? ld_fq12(target) : fq12_one()}, t;
  const int kc = kcount ? (int)kcount[i] : K;
  for (int k = 0; k < kc; ++k) {
    const Aff<FqOps> p = PtIO<FqOps>::ld(a.g1[k] + i * a.s1[k]);
    const Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(a.g2[k] + i * a.s2[k]);
    bool bad;
    const Fq12 f = miller_g1_g2_exact(p.x, p.y, q.x, q.y, bad);
    if (bad) { atomicMin(err, (unsigned long long)i); ok[i] = 0; return; }
    Fq12 v;
    if (fq12_is_zero(f)) { v = fq12_one(); v.c0.c0 = fq2_zero(); }      // 0^e = 0 (fq12.rs:42-57)
    else v = final_exponentiation(f);
    const int sd = a.neg[k] ? 1 : 0;
    t = fq12_mul(side[sd], v); side[sd] = t;
  }
  uint32_t l[144], r[144]; st_fq12(l, side[0]); st_fq12(r, side[1]);
  uint32_t diff = 0;
  for (int k = 0; k < 144; ++k) diff |= l[k] ^ r[k];
  ok[i] = diff == 0;
}
__global__ void __launch_bounds__(256) k_count_exact(const uint32_t* __restrict__ ok, size_t n, uint32_t* __restrict__ count, int to_zero) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n || ok[i] != OK_EXACT) return;
  if (to_zero) const_cast<uint32_t*>(ok)[i] = 0; else atomicAdd(count, 1u);
}
// S_i = sum_j stmt[i][j] * uvw_stmt[j] (verifier.rs:41-45) for the marked elements only, one lane each
__global__ void __launch_bounds__(64) k_stmt_sums_marked(const uint32_t* __restrict__ uvw_stmt, const uint32_t* __restrict__ stmt, int n_stmt, const uint32_t* __restrict__ ok,
                                                         uint32_t* __restrict__ S, size_t n) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n || ok[i] != OK_EXACT) return;
  Jac<FqOps> acc = jac_inf<FqOps>();
  for (int j = 0; j < n_stmt; ++j) acc = jac_add(acc, scalar_mul_aff<FqOps>(PtIO<FqOps>::ld(uvw_stmt + (size_t)j * ABI_G1_WORDS), stmt + ((size_t)i * n_stmt + j) * 8, 8));
  PtIO<FqOps>::st(S + i * ABI_G1_WORDS, jac_to_aff(acc));
}
static std::atomic<int> g_fail_closed{[] { const char* e = getenv("ZKT_VERIFY_FAIL_CLOSED"); return e && atoi(e) ? 1 : 0; }()};
void verify_set_fail_closed(int on) { g_fail_closed.store(on ? 1 : 0); }
int verify_fail_closed() { return g_fail_closed.load(); }
// The last step of every deciding launch sequence: elements the kernels before left marked OK_EXACT.  One 4-byte read-back (the stream is synchronised here: the
// entry points block on their result anyway); honest batches stop there.  fail-closed mode: the marks become rejections, no read-back.
static hipError_t finish_exact(const PairArgs& a, int K, const uint8_t* kcount, const uint32_t* target, uint32_t* ok, size_t n, unsigned long long* err, hipStream_t s) {
  if (n == 0) return hipSuccess;
  const dim3 g256((unsigned)((n + 255) / 256));
  if (verify_fail_closed()) { hipLaunchKernelGGL(k_count_exact, g256, dim3(256), 0, s, (const uint32_t*)ok, n, (uint32_t*)nullptr, 1); return hipGetLastError(); }
  uint32_t* d = nullptr; uint32_t host = 0; hipError_t e;
  if ((e = hipMallocAsync((void**)&d, 4, s)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(d, 0, 4, s)) != hipSuccess) { (void)hipFreeAsync(d, s); return e; }
  hipLaunchKernelGGL(k_count_exact, g256, dim3(256), 0, s, (const uint32_t*)ok, n, d, 0);
  if ((e = hipMemcpyAsync(&host, d, 4, hipMemcpyDeviceToHost, s)) != hipSuccess) { (void)hipFreeAsync(d, s); return e; }
  if ((e = hipFreeAsync(d, s)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
  if (host) hipLaunchKernelGGL(k_product_exact_marked, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, a, K, kcount, target, ok, n, err);
  return hipGetLastError();
}
static PairArgs groth16_pairs(const uint32_t* A, const uint32_t* B, const uint32_t* C, const uint32_t* S, const uint32_t* gamma, const uint32_t* delta) {
  PairArgs a{};
  a.g1[0] = A; a.s1[0] = ABI_G1_WORDS; a.g2[0] = B; a.s2[0] = ABI_G2_WORDS; a.neg[0] = 0;
  a.g1[1] = S; a.s1[1] = ABI_G1_WORDS; a.g2[1] = gamma; a.s2[1] = 0; a.neg[1] = 1;
  a.g1[2] = C; a.s1[2] = ABI_G1_WORDS; a.g2[2] = delta; a.s2[2] = 0; a.neg[2] = 1;
  return a;
}

// e(A,B) == alpha_beta e(S,gamma) e(C,delta) per proof on the lane-distributed kernels; tmp: n_stmt * n G1 points, S: n G1 points (device)
hipError_t launch_groth16_verify_small(const uint32_t* A, const uint32_t* B, const uint32_t* C, const uint32_t* uvw_stmt, const uint32_t* stmt_tables, const uint32_t* stmt, int n_stmt,
                                       const uint32_t* gamma, const uint32_t* delta, const uint32_t* alpha_beta, uint32_t* tmp, uint32_t* S, uint32_t* ok, size_t n,
                                       unsigned long long* err, hipStream_t s, const uint32_t* ate_target) {
  if (n == 0) return hipSuccess;
  if (n_stmt < 1 || n_stmt > 12) return hipErrorInvalidValue;
  hipError_t e;
  if (stmt_tables) e = launch_fixed_muls_batch(G_G1, stmt_tables, stmt, tmp, n, n_stmt, s);      // one wave per term from the key's fixed-base tables: ~0.3 ms instead of a 4 ms chain
  else {
    MulSegs segs; segs.n = n_stmt;
    for (int j = 0; j < n_stmt; ++j) segs.s[j] = MulSeg{uvw_stmt + (size_t)j * ABI_G1_WORDS, stmt + (size_t)j * 8, tmp + (size_t)j * n * ABI_G1_WORDS, (uint32_t)n, 0u, (uint32_t)(n_stmt * 8)};
    e = launch_group_mul_segs(G_G1, segs, 8, s);
  }
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_stmt_sums, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, (const uint32_t*)tmp, n_stmt, S, n);
  PairArgs a{};
  a.g1[0] = A; a.s1[0] = ABI_G1_WORDS; a.g2[0] = B; a.s2[0] = ABI_G2_WORDS; a.neg[0] = 0;
  a.g1[1] = S; a.s1[1] = ABI_G1_WORDS; a.g2[1] = gamma; a.s2[1] = 0; a.neg[1] = 1;
  a.g1[2] = C; a.s1[2] = ABI_G1_WORDS; a.g2[2] = delta; a.s2[2] = 0; a.neg[2] = 1;
  // 127-step loops on the lane-distributed kernels, their preconditions checked beside them; what fails is redone by the 255-step kernel
  uint32_t* flags = nullptr; hipStream_t side;
  if ((e = hipMallocAsync((void**)&flags, n * sizeof(uint32_t), s)) != hipSuccess) return e;
  // ate_target (the key's ate counterpart of alpha_beta, k_ate_key_prep): the 63-step loop, its guards beside it; otherwise the 127-step loop against alpha_beta itself
  const bool ate = ate_target && small_ate();
  if ((e = guard_fork(s, &side)) != hipSuccess || (e = (ate ? launch_ate_guards(a, 3, flags, n, side, 2u) : launch_short_loop_guards(a, 3, flags, n, side))) != hipSuccess ||
      (e = (ate ? launch_dproduct_ate(a, 3, ate_target, ok, n, err, s) : launch_dproduct(a, 3, alpha_beta, ok, n, err, true, s))) != hipSuccess ||
      (e = guard_join(s, side)) != hipSuccess) { (void)hipFreeAsync(flags, s); return e; }
  hipLaunchKernelGGL(k_product_resolve, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const uint32_t*)flags, ok, n);
  if ((e = hipFreeAsync(flags, s)) != hipSuccess) return e;
  hipLaunchKernelGGL(k_groth16_verify<false>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, A, B, C, uvw_stmt, stmt, n_stmt, gamma, delta, alpha_beta, ok, n, err, 1, (const uint32_t*)nullptr, (const uint32_t*)S);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  return finish_exact(a, 3, nullptr, alpha_beta, ok, n, err, s);
}
hipError_t launch_groth16_verify(const uint32_t* A, const uint32_t* B, const uint32_t* C, const uint32_t* uvw_stmt, const uint32_t* stmt, int n_stmt,
                                 const uint32_t* gamma, const uint32_t* delta, const uint32_t* alpha_beta, uint32_t* ok, size_t n,
                                 unsigned long long* err, hipStream_t s, const uint32_t* ate_key) {
  if (n == 0) return hipSuccess;
  if (ate_key && n_stmt >= 1 && n_stmt <= 12) {             // a key the 63-step loop may serve (k_ate_key_prep): statement sums, the ate kernel, the older kernel for what it marked
    uint32_t* S = nullptr; hipError_t e;
    if ((e = hipMallocAsync((void**)&S, n * ABI_G1_WORDS * 4, s)) != hipSuccess) return e;
    // the statement sums from the key's 8-bit window tables (behind the line tables in the key buffer): 32 additions per wire instead of a 255-step chain
    if ((e = launch_stmt_sums_wide(ate_key + ATE_KEY_WORDS, stmt, n_stmt, S, n, s)) != hipSuccess) { (void)hipFreeAsync(S, s); return e; }
    uint32_t* fits = nullptr;
    if ((e = hipMallocAsync((void**)&fits, 2 * n * sizeof(uint32_t), s)) != hipSuccess) { (void)hipFreeAsync(S, s); return e; }
    G1Fits gf{}; gf.pts[0] = A; gf.stride[0] = ABI_G1_WORDS; gf.pts[1] = C; gf.stride[1] = ABI_G1_WORDS;
    if ((e = launch_g1_fits(gf, 2, fits, n, s)) != hipSuccess) { (void)hipFreeAsync(S, s); (void)hipFreeAsync(fits, s); return e; }
    hipLaunchKernelGGL(k_groth16_verify_ate, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, A, B, C, (const uint32_t*)S, ate_key, ok, n, err, (const uint32_t*)fits);
    if ((e = hipFreeAsync(fits, s)) != hipSuccess) { (void)hipFreeAsync(S, s); return e; }
    hipLaunchKernelGGL(k_groth16_verify<false>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, A, B, C, uvw_stmt, stmt, n_stmt, gamma, delta, alpha_beta, ok, n, err, 1, (const uint32_t*)nullptr, (const uint32_t*)S);
    if ((e = hipGetLastError()) != hipSuccess || (e = finish_exact(groth16_pairs(A, B, C, S, gamma, delta), 3, nullptr, alpha_beta, ok, n, err, s)) != hipSuccess) { (void)hipFreeAsync(S, s); return e; }
    if ((e = hipFreeAsync(S, s)) != hipSuccess) return e;
    return hipGetLastError();
  }
  uint32_t* good = nullptr; hipError_t e;                   // gamma and delta are the same for every proof: their G2 membership is decided by one wave, not by every lane
  if ((e = hipMallocAsync((void**)&good, sizeof(uint32_t), s)) != hipSuccess) return e;
  hipLaunchKernelGGL(k_shared_g2_guards, dim3(1), dim3(64), 0, s, gamma, delta, good);
  // The statement sums S_i = sum_j stmt[i][j] uvw_j as ONE batched scalar multiplication, one lane per term, on a kernel that runs four waves per SIMD —
  // inside the verification kernel (one wave per SIMD, 512 registers) the same multiply-adds issue at less than half the rate.
  uint32_t *tmp = nullptr, *S = nullptr;
  if (n_stmt >= 1 && n_stmt <= 12) {
    if ((e = hipMallocAsync((void**)&tmp, (size_t)n_stmt * n * ABI_G1_WORDS * 4, s)) != hipSuccess) { (void)hipFreeAsync(good, s); return e; }
    if ((e = hipMallocAsync((void**)&S, n * ABI_G1_WORDS * 4, s)) != hipSuccess) { (void)hipFreeAsync(good, s); (void)hipFreeAsync(tmp, s); return e; }
    MulSegs segs; segs.n = n_stmt;
    for (int j = 0; j < n_stmt; ++j) segs.s[j] = MulSeg{uvw_stmt + (size_t)j * ABI_G1_WORDS, stmt + (size_t)j * 8, tmp + (size_t)j * n * ABI_G1_WORDS, (uint32_t)n, 0u, (uint32_t)(n_stmt * 8)};
    if ((e = launch_group_mul_segs(G_G1, segs, 8, s)) != hipSuccess) { (void)hipFreeAsync(good, s); (void)hipFreeAsync(tmp, s); (void)hipFreeAsync(S, s); return e; }
    hipLaunchKernelGGL(k_stmt_sums, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, (const uint32_t*)tmp, n_stmt, S, n);
  }
  hipLaunchKernelGGL(k_groth16_verify<true>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, A, B, C, uvw_stmt, stmt, n_stmt, gamma, delta, alpha_beta, ok, n, err, 0, (const uint32_t*)good, (const uint32_t*)S);
  hipLaunchKernelGGL(k_groth16_verify<false>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, A, B, C, uvw_stmt, stmt, n_stmt, gamma, delta, alpha_beta, ok, n, err, 1, (const uint32_t*)nullptr, (const uint32_t*)S);
  if (!S) {                                  // statements the batched multiplication does not take (n_stmt = 0 or > 12): the sums of the marked elements, one lane each
    if ((e = hipMallocAsync((void**)&S, n * ABI_G1_WORDS * 4, s)) != hipSuccess) { (void)hipFreeAsync(good, s); return e; }
    hipLaunchKernelGGL(k_stmt_sums_marked, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, uvw_stmt, stmt, n_stmt, (const uint32_t*)ok, S, n);
  }
  if ((e = hipGetLastError()) != hipSuccess || (e = finish_exact(groth16_pairs(A, B, C, S, gamma, delta), 3, nullptr, alpha_beta, ok, n, err, s)) != hipSuccess) {
    (void)hipFreeAsync(good, s); if (tmp) (void)hipFreeAsync(tmp, s); (void)hipFreeAsync(S, s); return e; }
  if ((e = hipFreeAsync(good, s)) != hipSuccess) return e;
  if (tmp && (e = hipFreeAsync(tmp, s)) != hipSuccess) return e;
  if (S && (e = hipFreeAsync(S, s)) != hipSuccess) return e;
  return hipGetLastError();
}

// Equality of pairing products as the reference's callers test it (lhs == rhs on GTPoints: signature.rs:34-39,
// pinocchio/verifier.rs:43-84): e(P1,Q1) == e(P2,Q2) e(P3,Q3)  <=>  tate-product(P1,Q1; -P2,Q2; -P3,Q3) == 1, since
// e(-P,Q) = e(P,Q)^-1 exactly.  One element per lane; an argument at infinity is the reference's panic (rational_function.rs:36,59).
template <int K, bool SHORT>
__global__ void __launch_bounds__(64) k_pairing_product_check(PairArgs a, uint32_t* __restrict__ ok, size_t n, unsigned long long* err, int only_redo) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  if (only_redo && ok[i] != OK_REDO) return;
  Fq xp[K], yp[K]; Fq2 xq[K], yq[K];
  bool inf = false;
  for (int k = 0; k < K; ++k) {
    Aff<FqOps> p = PtIO<FqOps>::ld(a.g1[k] + i * a.s1[k]);
    Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(a.g2[k] + i * a.s2[k]);
    inf = inf || p.inf || q.inf;
    xp[k] = p.x; yp[k] = a.neg[k] ? fp_neg(p.y) : p.y; xq[k] = q.x; yq[k] = q.y;
  }
  if (inf) { atomicMin(err, (unsigned long long)i); ok[i] = 0; return; }
  bool in_g1;
  Fq12 e;
  if constexpr (SHORT) {                   // (prod tate)^(1/(2x^2-1)) is one exactly when the Tate product is: no correction needed for "== 1"
    if (!pairing_args_fit_short_loop<K>(xp, yp, xq, yq)) { ok[i] = OK_REDO; return; }
    e = final_exponentiation(miller_g1_g2_multi_short<K>(xp, yp, xq, yq, in_g1));
  } else {
    for (int k = 0; k < K; ++k) if (!g1_on_curve(xp[k], yp[k]) || !g2_on_curve(xq[k], yq[k])) { ok[i] = OK_EXACT; return; }
    e = final_exponentiation(miller_g1_g2_multi<K>(xp, yp, xq, yq, in_g1));
  }
  if (!in_g1) { ok[i] = OK_EXACT; return; }       // as in k_groth16_verify
  uint32_t got[144]; st_fq12(got, e);
  uint32_t diff = got[132] ^ 1u;                       // canonical one: w0.v0.u0 = 1 (the last Fq of the {w1,w0} layout), all else 0
  for (int k = 0; k < 144; ++k) if (k != 132) diff |= got[k];
  ok[i] = diff == 0;
}
// small batches go to the lane-distributed kernels (zkt_dpairing.hip): ~10 ms per element instead of ~100 ms, lower peak throughput
size_t dproduct_limit() { static const size_t v = [] { const char* e = getenv("ZKT_DPRODUCT_MAX"); return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)24576; }(); return v; }
hipError_t launch_pairing_product_check_counts(const PairArgs& a, int K, const uint8_t* kcount, uint32_t* ok, size_t n, unsigned long long* err, hipStream_t s) {
  if (n == 0) return hipSuccess;
  if (K < 1 || K > 4 || !kcount || n * (size_t)K > dproduct_limit()) return hipErrorInvalidValue;
  uint32_t* flags = nullptr; hipStream_t side; hipError_t e;
  if ((e = hipMallocAsync((void**)&flags, n * sizeof(uint32_t), s)) != hipSuccess) return e;
  const bool ate = small_ate();
  if ((e = guard_fork(s, &side)) != hipSuccess || (e = (ate ? launch_ate_guards(a, K, flags, n, side) : launch_short_loop_guards(a, K, flags, n, side))) != hipSuccess ||          // the unused slots repeat pair 0: same verdict
      (e = (ate ? launch_dproduct_ate(a, K, nullptr, ok, n, err, s, kcount) : launch_dproduct(a, K, nullptr, ok, n, err, true, s, kcount))) != hipSuccess ||
      (e = guard_join(s, side)) != hipSuccess) { (void)hipFreeAsync(flags, s); return e; }
  hipLaunchKernelGGL(k_product_resolve, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const uint32_t*)flags, ok, n);
  if ((e = hipFreeAsync(flags, s)) != hipSuccess) return e;
  return hipGetLastError();
}
hipError_t launch_pairing_product_check(const PairArgs& a, int K, uint32_t* ok, size_t n, unsigned long long* err, hipStream_t s, uint32_t p_trusted) {
  if (n == 0) return hipSuccess;
  dim3 g((unsigned)((n + 63) / 64)), t(64);
  const bool small = n * (size_t)K <= dproduct_limit();
  if (K < 1 || K > 4) return hipErrorInvalidValue;
  if (small) {                             // 127-step loops on the lane-distributed kernels, their preconditions checked beside them
    uint32_t* flags = nullptr; hipStream_t side; hipError_t e;
    if ((e = hipMallocAsync((void**)&flags, n * sizeof(uint32_t), s)) != hipSuccess) return e;
    const bool sate = small_ate();
    if ((e = guard_fork(s, &side)) != hipSuccess || (e = (sate ? launch_ate_guards(a, K, flags, n, side, p_trusted) : launch_short_loop_guards(a, K, flags, n, side))) != hipSuccess ||
        (e = (sate ? launch_dproduct_ate(a, K, nullptr, ok, n, err, s) : launch_dproduct(a, K, nullptr, ok, n, err, true, s))) != hipSuccess ||
        (e = guard_join(s, side)) != hipSuccess) { (void)hipFreeAsync(flags, s); return e; }
    hipLaunchKernelGGL(k_product_resolve, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const uint32_t*)flags, ok, n);
    if ((e = hipFreeAsync(flags, s)) != hipSuccess) return e;
  }
  // large batches: the 63-step loop decides; what it marks (an argument outside its group) goes to the 255-step kernel as before.  ZKT_PRODUCT_LOOP=127 keeps
  // the round-2 kernel (the twisted-ate loop over x^2) for A/B measurements.
  static const bool ate = [] { const char* e = getenv("ZKT_PRODUCT_LOOP"); return !(e && atoi(e) == 127); }();
  uint32_t* p_good = nullptr;
  bool shared_untrusted = false;            // a G1 point shared by the batch that is not one of the library's own constants: tested once per launch (1.7 ms on one lane)
  for (int k = 0; k < K; ++k) shared_untrusted = shared_untrusted || (a.s1[k] == 0 && !((p_trusted >> k) & 1));
  if (!small && ate && shared_untrusted) {
    hipError_t e;
    if ((e = hipMallocAsync((void**)&p_good, sizeof(uint32_t), s)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_shared_g1_guards, dim3(1), dim3(64), 0, s, a, K, p_good);
  }
  uint32_t* fits = nullptr;
  if (!small && ate) {            // membership of the per-element G1 arguments at high occupancy (shared slots: p_good / p_trusted; their rows of `fits` are not read)
    hipError_t e;
    if ((e = hipMallocAsync((void**)&fits, (size_t)K * n * sizeof(uint32_t), s)) != hipSuccess) return e;
    G1Fits gf{};
    for (int k = 0; k < K; ++k) { gf.pts[k] = a.g1[k]; gf.stride[k] = a.s1[k]; }
    if ((e = launch_g1_fits(gf, K, fits, n, s)) != hipSuccess) { (void)hipFreeAsync(fits, s); return e; }
  }
#define ZKT_PRODUCT_CHECK(KK) if (!small) { if (ate) hipLaunchKernelGGL((k_pairing_product_check_ate<KK>), g, t, 0, s, a, ok, n, err, (const uint32_t*)p_good, p_trusted, (const uint32_t*)fits); \
                                            else hipLaunchKernelGGL((k_pairing_product_check<KK, true>), g, t, 0, s, a, ok, n, err, 0); } \
                              hipLaunchKernelGGL((k_pairing_product_check<KK, false>), g, t, 0, s, a, ok, n, err, 1)
  switch (K) {
    case 1: ZKT_PRODUCT_CHECK(1); break;
    case 2: ZKT_PRODUCT_CHECK(2); break;
    case 3: ZKT_PRODUCT_CHECK(3); break;
    case 4: ZKT_PRODUCT_CHECK(4); break;
    default: return hipErrorInvalidValue;
  }
#undef ZKT_PRODUCT_CHECK
  if (p_good) { hipError_t e = hipFreeAsync(p_good, s); if (e != hipSuccess) return e; }
  if (fits) { hipError_t e = hipFreeAsync(fits, s); if (e != hipSuccess) return e; }
  { hipError_t e = hipGetLastError(); if (e != hipSuccess) return e; }
  return finish_exact(a, K, nullptr, nullptr, ok, n, err, s);
}

}  // namespace zkt
