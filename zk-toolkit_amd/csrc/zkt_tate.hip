// Batched Tate pairing kernel: one pairing per lane (rows a10–a13 of SURVEY §8).
//   Pairing::tate   bls12_381/pairing.rs:86-100
// The algorithm and why it is bit-identical to the reference are in pairing.h.
// This kernel has its own translation unit: the register budget of a kernel is propagated to the (non-inlined) field and tower
// functions it calls only when no other kernel shares them.  v_mad_u64_u32 issues at half rate with one wave per SIMD
// (profiles/r01_mad_issue_latency.txt), so two waves per SIMD (256 registers, no AGPRs) is the goal — but measured on MI355X with
// -DZKT_TATE_ATTR='__attribute__((amdgpu_waves_per_eu(2,2)))' the extra spills (9.5 instead of 7.9 KB of scratch per lane) cost more
// than the issue rate gains: 426 k instead of 667 k pairings/s.  The default therefore stays at one wave per SIMD until the Fq12
// working set is cut (or spread over several lanes).
#include <cstdlib>
#include "abi.h"
#include "zkt_internal.h"

namespace zkt {

// word 11 of an output element is the top word of a canonical Fq (<= 0x1a0111ea): these values mark "not computed yet"
static constexpr int TATE_MARK_WORD = 11;
static constexpr uint32_t TATE_MARK_LONG = 0xffffffffu;       // left to k_tate_long_marked: the preconditions of the 127-step loop do not hold
static constexpr uint32_t TATE_MARK_EXACT = 0xfffffffeu;      // left to k_tate_exact_marked: r P != infinity
#ifndef ZKT_TATE_ATTR
#define ZKT_TATE_ATTR
#endif
// First pass, the only one honest inputs see: P on E and in G1, Q in G2 -> 127-step loop, final exponentiation, tate = eta^(2x^2-1) (pairing.h).
__global__ void __launch_bounds__(64) ZKT_TATE_ATTR k_tate(const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g2,
                                             uint32_t* __restrict__ out, size_t n, unsigned long long* err) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  Aff<FqOps> p = PtIO<FqOps>::ld(g1 + i * ABI_G1_WORDS);
  Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(g2 + i * ABI_G2_WORDS);
  if (p.inf || q.inf) {   // RationalFunction::new_* / eval_with_* panic on infinity (rational_function.rs:36,59)
    atomicMin(err, (unsigned long long)i);
    return;
  }
  // (Round 4 measured the curve equations and the G2 membership test of tate_short moved AHEAD of this kernel, onto lane pairs at two waves per SIMD: k_tate 48.0 ms
  //  against 47.8 with them inside — the 63-doubling chain over Fq2 is not where this kernel's time goes.  Taken out again.)
  Fq12 r;
  const int route = tate_short(p.x, p.y, q.x, q.y, r);
  if (route == TATE_ROUTE_SHORT) st_fq12(out + i * 144, r);
  else out[i * 144 + TATE_MARK_WORD] = route == TATE_ROUTE_LONG ? TATE_MARK_LONG : TATE_MARK_EXACT;
}
// Second pass, for the elements the first one marked: the 255-step loop over r - 1, which needs nothing of Q and only r P = infinity of P
// (checked for free where it ends).  Honest inputs never get here: every lane of every wave leaves after one load.
__global__ void __launch_bounds__(64) ZKT_TATE_ATTR k_tate_long_marked(const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g2,
                                                                        uint32_t* __restrict__ out, size_t n, unsigned long long* err) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n || out[i * 144 + TATE_MARK_WORD] != TATE_MARK_LONG) return;
  Aff<FqOps> p = PtIO<FqOps>::ld(g1 + i * ABI_G1_WORDS);
  Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(g2 + i * ABI_G2_WORDS);
  bool in_g1;
  Fq12 f = miller_g1_g2(p.x, p.y, q.x, q.y, in_g1);
  if (!in_g1) { out[i * 144 + TATE_MARK_WORD] = TATE_MARK_EXACT; return; }  // r P != infinity: left to k_tate_exact_marked
  st_fq12(out + i * 144, final_exponentiation(f));
}
// After the small-batch kernels (zkt_dpairing.hip: the 127-step loop for every element, on trust): elements the guard kernel (zkt_pairing.hip,
// run beside them) found unfit are re-marked — a point off its curve for the reference's chain, Q on E' outside G2 for the 255-step loop.
__global__ void __launch_bounds__(64) k_tate_resolve(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g2,
                                                     uint32_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n || flags[i]) return;
  Aff<FqOps> p = PtIO<FqOps>::ld(g1 + i * ABI_G1_WORDS);
  Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(g2 + i * ABI_G2_WORDS);
  if (p.inf || q.inf) return;
  out[i * 144 + TATE_MARK_WORD] = (g1_on_curve(p.x, p.y) && g2_on_curve(q.x, q.y)) ? TATE_MARK_LONG : TATE_MARK_EXACT;
}
// Third pass, for the elements marked EXACT: P outside the order-r subgroup (or a point off its curve).  There the reference's result depends on
// the order of P — it panics when a multiple of P met by its binary chain is infinity (rational_function.rs:36) — so these lanes follow
// the reference's chain step by step (miller_g1_g2_exact) and report its panics as the batch's error index.
__global__ void __launch_bounds__(64) ZKT_TATE_ATTR k_tate_exact_marked(const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g2,
                                                          uint32_t* __restrict__ out, size_t n, unsigned long long* err) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n || out[i * 144 + TATE_MARK_WORD] != TATE_MARK_EXACT) return;
  Aff<FqOps> p = PtIO<FqOps>::ld(g1 + i * ABI_G1_WORDS);
  Aff<Fq2Ops> q = PtIO<Fq2Ops>::ld(g2 + i * ABI_G2_WORDS);
  bool bad;
  Fq12 f = miller_g1_g2_exact(p.x, p.y, q.x, q.y, bad);
  if (bad) { atomicMin(err, (unsigned long long)i); return; }
  if (fq12_is_zero(f)) { for (int k = 0; k < 144; ++k) out[i * 144 + k] = 0; return; }      // 0^e = 0 (fq12.rs:42-57)
  st_fq12(out + i * 144, final_exponentiation(f));
}

// How many elements the passes so far left marked (counts[0]: for the 255-step loop, counts[1]: for the reference's chain).  The two redo kernels have the
// largest scratch frames of the library (9.5 and 17 KB per lane: the runtime would keep 6 GiB of scratch on the queue for kernels that, on honest
// input, have nothing to do), so they are launched only when an 8-byte read-back says there is work for them.
__global__ void __launch_bounds__(256) k_count_marks(const uint32_t* __restrict__ out, size_t n, uint32_t* __restrict__ counts) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t w = out[i * 144 + TATE_MARK_WORD];
  if (w == TATE_MARK_LONG) atomicAdd(&counts[0], 1u); else if (w == TATE_MARK_EXACT) atomicAdd(&counts[1], 1u);
}
static hipError_t count_marks(const uint32_t* out, size_t n, hipStream_t s, uint32_t host[2]) {
  uint32_t* d = nullptr; hipError_t e;
  if ((e = hipMallocAsync((void**)&d, 8, s)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(d, 0, 8, s)) != hipSuccess) { (void)hipFreeAsync(d, s); return e; }
  hipLaunchKernelGGL(k_count_marks, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, n, d);
  if ((e = hipMemcpyAsync(host, d, 8, hipMemcpyDeviceToHost, s)) != hipSuccess) { (void)hipFreeAsync(d, s); return e; }
  if ((e = hipFreeAsync(d, s)) != hipSuccess) return e;
  return hipStreamSynchronize(s);
}
hipError_t launch_tate(const uint32_t* g1, const uint32_t* g2, uint32_t* out, size_t n, unsigned long long* err, hipStream_t s) {
  if (n == 0) return hipSuccess;
  // small batches: one pairing per 12 lanes (zkt_dpairing.hip) — ~10x lower latency per pairing, lower peak throughput.  ZKT_DTATE_MAX overrides the switch-over.
  static const size_t dmax = [] { const char* e = getenv("ZKT_DTATE_MAX"); return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)24576; }();      // measured after the 127-step loop: 24,576 pairings 37.2 ms here, 42.4 ms on k_tate; 32,768: 47.6 against 43.5
  if (n <= dmax) {
    PairArgs a{}; a.g1[0] = g1; a.s1[0] = ABI_G1_WORDS; a.g2[0] = g2; a.s2[0] = ABI_G2_WORDS;
    uint32_t* flags = nullptr; hipStream_t side; hipError_t e;
    if ((e = hipMallocAsync((void**)&flags, n * sizeof(uint32_t), s)) != hipSuccess) return e;
    if ((e = guard_fork(s, &side)) != hipSuccess || (e = launch_short_loop_guards(a, 1, flags, n, side)) != hipSuccess ||
        (e = launch_dtate(g1, g2, out, n, err, TATE_MARK_WORD, TATE_MARK_EXACT, true, s)) != hipSuccess || (e = guard_join(s, side)) != hipSuccess) { (void)hipFreeAsync(flags, s); return e; }
    hipLaunchKernelGGL(k_tate_resolve, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, (const uint32_t*)flags, g1, g2, out, n);
    if ((e = hipFreeAsync(flags, s)) != hipSuccess) return e;
  } else {
    hipLaunchKernelGGL(k_tate, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, g1, g2, out, n, err);
  }
  hipError_t e; uint32_t marks[2] = {0, 0};
  if ((e = hipGetLastError()) != hipSuccess || (e = count_marks(out, n, s, marks)) != hipSuccess) return e;
  if (marks[0]) {                                           // Q on E' outside G2 somewhere: the 255-step loop for those elements (it may hand some on)
    hipLaunchKernelGGL(k_tate_long_marked, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, g1, g2, out, n, err);
    if ((e = hipGetLastError()) != hipSuccess || (e = count_marks(out, n, s, marks)) != hipSuccess) return e;
  }
  if (marks[1]) hipLaunchKernelGGL(k_tate_exact_marked, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, g1, g2, out, n, err);
  return hipGetLastError();
}

}  // namespace zkt
