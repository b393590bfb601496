// C ABI (include/zkt.h) over the HIP kernels.  Host-pointer entry points stage through
// a grow-only device arena; `_dev` entry points launch on the caller's stream.  There is
// no CPU compute path in this file: without a device every entry point fails.
#include <hip/hip_runtime.h>
#include <mutex>
#include <vector>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include "../../include/zkt.h"
#include "zkt_internal.h"
#include "zkt_constants.h"
static_assert(ZKT_G1_PARTIAL_WORDS == 3 * zkt::FqC::N && ZKT_G2_PARTIAL_WORDS == 6 * zkt::FqC::N && ZKT_SECP_PARTIAL_WORDS == 3 * zkt::SpC::N,
              "include/zkt.h partial sizes follow the internal limb layout");

using namespace zkt;

namespace {

struct Ctx {
  bool ready = false;
  int device = -1;
  hipStream_t stream = nullptr;
  unsigned long long* d_err = nullptr;
  uint8_t* arena = nullptr; size_t arena_bytes = 0;
  uint32_t* d_small = nullptr;          // 4 KB of device scratch for one-point results (caller holds mu)
  std::mutex mu;
};
Ctx g;
thread_local size_t t_err_index = 0;
thread_local float t_kernel_ms = 0.f;
thread_local const char* t_kernel_name = "";

#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "[zkt] HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); return ZKT_ERR_DEVICE; } } while (0)

// every entry point starts here: the library owns ONE device (zkt_init), and a thread's current HIP device is per-thread state,
// so it is set on entry — handles, streams and workspaces are then always created and used on that device
int ensure_ready() {
  if (!g.ready) return ZKT_ERR_DEVICE;
  return hipSetDevice(g.device) == hipSuccess ? ZKT_OK : ZKT_ERR_DEVICE;
}

// grow-only staging arena (caller holds g.mu)
int arena_reserve(size_t bytes) {
  if (bytes <= g.arena_bytes) return ZKT_OK;
  if (g.arena) { HIPCHK(hipFree(g.arena)); g.arena = nullptr; g.arena_bytes = 0; }
  size_t want = bytes + (bytes >> 2) + (1 << 20);
  HIPCHK(hipMalloc((void**)&g.arena, want));
  g.arena_bytes = want;
  return ZKT_OK;
}
struct Carver {
  uint8_t* p; size_t off = 0;
  explicit Carver(uint8_t* base) : p(base) {}
  template <class T> T* take(size_t bytes) { off = (off + 255) & ~size_t(255); T* r = (T*)(p + off); off += bytes; return r; }
};
size_t padded(size_t b) { return (b + 255) & ~size_t(255); }

int reset_err(hipStream_t s) {
  unsigned long long v = NO_ERR;
  HIPCHK(hipMemcpyAsync(g.d_err, &v, sizeof(v), hipMemcpyHostToDevice, s));
  return ZKT_OK;
}
int fetch_err(hipStream_t s, int code_if_set) {
  unsigned long long v = NO_ERR;
  HIPCHK(hipMemcpyAsync(&v, g.d_err, sizeof(v), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (v != NO_ERR) { t_err_index = (size_t)v; return code_if_set; }
  return ZKT_OK;
}

// generic host-staged op: in_a, in_b (optional) -> out, sizes in bytes
template <class Launch>
int staged_raw(const void* a, size_t a_total, const void* b, size_t b_total, void* out, size_t out_total, int err_code, Launch launch) {
  if (ensure_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (!a || !out || (b_total && !b)) return ZKT_ERR_SHAPE;
  std::lock_guard<std::mutex> lk(g.mu);
  HIPCHK(hipSetDevice(g.device));
  int rc = arena_reserve(padded(a_total) + padded(b_total) + padded(out_total) + 1024);
  if (rc) return rc;
  Carver cv(g.arena);
  uint32_t* da = cv.take<uint32_t>(a_total);
  uint32_t* db = b_total ? cv.take<uint32_t>(b_total) : nullptr;
  uint32_t* dout = cv.take<uint32_t>(out_total);
  HIPCHK(hipMemcpyAsync(da, a, a_total, hipMemcpyHostToDevice, g.stream));
  if (db) HIPCHK(hipMemcpyAsync(db, b, b_total, hipMemcpyHostToDevice, g.stream));
  if ((rc = reset_err(g.stream))) return rc;
  HIPCHK(launch(da, db, dout, g.stream));
  HIPCHK(hipMemcpyAsync(out, dout, out_total, hipMemcpyDeviceToHost, g.stream));
  return fetch_err(g.stream, err_code);
}
// elementwise form: fixed bytes per element
template <class Launch>
int staged(const void* a, size_t a_bytes, const void* b, size_t b_bytes, void* out, size_t out_bytes, size_t n, int err_code, Launch launch) {
  if (ensure_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (n == 0) return ZKT_OK;
  return staged_raw(a, a_bytes * n, b, b_bytes * n, out, out_bytes * n, err_code, launch);
}

size_t field_bytes(int field) { return field == F_FQ ? 48 : 32; }
int fp_batch(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  const size_t w = field_bytes(field);
  const bool binary = op == OP_ADD || op == OP_SUB || op == OP_MUL;
  return staged(a, w, binary ? b : nullptr, binary ? w : 0, out, w, n, ZKT_ERR_INV_ZERO,
                [&](uint32_t* da, uint32_t* db, uint32_t* dout, hipStream_t s) { return launch_fp_op(field, op, da, db, dout, n, g.d_err, s); });
}
// a3: pow with per-element or shared exponents of exp_limbs u64 limbs; pow_seq / repeat from one base element
int fp_pow_batch(int field, const uint64_t* a, const uint64_t* exps, size_t exp_limbs, int exp_shared, uint64_t* out, size_t n) {
  if (exp_limbs == 0 || exp_limbs > 64) return ZKT_ERR_SHAPE;
  if (n == 0) return ensure_ready();
  const size_t w = field_bytes(field);
  return staged_raw(a, w * n, exps, exp_limbs * 8 * (exp_shared ? 1 : n), out, w * n, ZKT_ERR_SHAPE,
                    [&](uint32_t* da, uint32_t* db, uint32_t* dout, hipStream_t s) { return launch_fp_pow(field, da, db, (int)exp_limbs * 2, exp_shared != 0, dout, n, s); });
}
int fp_pow_seq(int field, const uint64_t* base, size_t n, uint64_t* out, bool repeat) {
  if (n == 0) return ensure_ready();
  const size_t w = field_bytes(field);
  return staged_raw(base, w, nullptr, 0, out, w * n, ZKT_ERR_SHAPE,
                    [&](uint32_t* da, uint32_t*, uint32_t* dout, hipStream_t s) { return launch_fp_pow_seq(field, da, dout, n, repeat, s); });
}
// a18's vector forms: PrimeFieldElems::sum (prime_field_elems.rs:35-41, panics on an empty vector) and PrimeFieldElems * PrimeFieldElem (:152-175)
int fp_sum(int field, const uint64_t* a, size_t n, uint64_t* out) {
  if (ensure_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (n == 0 || !a || !out) return ZKT_ERR_SHAPE;                    // assert!(self.0.len() > 0)
  const size_t w = field_bytes(field);
  std::lock_guard<std::mutex> lk(g.mu);
  HIPCHK(hipSetDevice(g.device));
  int rc = arena_reserve(padded(w * n) + padded(w * (1 + fp_sum_scratch_elems())) + 1024);
  if (rc) return rc;
  Carver cv(g.arena);
  uint32_t* da = cv.take<uint32_t>(w * n); uint32_t* dout = cv.take<uint32_t>(w * (1 + fp_sum_scratch_elems()));
  HIPCHK(hipMemcpyAsync(da, a, w * n, hipMemcpyHostToDevice, g.stream));
  HIPCHK(launch_fp_sum(field, da, n, dout, dout + w / 4, g.stream));
  HIPCHK(hipMemcpyAsync(out, dout, w, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  return ZKT_OK;
}
int fp_scale(int field, const uint64_t* a, const uint64_t* k, uint64_t* out, size_t n) {
  if (n == 0) return ZKT_ERR_SHAPE;                                  // assert!(self.len() > 0)
  const size_t w = field_bytes(field);
  return staged_raw(a, w * n, k, w, out, w * n, ZKT_ERR_SHAPE,
                    [&](uint32_t* da, uint32_t* db, uint32_t* dout, hipStream_t s) { return launch_fp_scale(field, da, db, dout, n, s); });
}
int tower_batch(int deg, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  const size_t w = (size_t)deg * 48;
  const bool binary = op == T_ADD || op == T_SUB || op == T_MUL;
  return staged(a, w, binary ? b : nullptr, binary ? w : 0, out, w, n, ZKT_ERR_INV_ZERO,
                [&](uint32_t* da, uint32_t* db, uint32_t* dout, hipStream_t s) { return launch_tower_op(deg, op, da, db, dout, n, g.d_err, s); });
}
size_t pt_bytes(int grp) { return grp == G_G1 ? sizeof(zkt_g1_affine) : grp == G_G2 ? sizeof(zkt_g2_affine) : sizeof(zkt_secp_affine); }

}  // namespace

static constexpr int MSM_SLOTS = 8;
struct MsmSlot {                      // one in-flight MSM: workspace, result buffers, stage events
  hipEvent_t e_in = nullptr, e_sorted = nullptr, e_acc0 = nullptr, e_acc1 = nullptr, e_done = nullptr;
  void* workspace = nullptr;
  uint32_t* d_result_jac = nullptr;   // 3 coordinates (Jacobian partial)
  uint32_t* d_out_abi = nullptr;      // one ABI point
  uint8_t* h_out = nullptr;           // pinned, one ABI point
  bool busy = false;
  // graph replay of a small MSM's pipeline (msm_submit_locked): executable graphs captured on this slot, keyed by the scalar vector's address
  static constexpr int NGRAPH = 4;
  hipGraphExec_t gexec[NGRAPH] = {}; const void* gkey[NGRAPH] = {}; unsigned gnext = 0; bool timed = false;
};
struct zkt_bases_impl {               // one resident base set of any group; zkt_g1_bases / zkt_g2_bases / zkt_secp_bases are this
  size_t n = 0;
  int grp = G_G1;
  MsmPlan plan{};
  uint32_t* table = nullptr;     // nwin*n affine points, 2 internal coordinates each
  uint8_t* inf = nullptr;        // nwin*n flags
  // software pipeline: the three stages of consecutive MSMs run on three streams (sort | accumulate | reduce),
  // chained by events, so the atomic-bound sort and the latency-bound reduce of neighbours hide under the
  // VALU-bound accumulation of the current one.
  static constexpr int GROUP_TAILS = 4;   // reduce streams of a group of sets that share their streams (zkt_internal_bases_share_streams)
  static constexpr int NTAIL = 8;   // reduce chains of alternate MSMs run side by side: each is latency-bound, not throughput-bound (large MSMs use two of them)
  hipStream_t s_sort = nullptr, s_acc = nullptr, s_tail[NTAIL] = {};
  // a group of base sets that always work on the same job (the four sets of a Groth16 key) shares ONE set of streams: every stream beyond the
  // hardware queues (8) is folded onto a queue that already carries another stream, and a sort queued behind someone else's reduce chain waits for it
  // (measured: the A sum of a proof started 16 ms late behind the C1 reduce, profiles/r03_groth16_timeline.txt)
  bool own_streams = true, acc_owned = false, grouped = false; int tail_base = 0, tail_span = 0;
  MsmSlot slot[MSM_SLOTS];
  std::mutex mu;                 // slot state: calls on one handle are serialised (submit/collect of different slots may come from different threads)
};
struct zkt_g1_bases : zkt_bases_impl {};
struct zkt_g2_bases : zkt_bases_impl {};
struct zkt_secp_bases : zkt_bases_impl {};
static size_t grp_pt_bytes(int grp) { return grp == G_G1 ? sizeof(zkt_g1_affine) : grp == G_G2 ? sizeof(zkt_g2_affine) : sizeof(zkt_secp_affine); }
static size_t grp_coord_bytes(int grp) { return 4 * (grp == G_G1 ? zkt::FqC::N : grp == G_G2 ? 2 * zkt::FqC::N : zkt::SpC::N); }   // internal (Montgomery) coordinate
static int streams_ready(zkt_bases_impl* h) {
  if (h->grouped || !h->own_streams || (h->s_acc && h->s_sort && h->s_tail[zkt_bases_impl::NTAIL - 1])) return ZKT_OK;
  int lo = 0, hi = 0;
  HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));        // hi = numerically smallest = highest priority
  if (!h->s_sort) HIPCHK(hipStreamCreateWithPriority(&h->s_sort, hipStreamNonBlocking, hi));
  if (!h->s_acc) HIPCHK(hipStreamCreateWithPriority(&h->s_acc, hipStreamNonBlocking, lo));
  for (int k = 0; k < zkt_bases_impl::NTAIL; ++k) if (!h->s_tail[k]) HIPCHK(hipStreamCreateWithPriority(&h->s_tail[k], hipStreamNonBlocking, hi));
  return ZKT_OK;
}
// ZKT_DEBUG_POISON=1: every MSM workspace is filled with a garbage pattern when it is allocated, so a kernel that reads a word nobody wrote gets 0xA5A5A5A5
// instead of the zeros a fresh allocation happens to hold (tools/diag/msm_repeat.py and the MSM tests are run this way).
static bool debug_poison() { static const bool on = [] { const char* e = getenv("ZKT_DEBUG_POISON"); return e && *e == '1'; }(); return on; }
// The pipeline of an MSM below 2^19 terms is replayed as one graph launch per submit (msm_submit_locked); ZKT_MSM_GRAPH=0 issues its launches one by one instead.
// Round 2 took this out because the 65,536-bit range-proof test aborted; round 3 found why (tools/diag/rp_graph.py): the captured graph held runtime-owned nodes — the
// hipMemsetAsync of the counters and the copy of the result — and replaying such a graph after ANY later hipFree in the process (a torch cache flush, a second context
// freeing its build scratch) ended in a memory access fault.  With kernel nodes only (k_zero_words clears the counters, the copy follows the graph on the stream) the
// replay survives all of that: every variant of the diagnostic, and the whole GPU suite, run in this mode.
static bool msm_graphs() { static const bool on = [] { const char* e = getenv("ZKT_MSM_GRAPH"); return !(e && *e == '0'); }(); return on; }
static int slot_ready(zkt_bases_impl* h, int k) {   // lazily create the slot's workspace
  int rc = streams_ready(h); if (rc) return rc;
  MsmSlot& S = h->slot[k];
  if (S.workspace) return ZKT_OK;
  HIPCHK(hipEventCreateWithFlags(&S.e_in, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&S.e_sorted, hipEventDisableTiming));
  HIPCHK(hipEventCreate(&S.e_acc0)); HIPCHK(hipEventCreate(&S.e_acc1)); HIPCHK(hipEventCreateWithFlags(&S.e_done, hipEventDisableTiming));
  HIPCHK(hipMalloc(&S.workspace, h->plan.ws_bytes));
  if (debug_poison()) { HIPCHK(hipMemset(S.workspace, 0xA5, h->plan.ws_bytes)); HIPCHK(hipDeviceSynchronize()); }
  HIPCHK(hipMalloc((void**)&S.d_result_jac, 3 * grp_coord_bytes(h->grp)));
  HIPCHK(hipMalloc((void**)&S.d_out_abi, grp_pt_bytes(h->grp)));
  HIPCHK(hipHostMalloc((void**)&S.h_out, grp_pt_bytes(h->grp), hipHostMallocDefault));
  return ZKT_OK;
}

int zkt_internal_ready() { return ensure_ready(); }
hipStream_t zkt_internal_stream() { return g.stream; }
int zkt_internal_device() { return g.device; }
void zkt_internal_set_error_index(size_t i) { t_err_index = i; }

extern "C" {

int zkt_version(void) { return 1; }
const char* zkt_strerror(int s) {
  switch (s) {
    case ZKT_OK: return "ok";
    case ZKT_ERR_INV_ZERO: return "Cannot find inverse of zero";
    case ZKT_ERR_INFINITY: return "pairing argument is the point at infinity";
    case ZKT_ERR_SHAPE: return "bad size, null pointer or non-canonical input";
    case ZKT_ERR_DEVICE: return "no HIP device / HIP error / zkt_init not called";
  }
  return "unknown";
}
size_t zkt_last_error_index(void) { return t_err_index; }
float zkt_last_kernel_ms(void) { return t_kernel_ms; }
const char* zkt_last_kernel_name(void) { return t_kernel_name; }

int zkt_init(int device) {
  std::lock_guard<std::mutex> lk(g.mu);
  if (g.ready) return ZKT_OK;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { fprintf(stderr, "[zkt] no HIP device: this library has no CPU path\n"); return ZKT_ERR_DEVICE; }
  if (device < 0) { if (hipGetDevice(&device) != hipSuccess) return ZKT_ERR_DEVICE; }
  if (device >= count) return ZKT_ERR_DEVICE;
  HIPCHK(hipSetDevice(device));
  g.device = device;
  HIPCHK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
  HIPCHK(hipMalloc((void**)&g.d_err, 64));
  HIPCHK(hipMalloc((void**)&g.d_small, 4096));
  g.ready = true;
  return ZKT_OK;
}
extern "C" void zkt_internal_clear_caches();       // zkt_protocols.hip
static void tate_events_release();
void zkt_shutdown(void) {
  if (g.ready) (void)hipSetDevice(g.device);
  zkt_comm_finalize();                                // the communicator and its buffers live on this device
  zkt_internal_clear_caches();                        // before g.mu is taken: releasing a cached context frees base sets, which lock it themselves
  std::lock_guard<std::mutex> lk(g.mu);
  if (!g.ready) return;
  (void)hipSetDevice(g.device);
  group_release_device_state();                       // generator comb tables
  pairing_release_device_state();                     // guard side stream + events
  tate_events_release();
  if (g.arena) (void)hipFree(g.arena);
  if (g.d_err) (void)hipFree(g.d_err);
  if (g.d_small) (void)hipFree(g.d_small);
  if (g.stream) (void)hipStreamDestroy(g.stream);
  g.ready = false; g.device = -1; g.stream = nullptr; g.d_err = nullptr; g.d_small = nullptr; g.arena = nullptr; g.arena_bytes = 0;
}

#define FP_BIN(name, field, op) int name(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) { return fp_batch(field, op, a, b, out, n); }
#define FP_UN(name, field, op) int name(const uint64_t* a, uint64_t* out, size_t n) { return fp_batch(field, op, a, nullptr, out, n); }
// the four prime fields the reference instantiates: BLS12-381 Fq and Fr, secp256k1's base field (sp) and group order (sn)
#define FP_FIELD_API(P, F)                                                                                       \
  FP_BIN(zkt_##P##_add_batch, F, OP_ADD) FP_BIN(zkt_##P##_sub_batch, F, OP_SUB) FP_BIN(zkt_##P##_mul_batch, F, OP_MUL) \
  FP_UN(zkt_##P##_sqr_batch, F, OP_SQR) FP_UN(zkt_##P##_neg_batch, F, OP_NEG) FP_UN(zkt_##P##_inv_batch, F, OP_INV)     \
  FP_UN(zkt_##P##_cube_batch, F, OP_CUBE)                                                                        \
  int zkt_##P##_pow_batch(const uint64_t* a, const uint64_t* exps, size_t exp_limbs, int exp_shared, uint64_t* out, size_t n) { \
    return fp_pow_batch(F, a, exps, exp_limbs, exp_shared, out, n); }                                            \
  int zkt_##P##_sum(const uint64_t* a, size_t n, uint64_t* out) { return fp_sum(F, a, n, out); }                   \
  int zkt_##P##_scale_batch(const uint64_t* a, const uint64_t* k, uint64_t* out, size_t n) { return fp_scale(F, a, k, out, n); } \
  int zkt_##P##_pow_seq(const uint64_t* base, size_t n, uint64_t* out) { return fp_pow_seq(F, base, n, out, false); }  \
  int zkt_##P##_repeat(const uint64_t* base, size_t n, uint64_t* out) { return fp_pow_seq(F, base, n, out, true); }
FP_FIELD_API(fq, F_FQ) FP_FIELD_API(fr, F_FR) FP_FIELD_API(sp, F_SP) FP_FIELD_API(sn, F_SN)

#define TW_BIN(name, deg, op) int name(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) { return tower_batch(deg, op, a, b, out, n); }
#define TW_UN(name, deg, op) int name(const uint64_t* a, uint64_t* out, size_t n) { return tower_batch(deg, op, a, nullptr, out, n); }
TW_BIN(zkt_fq2_add_batch, 2, T_ADD) TW_BIN(zkt_fq2_sub_batch, 2, T_SUB) TW_BIN(zkt_fq2_mul_batch, 2, T_MUL)
TW_UN(zkt_fq2_inv_batch, 2, T_INV) TW_UN(zkt_fq2_neg_batch, 2, T_NEG) TW_UN(zkt_fq2_reduce_batch, 2, T_REDUCE)
TW_BIN(zkt_fq6_add_batch, 6, T_ADD) TW_BIN(zkt_fq6_sub_batch, 6, T_SUB) TW_BIN(zkt_fq6_mul_batch, 6, T_MUL)
TW_UN(zkt_fq6_inv_batch, 6, T_INV) TW_UN(zkt_fq6_neg_batch, 6, T_NEG) TW_UN(zkt_fq6_reduce_batch, 6, T_REDUCE)
TW_BIN(zkt_fq12_add_batch, 12, T_ADD) TW_BIN(zkt_fq12_sub_batch, 12, T_SUB) TW_BIN(zkt_fq12_mul_batch, 12, T_MUL)
TW_UN(zkt_fq12_inv_batch, 12, T_INV) TW_UN(zkt_fq12_neg_batch, 12, T_NEG)

int zkt_fq12_pow_batch(const uint64_t* a, const uint32_t* e, size_t nl, uint64_t* out, size_t n) {
  if (!e || nl == 0 || nl > 4096) return ZKT_ERR_SHAPE;
  // the exponent rides in front of the `a` staging area as a second input of fixed size
  if (ensure_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (n == 0) return ZKT_OK;
  if (!a || !out) return ZKT_ERR_SHAPE;
  std::lock_guard<std::mutex> lk(g.mu);
  HIPCHK(hipSetDevice(g.device));
  int rc = arena_reserve(padded(576 * n) * 2 + padded(nl * 4) + 1024);
  if (rc) return rc;
  Carver cv(g.arena);
  uint32_t* da = cv.take<uint32_t>(576 * n); uint32_t* dout = cv.take<uint32_t>(576 * n); uint32_t* de = cv.take<uint32_t>(nl * 4);
  HIPCHK(hipMemcpyAsync(da, a, 576 * n, hipMemcpyHostToDevice, g.stream));
  HIPCHK(hipMemcpyAsync(de, e, nl * 4, hipMemcpyHostToDevice, g.stream));
  HIPCHK(launch_fq12_pow(da, de, (int)nl, dout, n, g.stream));
  HIPCHK(hipMemcpyAsync(out, dout, 576 * n, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  return ZKT_OK;
}

// diagnostic: the lazy-limb Fq self-test program (fq_program.h) on the device, one run per lane
int zkt_selftest_fq_program(uint64_t seed0, int steps, const uint64_t* in, uint64_t* out, int32_t* violations, size_t count) {
  if (ensure_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (count == 0) return ZKT_OK;
  if (!in || !out || !violations || steps < 0) return ZKT_ERR_SHAPE;
  std::lock_guard<std::mutex> lk(g.mu);
  HIPCHK(hipSetDevice(g.device));
  const size_t eb = 4 * 48;
  int rc = arena_reserve(padded(eb * count) * 2 + padded(4 * count) + 1024);
  if (rc) return rc;
  Carver cv(g.arena);
  uint32_t* din = cv.take<uint32_t>(eb * count); uint32_t* dout = cv.take<uint32_t>(eb * count); int* dbad = (int*)cv.take<uint32_t>(4 * count);
  HIPCHK(hipMemcpyAsync(din, in, eb * count, hipMemcpyHostToDevice, g.stream));
  HIPCHK(launch_selftest_fq_program(seed0, steps, din, dout, dbad, count, g.stream));
  HIPCHK(hipMemcpyAsync(out, dout, eb * count, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipMemcpyAsync(violations, dbad, 4 * count, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  return ZKT_OK;
}

// diagnostic: Fq12 operations of the lane-distributed pairing (zkt_dpairing.hip), same layouts as zkt_fq12_*_batch
int zkt_debug_dfq12_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  if (op < 0 || op > 6) return ZKT_ERR_SHAPE;
  return staged(a, 576, op == 0 ? b : nullptr, op == 0 ? 576 : 0, out, 576, n, ZKT_ERR_SHAPE,
                [&](uint32_t* da, uint32_t* db, uint32_t* dout, hipStream_t s) { return launch_dfq12_op(op, da, db, dout, n, s); });
}

static int group_add(int grp, const void* a, const void* b, void* out, size_t n) {
  size_t w = pt_bytes(grp);
  return staged(a, w, b, w, out, w, n, ZKT_ERR_SHAPE,
                [&](uint32_t* da, uint32_t* db, uint32_t* dout, hipStream_t s) { return launch_group_add(grp, da, db, dout, n, s); });
}
static int group_neg(int grp, const void* a, void* out, size_t n) {
  size_t w = pt_bytes(grp);
  return staged(a, w, nullptr, 0, out, w, n, ZKT_ERR_SHAPE,
                [&](uint32_t* da, uint32_t*, uint32_t* dout, hipStream_t s) { return launch_group_neg(grp, da, dout, n, s); });
}
static int group_mul(int grp, const void* pts, const uint64_t* scalars, int limbs, void* out, size_t n) {
  if (limbs < 1 || limbs > 6) return ZKT_ERR_SHAPE;
  size_t w = pt_bytes(grp);
  return staged(pts, w, scalars, (size_t)limbs * 8, out, w, n, ZKT_ERR_SHAPE,
                [&](uint32_t* da, uint32_t* db, uint32_t* dout, hipStream_t s) { return launch_group_mul(grp, da, db, limbs * 2, dout, n, s); });
}
// AffinePoints::sum (secp256k1/affine_points.rs:25-31: the fold from AffinePoint::zero(), so an empty vector sums to infinity) and
// AffinePoints * PrimeFieldElem (:105-122): every point times ONE scalar
static int group_sum(int grp, const void* pts, size_t n, void* out) {
  if (ensure_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  const size_t w = pt_bytes(grp);
  if (!out || (n && !pts)) return ZKT_ERR_SHAPE;
  if (n == 0) { memset(out, 0, w); ((uint32_t*)out)[w / 4 - 2] = 1; return ZKT_OK; }
  std::lock_guard<std::mutex> lk(g.mu);
  HIPCHK(hipSetDevice(g.device));
  int rc = arena_reserve(padded(w * (n + 1)) + 1024);
  if (rc) return rc;
  Carver cv(g.arena);
  uint32_t* da = cv.take<uint32_t>(w * (n + 1));
  HIPCHK(hipMemcpyAsync(da, pts, w * n, hipMemcpyHostToDevice, g.stream));
  if (n == 1) {                                                      // zero + p: goes through the addition so that the coordinates come back reduced
    std::vector<uint8_t> inf(w, 0); ((uint32_t*)inf.data())[w / 4 - 2] = 1;
    HIPCHK(hipMemcpyAsync((uint8_t*)da + w, inf.data(), w, hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
  }
  HIPCHK(launch_group_sum_inplace(grp, da, n == 1 ? 2 : n, g.stream));            // result in da[0]
  HIPCHK(hipMemcpyAsync(out, da, w, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  return ZKT_OK;
}
static int group_scale(int grp, const void* pts, const uint64_t* k, int limbs, void* out, size_t n) {
  if (limbs < 1 || limbs > 6) return ZKT_ERR_SHAPE;
  if (n == 0) return ensure_ready();
  size_t w = pt_bytes(grp);
  return staged_raw(pts, w * n, k, (size_t)limbs * 8, out, w * n, ZKT_ERR_SHAPE,
                    [&](uint32_t* da, uint32_t* db, uint32_t* dout, hipStream_t s) { return launch_group_mul(grp, da, db, limbs * 2, dout, n, s, false, true); });
}
int zkt_g1_sum(const zkt_g1_affine* p, size_t n, zkt_g1_affine* o) { return group_sum(G_G1, p, n, o); }
int zkt_g2_sum(const zkt_g2_affine* p, size_t n, zkt_g2_affine* o) { return group_sum(G_G2, p, n, o); }
int zkt_secp_sum(const zkt_secp_affine* p, size_t n, zkt_secp_affine* o) { return group_sum(G_SECP, p, n, o); }
int zkt_g1_scale_batch(const zkt_g1_affine* p, const uint64_t* k, int l, zkt_g1_affine* o, size_t n) { return group_scale(G_G1, p, k, l, o, n); }
int zkt_g2_scale_batch(const zkt_g2_affine* p, const uint64_t* k, int l, zkt_g2_affine* o, size_t n) { return group_scale(G_G2, p, k, l, o, n); }
int zkt_secp_scale_batch(const zkt_secp_affine* p, const uint64_t* k, int l, zkt_secp_affine* o, size_t n) { return group_scale(G_SECP, p, k, l, o, n); }
int zkt_g1_add_batch(const zkt_g1_affine* a, const zkt_g1_affine* b, zkt_g1_affine* o, size_t n) { return group_add(G_G1, a, b, o, n); }
int zkt_g2_add_batch(const zkt_g2_affine* a, const zkt_g2_affine* b, zkt_g2_affine* o, size_t n) { return group_add(G_G2, a, b, o, n); }
int zkt_secp_add_batch(const zkt_secp_affine* a, const zkt_secp_affine* b, zkt_secp_affine* o, size_t n) { return group_add(G_SECP, a, b, o, n); }
int zkt_g1_neg_batch(const zkt_g1_affine* a, zkt_g1_affine* o, size_t n) { return group_neg(G_G1, a, o, n); }
int zkt_g2_neg_batch(const zkt_g2_affine* a, zkt_g2_affine* o, size_t n) { return group_neg(G_G2, a, o, n); }
int zkt_g1_mul_batch(const zkt_g1_affine* p, const uint64_t* k, int l, zkt_g1_affine* o, size_t n) { return group_mul(G_G1, p, k, l, o, n); }
int zkt_g2_mul_batch(const zkt_g2_affine* p, const uint64_t* k, int l, zkt_g2_affine* o, size_t n) { return group_mul(G_G2, p, k, l, o, n); }
int zkt_secp_mul_batch(const zkt_secp_affine* p, const uint64_t* k, int l, zkt_secp_affine* o, size_t n) { return group_mul(G_SECP, p, k, l, o, n); }

// a16: is_rational_point / order-r membership / generators
static int group_pred(int grp, int pred, const void* pts, uint32_t* out, size_t n) {
  static const uint32_t R_WORDS[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};      // params.rs:14
  static const uint32_t SN_WORDS[8] = {0xd0364141u, 0xbfd25e8cu, 0xaf48a03bu, 0xbaaedce6u, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0xffffffffu};   // secp256k1 n
  return staged(pts, pt_bytes(grp), grp == G_SECP ? SN_WORDS : R_WORDS, 0, out, 4, n, ZKT_ERR_SHAPE,
                [&](uint32_t* da, uint32_t*, uint32_t* dout, hipStream_t s) -> hipError_t {
                  uint32_t* d_order = g.d_small;                           // g.mu is held by staged_raw
                  hipError_t e = hipMemcpyAsync(d_order, grp == G_SECP ? SN_WORDS : R_WORDS, 32, hipMemcpyHostToDevice, s);
                  if (e != hipSuccess) return e;
                  return launch_group_pred(grp, pred, da, d_order, 8, dout, n, s);
                });
}
int zkt_g1_is_on_curve_batch(const zkt_g1_affine* p, uint32_t* out, size_t n) { return group_pred(G_G1, 0, p, out, n); }
int zkt_g2_is_on_curve_batch(const zkt_g2_affine* p, uint32_t* out, size_t n) { return group_pred(G_G2, 0, p, out, n); }
int zkt_secp_is_on_curve_batch(const zkt_secp_affine* p, uint32_t* out, size_t n) { return group_pred(G_SECP, 0, p, out, n); }
int zkt_g1_in_subgroup_batch(const zkt_g1_affine* p, uint32_t* out, size_t n) { return group_pred(G_G1, 1, p, out, n); }
int zkt_g2_in_subgroup_batch(const zkt_g2_affine* p, uint32_t* out, size_t n) { return group_pred(G_G2, 1, p, out, n); }
int zkt_secp_in_subgroup_batch(const zkt_secp_affine* p, uint32_t* out, size_t n) { return group_pred(G_SECP, 1, p, out, n); }
static void put_limbs(uint64_t* dst, const char* hex, int limbs) {           // big-endian hex literal -> little-endian u64 limbs
  const size_t len = strlen(hex);
  for (int i = 0; i < limbs; ++i) {
    uint64_t v = 0;
    for (int d = 0; d < 16; ++d) {
      const long pos = (long)len - 16 * (i + 1) + d;
      if (pos < 0) continue;
      const char c = hex[pos];
      v = (v << 4) | (uint64_t)(c <= '9' ? c - '0' : (c | 32) - 'a' + 10);
    }
    dst[i] = v;
  }
}
void zkt_g1_generator(zkt_g1_affine* out) {                                   // g1_point.rs:38-47
  memset(out, 0, sizeof(*out));
  put_limbs(out->x, "17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb", 6);
  put_limbs(out->y, "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1", 6);
}
void zkt_g2_generator(zkt_g2_affine* out) {                                   // g2_point.rs:36-46; Fq2 = {u1, u0}
  memset(out, 0, sizeof(*out));
  put_limbs(out->x, "13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e", 6);
  put_limbs(out->x + 6, "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8", 6);
  put_limbs(out->y, "0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be", 6);
  put_limbs(out->y + 6, "0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801", 6);
}
void zkt_secp_generator(zkt_secp_affine* out) {                               // secp256k1/affine_point.rs:40-47
  memset(out, 0, sizeof(*out));
  put_limbs(out->x, "79be667ef9dcbbac55a06295ce870b07029bfcdb2dce28d959f2815b16f81798", 4);
  put_limbs(out->y, "483ada7726a3c4655da4fbfc0e1108a8fd17b448a68554199c47d08ffb10d4b8", 4);
}

int zkt_tate_batch(const zkt_g1_affine* g1, const zkt_g2_affine* g2, uint64_t* out, size_t n) {
  return staged(g1, sizeof(zkt_g1_affine), g2, sizeof(zkt_g2_affine), out, 576, n, ZKT_ERR_INFINITY,
                [&](uint32_t* da, uint32_t* db, uint32_t* dout, hipStream_t s) { return launch_tate(da, db, dout, n, g.d_err, s); });
}
static int miller_exact(int which, const zkt_g1_affine* g1, const zkt_g2_affine* g2, uint64_t* out, size_t n) {
  return staged(g1, sizeof(zkt_g1_affine), g2, sizeof(zkt_g2_affine), out, 576, n, ZKT_ERR_INFINITY,
                [&](uint32_t* da, uint32_t* db, uint32_t* dout, hipStream_t s) { return launch_miller_exact(which, da, db, dout, n, g.d_err, s); });
}
int zkt_miller_g1g2_batch(const zkt_g1_affine* g1, const zkt_g2_affine* g2, uint64_t* out, size_t n) { return miller_exact(0, g1, g2, out, n); }
int zkt_miller_g2g1_batch(const zkt_g2_affine* g2, const zkt_g1_affine* g1, uint64_t* out, size_t n) { return miller_exact(1, g1, g2, out, n); }
int zkt_weil_batch(const zkt_g1_affine* g1, const zkt_g2_affine* g2, uint64_t* out, size_t n) { return miller_exact(2, g1, g2, out, n); }
int zkt_gt_eq(const uint64_t* a, const uint64_t* b) {
  if (!a || !b) return -ZKT_ERR_SHAPE;
  return memcmp(a, b, 576) == 0 ? 1 : 0;   // canonical residues: memcmp equality <=> Fq12 equality (fq12.rs:88-93)
}

// ---- device-resident entry points --------------------------------------------------
int zkt_fq_mul_batch_dev(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream) {
  if (ensure_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  HIPCHK(launch_fp_op(F_FQ, OP_MUL, (const uint32_t*)a, (const uint32_t*)b, (uint32_t*)out, n, g.d_err, (hipStream_t)stream));
  return ZKT_OK;
}
int zkt_g1_mul_batch_dev(const zkt_g1_affine* p, const uint64_t* k, int limbs, zkt_g1_affine* out, size_t n, void* stream) {
  if (ensure_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (limbs < 1 || limbs > 6) return ZKT_ERR_SHAPE;
  HIPCHK(launch_group_mul(G_G1, (const uint32_t*)p, (const uint32_t*)k, limbs * 2, (uint32_t*)out, n, (hipStream_t)stream));
  return ZKT_OK;
}
int zkt_g2_mul_batch_dev(const zkt_g2_affine* p, const uint64_t* k, int limbs, zkt_g2_affine* out, size_t n, void* stream) {
  if (ensure_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (limbs < 1 || limbs > 6) return ZKT_ERR_SHAPE;
  HIPCHK(launch_group_mul(G_G2, (const uint32_t*)p, (const uint32_t*)k, limbs * 2, (uint32_t*)out, n, (hipStream_t)stream));
  return ZKT_OK;
}
// The pairing kernels keep their Fq12 temporaries in 10-18 KB of scratch per lane, and the runtime sizes — and keeps — a queue's scratch for every wave slot
// of the device: 5-6 GiB per hardware queue that ever ran one (the process aborts when its pool is spent, at ~30 GB).  A caller with
// eight streams of its own would spend 8 x 6 GiB on them.  So the kernel never runs on the caller's stream: it runs on the library's ONE staging stream,
// ordered behind the caller's stream by an event (the inputs were produced there), and the call returns after the result is complete, so work the caller
// queues afterwards on ANY stream sees it.  tests/test_gpu_parity.py::test_tate_dev_from_many_caller_streams drives 8 streams through this.
// Three persistent events of the staging stream (created on first use under g.mu, released with the library's other device state): nothing is created or
// destroyed per call, and an early return leaks nothing.
static hipEvent_t g_tate_ev[3] = {nullptr, nullptr, nullptr};
static int tate_events() {
  if (g_tate_ev[0]) return ZKT_OK;
  hipEvent_t e[3] = {nullptr, nullptr, nullptr};
  if (hipEventCreateWithFlags(&e[0], hipEventDisableTiming) != hipSuccess || hipEventCreate(&e[1]) != hipSuccess || hipEventCreate(&e[2]) != hipSuccess) {
    for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x);
    (void)hipGetLastError(); return ZKT_ERR_DEVICE;
  }
  for (int i = 0; i < 3; ++i) g_tate_ev[i] = e[i];
  return ZKT_OK;
}
static void tate_events_release() { for (hipEvent_t& x : g_tate_ev) { if (x) (void)hipEventDestroy(x); x = nullptr; } }
// BLOCKING, unlike the other *_dev entry points (include/zkt.h): the result is complete on return, and calls are serialised on the staging stream.
int zkt_tate_batch_dev(const zkt_g1_affine* g1, const zkt_g2_affine* g2, uint64_t* out, size_t n, void* stream) {
  if (ensure_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  std::lock_guard<std::mutex> lk(g.mu);
  hipStream_t s = g.stream;
  int rc = tate_events(); if (rc) return rc;
  HIPCHK(hipEventRecord(g_tate_ev[0], (hipStream_t)stream));
  HIPCHK(hipStreamWaitEvent(s, g_tate_ev[0], 0));
  rc = reset_err(s); if (rc) return rc;
  HIPCHK(hipEventRecord(g_tate_ev[1], s));
  HIPCHK(launch_tate((const uint32_t*)g1, (const uint32_t*)g2, (uint32_t*)out, n, g.d_err, s));
  HIPCHK(hipEventRecord(g_tate_ev[2], s));
  rc = fetch_err(s, ZKT_ERR_INFINITY);                  // synchronises the staging stream: the pairings are done when this returns
  HIPCHK(hipEventElapsedTime(&t_kernel_ms, g_tate_ev[1], g_tate_ev[2])); t_kernel_name = "k_tate";
  return rc;
}

static int bases_build(zkt_bases_impl* h, const uint32_t* dev_abi, hipStream_t s) {
  const size_t n = h->n;
  h->plan = msm_plan(n, h->grp);
  const size_t tot = (size_t)h->plan.nwin * (n ? n : 1);
  HIPCHK(hipMalloc((void**)&h->table, tot * 2 * grp_coord_bytes(h->grp)));
  HIPCHK(hipMalloc((void**)&h->inf, tot));
  HIPCHK(launch_msm_to_kernel_layout(h->grp, dev_abi, h->table, h->inf, n, s));
  uint32_t* tmp = nullptr;                                   // Z and prefix products of the per-lane batched normalisation
  if (h->plan.nwin > 1) HIPCHK(hipMalloc((void**)&tmp, (size_t)(h->plan.nwin - 1) * (n ? n : 1) * 2 * grp_coord_bytes(h->grp)));
  hipError_t e = launch_msm_precompute(h->grp, h->table, h->inf, n, h->plan.c, h->plan.nwin, tmp, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (tmp) hipFree(tmp);
  HIPCHK(e);
  return ZKT_OK;
}
static void bases_free(zkt_bases_impl* h) {
  if (!h) return;
  if (h->table) hipFree(h->table);
  if (h->inf) hipFree(h->inf);
  if (h->own_streams) {
    for (hipStream_t st : {h->s_sort, h->s_acc}) if (st) { hipStreamSynchronize(st); hipStreamDestroy(st); }
    for (hipStream_t st : h->s_tail) if (st) { hipStreamSynchronize(st); hipStreamDestroy(st); }
  } else {                                     // borrowed streams: wait for this set's own work, leave them to their owner (freed after the borrowers)
    for (hipStream_t st : {h->s_sort, h->s_acc}) if (st) hipStreamSynchronize(st);
    for (hipStream_t st : h->s_tail) if (st) hipStreamSynchronize(st);
    if (h->acc_owned && h->s_acc) hipStreamDestroy(h->s_acc);
  }
  for (MsmSlot& S : h->slot) {
    for (hipEvent_t ev : {S.e_in, S.e_sorted, S.e_acc0, S.e_acc1, S.e_done}) if (ev) hipEventDestroy(ev);
    for (hipGraphExec_t& ge : S.gexec) if (ge) { (void)hipGraphExecDestroy(ge); ge = nullptr; }
    if (S.workspace) hipFree(S.workspace); if (S.d_result_jac) hipFree(S.d_result_jac); if (S.d_out_abi) hipFree(S.d_out_abi);
    if (S.h_out) hipHostFree(S.h_out);
  }
  delete h;
}
static int bases_from_device(int grp, const void* dev_bases, size_t n, void* stream, zkt_bases_impl** out) {
  if (ensure_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (!out || (n && !dev_bases) || n >= (size_t(1) << 26)) return ZKT_ERR_SHAPE;
  zkt_bases_impl* h = new zkt_bases_impl(); h->n = n; h->grp = grp;
  int rc = bases_build(h, (const uint32_t*)dev_bases, (hipStream_t)stream);
  if (rc) { bases_free(h); return rc; }
  *out = h; return ZKT_OK;
}
static int bases_upload(int grp, const void* host, size_t n, zkt_bases_impl** out) {
  if (ensure_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (!out || (n && !host)) return ZKT_ERR_SHAPE;
  HIPCHK(hipSetDevice(g.device));
  uint32_t* tmp = nullptr;
  HIPCHK(hipMalloc((void**)&tmp, (n ? n : 1) * grp_pt_bytes(grp)));
  if (n) HIPCHK(hipMemcpy(tmp, host, n * grp_pt_bytes(grp), hipMemcpyHostToDevice));
  int rc = bases_from_device(grp, tmp, n, g.stream, out);
  hipFree(tmp);
  return rc;
}
static hipStream_t slot_tail_stream(zkt_bases_impl* h, int slot) {
  if (h->grouped) return h->s_tail[(h->tail_base + slot % h->tail_span) % zkt_bases_impl::GROUP_TAILS];      // a span may wrap around the group's four reduce streams
  const bool small = h->n < (size_t(1) << 19);
  return h->s_tail[small ? slot % zkt_bases_impl::NTAIL : slot % 2];
}
// caller holds h->mu
static int msm_submit_locked(zkt_bases_impl* h, const uint64_t* dev_scalars, size_t n, void* stream, int slot) {
  if (n != h->n || (n && !dev_scalars) || slot < 0 || slot >= MSM_SLOTS) return ZKT_ERR_SHAPE;
  if (h->slot[slot].busy) return ZKT_ERR_SHAPE;          // collect it first
  int rc = slot_ready(h, slot); if (rc) return rc;
  MsmSlot& S = h->slot[slot];
  // inputs are produced on the caller's stream: order the sort stage behind it
  HIPCHK(hipEventRecord(S.e_in, (hipStream_t)stream));
  const bool small = h->n < (size_t(1) << 19);       // (a set that shares a key's streams runs on the group's reduce stream for this slot: slot_tail_stream)
  hipStream_t st = slot_tail_stream(h, slot);
  hipStream_t ss = small ? st : h->s_sort;
  HIPCHK(hipStreamWaitEvent(ss, S.e_in, 0));
  if (small && msm_graphs()) {
    // A small MSM is ~20 launches of a few microseconds of GPU time each: the protocols that run several of them side by side (a range proof's five, a Pinocchio proof's
    // ten) are bound by the rate at which the host can submit them.  The whole pipeline of a slot — memsets, sort, accumulate, reduce, the copy of the result — touches only
    // the slot's own buffers and the scalar vector (kernel launches only: the counters are cleared by a kernel of the pipeline's own), so it is captured ONCE per (slot, scalar address) on the slot's stream and replayed as one graph launch.  Nothing inside
    // the capture waits on or records an event (the input dependency is the stream wait above, completion is e_done below); the slot is not busy here, so no launch of an
    // executable graph that gets evicted is still in flight.
    int hit = -1;
    for (int k = 0; k < MsmSlot::NGRAPH; ++k) if (S.gexec[k] && S.gkey[k] == (const void*)dev_scalars) hit = k;
    if (hit < 0) {
      hit = (int)(S.gnext++ % MsmSlot::NGRAPH);
      if (S.gexec[hit]) { (void)hipGraphExecDestroy(S.gexec[hit]); S.gexec[hit] = nullptr; }
      hipGraph_t graph = nullptr;
      HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
      hipError_t e = launch_msm_sort(h->plan, h->inf, (const uint32_t*)dev_scalars, S.workspace, st);
      if (e == hipSuccess) e = launch_msm_accumulate(h->plan, h->table, S.workspace, st);
      if (e == hipSuccess) e = launch_msm_reduce(h->plan, S.workspace, S.d_result_jac, S.d_out_abi, st);
      const hipError_t e2 = hipStreamEndCapture(st, &graph);                 // always: the stream must leave capture mode
      if (e != hipSuccess || e2 != hipSuccess || !graph) { if (graph) (void)hipGraphDestroy(graph); HIPCHK(e != hipSuccess ? e : (e2 != hipSuccess ? e2 : hipErrorUnknown)); }
      e = hipGraphInstantiate(&S.gexec[hit], graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (e != hipSuccess) { S.gexec[hit] = nullptr; HIPCHK(e); }
      S.gkey[hit] = (const void*)dev_scalars;
    }
    HIPCHK(hipGraphLaunch(S.gexec[hit], st));
    HIPCHK(hipMemcpyAsync(S.h_out, S.d_out_abi, grp_pt_bytes(h->grp), hipMemcpyDeviceToHost, st));      // kernels only inside the graph: the copy of the result follows it on the stream
    HIPCHK(hipEventRecord(S.e_done, st));
    S.timed = false; S.busy = true;
    return ZKT_OK;
  }
  S.timed = true;
  HIPCHK(launch_msm_sort(h->plan, h->inf, (const uint32_t*)dev_scalars, S.workspace, ss));
  HIPCHK(hipEventRecord(S.e_sorted, ss));
  // a large MSM fills the chip, so its stages queue on per-stage streams (sort of MSM k+1 under the accumulation of MSM k); below 2^19 terms every
  // stage is a latency-bound sliver of the chip (one short wave per SIMD), so each slot runs its whole MSM on its own stream, side by side with the others
  hipStream_t sa = small ? st : h->s_acc;
  HIPCHK(hipStreamWaitEvent(sa, S.e_sorted, 0));
  HIPCHK(hipEventRecord(S.e_acc0, sa));
  HIPCHK(launch_msm_accumulate(h->plan, h->table, S.workspace, sa));
  HIPCHK(hipEventRecord(S.e_acc1, sa));
  HIPCHK(hipStreamWaitEvent(st, S.e_acc1, 0));
  HIPCHK(launch_msm_reduce(h->plan, S.workspace, S.d_result_jac, S.d_out_abi, st));
  HIPCHK(hipMemcpyAsync(S.h_out, S.d_out_abi, grp_pt_bytes(h->grp), hipMemcpyDeviceToHost, st));
  HIPCHK(hipEventRecord(S.e_done, st));
  S.busy = true;
  return ZKT_OK;
}
static int msm_collect_locked(zkt_bases_impl* h, int slot, void* out, uint32_t* dev_partial_jac) {
  if (slot < 0 || slot >= MSM_SLOTS || !h->slot[slot].busy) return ZKT_ERR_SHAPE;
  MsmSlot& S = h->slot[slot];
  HIPCHK(hipEventSynchronize(S.e_done));
  if (dev_partial_jac) {        // copied on the slot's own tail stream and waited for: complete when this returns, and never overtaken by a re-submit of the slot
    hipStream_t st = slot_tail_stream(h, slot);
    HIPCHK(hipMemcpyAsync(dev_partial_jac, S.d_result_jac, 3 * grp_coord_bytes(h->grp), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  if (out) memcpy(out, S.h_out, grp_pt_bytes(h->grp));
  float ms = 0.f;
  if (S.timed && hipEventElapsedTime(&ms, S.e_acc0, S.e_acc1) == hipSuccess) { t_kernel_ms = ms; t_kernel_name = "k_accumulate"; }
  else { t_kernel_ms = 0.f; t_kernel_name = "msm_graph"; }      // a graph-replayed MSM carries no per-kernel events: say so instead of leaving the previous operation's figures
  S.busy = false;
  return ZKT_OK;
}
static int msm_submit(zkt_bases_impl* h, const uint64_t* dev_scalars, size_t n, void* stream, int slot) {
  if (ensure_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (!h) return ZKT_ERR_SHAPE;
  std::lock_guard<std::mutex> lk(h->mu);
  return msm_submit_locked(h, dev_scalars, n, stream, slot);
}
static int msm_collect(zkt_bases_impl* h, int slot, void* out, uint32_t* dev_partial_jac) {
  if (ensure_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (!h) return ZKT_ERR_SHAPE;
  std::lock_guard<std::mutex> lk(h->mu);
  return msm_collect_locked(h, slot, out, dev_partial_jac);
}
// blocking form: slot 0, submit + collect under one hold of the handle's lock (two threads may share a handle)
static int msm_dev(zkt_bases_impl* h, const uint64_t* dev_scalars, size_t n, void* stream, void* out, uint32_t* dev_partial_jac) {
  if (ensure_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (!h || (!out && !dev_partial_jac)) return ZKT_ERR_SHAPE;
  std::lock_guard<std::mutex> lk(h->mu);
  int rc = msm_submit_locked(h, dev_scalars, n, stream, 0);
  if (rc) return rc;
  return msm_collect_locked(h, 0, out, dev_partial_jac);
}
// `dst` works on `src`'s streams from now on (sort stream and reduce streams [tail_base, tail_base + tail_span); the accumulate stream too if share_acc,
// otherwise dst gets one of its own).  Call before dst's first MSM; free dst before src.
int zkt_internal_bases_share_streams(void* dst_, void* src_, int share_acc, int tail_base, int tail_span) {
  zkt_bases_impl *dst = (zkt_bases_impl*)dst_, *src = (zkt_bases_impl*)src_;
  if (!dst || !src || tail_span < 1 || tail_base < 0) return ZKT_ERR_SHAPE;
  constexpr int GROUP_TAILS = zkt_bases_impl::GROUP_TAILS;
  if (tail_base >= GROUP_TAILS || tail_span > GROUP_TAILS) return ZKT_ERR_SHAPE;
  int lo = 0, hi = 0; HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  {
    std::lock_guard<std::mutex> lk(src->mu);
    if (!src->grouped) {                       // the owner: exactly the streams the group uses (a stream that exists claims a hardware queue)
      if (src->s_sort || src->s_acc) return ZKT_ERR_SHAPE;
      HIPCHK(hipStreamCreateWithPriority(&src->s_sort, hipStreamNonBlocking, hi)); HIPCHK(hipStreamCreateWithPriority(&src->s_acc, hipStreamNonBlocking, lo));
      for (int k = 0; k < GROUP_TAILS; ++k) HIPCHK(hipStreamCreateWithPriority(&src->s_tail[k], hipStreamNonBlocking, hi));
      src->grouped = true; src->tail_base = 0; src->tail_span = 2;
    }
  }
  std::lock_guard<std::mutex> lk(dst->mu);
  if (dst->s_sort || dst->s_acc) return ZKT_ERR_SHAPE;
  dst->own_streams = false; dst->grouped = true; dst->s_sort = src->s_sort;
  for (int k = 0; k < GROUP_TAILS; ++k) dst->s_tail[k] = src->s_tail[k];
  dst->tail_base = tail_base; dst->tail_span = tail_span;
  if (share_acc) dst->s_acc = src->s_acc;
  else { HIPCHK(hipStreamCreateWithPriority(&dst->s_acc, hipStreamNonBlocking, lo)); dst->acc_owned = true; }
  return ZKT_OK;
}
// combine step of a sharded MSM: `count` Jacobian partials, `stride_words` u32 apart, summed by one wave and normalised
int zkt_internal_jac_sum(int grp, const uint32_t* dev_partials, size_t count, size_t stride_words, hipStream_t s, void* out) {
  if (!dev_partials || !out || count == 0) return ZKT_ERR_SHAPE;
  std::lock_guard<std::mutex> lk(g.mu);
  HIPCHK(launch_msm_jac_sum_to_affine(grp, dev_partials, count, stride_words, g.d_small, s));
  HIPCHK(hipMemcpyAsync(out, g.d_small, grp_pt_bytes(grp), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKT_OK;
}
static int jac_sum_dev(int grp, const uint32_t* dev_partials, size_t count, void* stream, void* out) {
  if (ensure_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  return zkt_internal_jac_sum(grp, dev_partials, count, 3 * grp_coord_bytes(grp) / 4, (hipStream_t)stream, out);
}
// one-shot host-pointer MSM: upload, build the window-multiple table, run, free
// one-shot host-pointer MSM (eval_with_g1_hidings called once, polynomial.rs:271-281): the table-free form — upload, kernel layout,
// sort / accumulate / per-window reduce / join on one stream, free.  No window-multiple table is built for a single use.
// ZKT_MSM_ONE_SHOT_TABLE=1 selects the resident-bases machinery instead (A/B and regression checks).
static int msm_host(int grp, const void* bases, const uint64_t* scalars, size_t n, void* out) {
  if (ensure_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (!out || (n && (!bases || !scalars)) || n >= (size_t(1) << 26)) return ZKT_ERR_SHAPE;      // entry offsets are 32-bit: nwin * n < 2^32
  if (n == 0) { memset(out, 0, grp_pt_bytes(grp)); ((uint32_t*)out)[grp_pt_bytes(grp) / 4 - 2] = 1; return ZKT_OK; }
  static const bool use_table = [] { const char* e = getenv("ZKT_MSM_ONE_SHOT_TABLE"); return e && *e == '1'; }();
  if (use_table) {
    zkt_bases_impl* h = nullptr;
    int rc = bases_upload(grp, bases, n, &h);
    if (rc) return rc;
    uint64_t* d_s = nullptr;
    if (hipMalloc((void**)&d_s, n * 32) != hipSuccess) { bases_free(h); return ZKT_ERR_DEVICE; }
    hipMemcpy(d_s, scalars, n * 32, hipMemcpyHostToDevice);
    rc = msm_dev(h, d_s, n, g.stream, out, nullptr);
    hipFree(d_s); bases_free(h);
    return rc;
  }
  const MsmPlan plan = msm_plan_direct(n, grp);
  const size_t ptb = grp_pt_bytes(grp), cb = grp_coord_bytes(grp);
  uint8_t* blob = nullptr;                                   // [abi points | scalars | kernel-layout points | inf flags | jac | abi out | workspace]
  const size_t o_abi = 0, o_sc = padded(n * ptb), o_tab = o_sc + padded(n * 32), o_inf = o_tab + padded(n * 2 * cb), o_jac = o_inf + padded(n),
               o_out = o_jac + padded(4 * cb), o_ws = o_out + padded(ptb), total = o_ws + plan.ws_bytes;
  HIPCHK(hipMalloc((void**)&blob, total));
  if (debug_poison()) { HIPCHK(hipMemset(blob, 0xA5, total)); HIPCHK(hipDeviceSynchronize()); }
  std::lock_guard<std::mutex> lk(g.mu);                      // g.stream is the library's staging stream
  hipStream_t s = g.stream;
  int rc = ZKT_OK;
  auto fail = [&](hipError_t e) { if (e != hipSuccess) { fprintf(stderr, "[zkt] HIP error %s in one-shot MSM\n", hipGetErrorString(e)); rc = ZKT_ERR_DEVICE; } return e != hipSuccess; };
  if (!fail(hipMemcpyAsync(blob + o_abi, bases, n * ptb, hipMemcpyHostToDevice, s)) && !fail(hipMemcpyAsync(blob + o_sc, scalars, n * 32, hipMemcpyHostToDevice, s)) &&
      !fail(launch_msm_to_kernel_layout(grp, (const uint32_t*)(blob + o_abi), (uint32_t*)(blob + o_tab), blob + o_inf, n, s)) &&
      !fail(launch_msm_sort(plan, blob + o_inf, (const uint32_t*)(blob + o_sc), blob + o_ws, s)) &&
      !fail(launch_msm_accumulate(plan, (const uint32_t*)(blob + o_tab), blob + o_ws, s)) &&
      !fail(launch_msm_reduce(plan, blob + o_ws, (uint32_t*)(blob + o_jac), (uint32_t*)(blob + o_out), s)) &&
      !fail(hipMemcpyAsync(out, blob + o_out, ptb, hipMemcpyDeviceToHost, s)))
    fail(hipStreamSynchronize(s));
  else hipStreamSynchronize(s);
  hipFree(blob);
  return rc;
}

#define ZKT_BASES_API(NAME, GRP, PT)                                                                                             \
  int zkt_##NAME##_bases_from_device(const PT* dev, size_t n, void* stream, zkt_##NAME##_bases** out) {                          \
    return bases_from_device(GRP, dev, n, stream, (zkt_bases_impl**)out); }                                                     \
  int zkt_##NAME##_bases_upload(const PT* host, size_t n, zkt_##NAME##_bases** out) { return bases_upload(GRP, host, n, (zkt_bases_impl**)out); } \
  size_t zkt_##NAME##_bases_len(const zkt_##NAME##_bases* b) { return b ? b->n : 0; }                                             \
  void zkt_##NAME##_bases_free(zkt_##NAME##_bases* b) { bases_free(b); }                                                          \
  int zkt_##NAME##_msm_submit(zkt_##NAME##_bases* b, const uint64_t* k, size_t n, void* stream, int slot) { return msm_submit(b, k, n, stream, slot); } \
  int zkt_##NAME##_msm_collect(zkt_##NAME##_bases* b, int slot, PT* out, uint32_t* partial) { return msm_collect(b, slot, out, partial); } \
  int zkt_##NAME##_msm_dev(const zkt_##NAME##_bases* b, const uint64_t* k, size_t n, void* stream, PT* out, uint32_t* partial) { \
    return msm_dev(const_cast<zkt_##NAME##_bases*>(b), k, n, stream, out, partial); }                                              \
  int zkt_##NAME##_jac_sum_dev(const uint32_t* partials, size_t count, void* stream, PT* out) { return jac_sum_dev(GRP, partials, count, stream, out); } \
  int zkt_##NAME##_msm(const PT* bases, const uint64_t* scalars, size_t n, PT* out) { return msm_host(GRP, bases, scalars, n, out); }
ZKT_BASES_API(g1, G_G1, zkt_g1_affine)
ZKT_BASES_API(g2, G_G2, zkt_g2_affine)
ZKT_BASES_API(secp, G_SECP, zkt_secp_affine)
size_t zkt_g1_msm_workspace_bytes(size_t n) { MsmPlan p = msm_plan(n, G_G1); return p.ws_bytes + (size_t)p.nwin * n * 97; }

}  // extern "C"
