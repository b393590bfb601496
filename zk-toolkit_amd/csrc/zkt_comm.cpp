// Multi-GPU inside the C ABI: one process per GPU, the MSM / proof index range partitioned over the ranks,
// and ONE exchange step per result — an all-gather of the fixed-size Jacobian partial sums over RCCL (xGMI), followed by a local
// combine (an elliptic-curve sum is not an RCCL reduction op, so "all-reduce of partial sums" is all-gather + local add).
// The payload is 168 B (G1) / 336 B (G2) / 672 B (a Groth16 proof) per rank, plus one status word: latency-bound, bandwidth irrelevant.
//
// Two transports behind the same entry points:
//   * RCCL: zkt_comm_unique_id on rank 0, the id shipped out of band by the host (as with ncclGetUniqueId), zkt_comm_init on every rank.
//     librccl.so.1 is opened on first use (dlopen), not linked: a single-GPU consumer of libzkt_hip.so has no RCCL dependency, and inside a
//     process that already hosts an RCCL (PyTorch's) the loader hands back that very library instead of a second copy.  world = 1 runs the
//     same ncclCommInitRank / ncclAllGather calls as world = 8 — the wire path is exercised on a one-GPU box — when the library is there;
//     without one (or with ZKT_COMM_LOCAL_WORLD1=1) world = 1 is a local device-to-device copy;
//   * a host callback (zkt_comm_init_callback) for hosts that bring their own exchange (MPI, gloo, a test harness): the partials go through
//     pinned host memory and `fn` all-gathers bytes_per_rank bytes per rank.  Also what rehearses world > 1 on a one-GPU box, where RCCL
//     refuses two ranks on one device.
// Contract of the collective entry points: EVERY rank takes part in the exchange even when its local stage failed — it sends its status
// word, and all ranks return the first non-zero status (by rank) after the gather.  No rank is left waiting inside ncclAllGather.
// There is no CPU compute here: the partial sums and their combination are HIP kernels (zkt_msm.hip).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and prototypes only: the functions are bound by dlsym below
#include <dlfcn.h>
#include <mutex>
#include <atomic>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include "../../include/zkt.h"
#include "zkt_internal.h"

extern int zkt_internal_ready();                 // zkt_api.cpp: ready + hipSetDevice(the library's device)
extern hipStream_t zkt_internal_stream();
extern "C" int zkt_internal_jac_sum(int grp, const uint32_t* dev_partials, size_t count, size_t stride_words, hipStream_t s, void* out);

namespace {
// ---- RCCL, bound at run time ------------------------------------------------------------------------------------------------------
struct Rccl {
  void* so = nullptr; bool tried = false;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  std::mutex mu;
  bool load() {
    std::lock_guard<std::mutex> lk(mu);
    if (tried) return so != nullptr;
    tried = true;
    const char* names[] = {getenv("ZKT_RCCL_LIB"), "librccl.so.1", "librccl.so"};
    for (const char* nm : names) { if (!nm || !*nm) continue; so = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); if (so) break; }
    if (!so) {
      const char* rp = getenv("ROCM_PATH"); char path[512];
      snprintf(path, sizeof(path), "%s/lib/librccl.so.1", rp && *rp ? rp : "/opt/rocm");
      so = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!so) { fprintf(stderr, "[zkt] RCCL not found (%s): use zkt_comm_init_callback or set ZKT_RCCL_LIB\n", dlerror()); return false; }
#define ZKT_BIND(N) N = (decltype(N))dlsym(so, "nccl" #N); if (!N) { fprintf(stderr, "[zkt] librccl lacks nccl" #N "\n"); dlclose(so); so = nullptr; return false; }
    ZKT_BIND(GetUniqueId) ZKT_BIND(CommInitRank) ZKT_BIND(CommDestroy) ZKT_BIND(CommAbort) ZKT_BIND(AllGather) ZKT_BIND(GetErrorString) ZKT_BIND(GetVersion)
#undef ZKT_BIND
    int v = 0;
    if (GetVersion(&v) == ncclSuccess && v / 10000 != NCCL_MAJOR)       // the ABI this file was compiled against (rccl.h)
      fprintf(stderr, "[zkt] warning: librccl major version %d differs from the headers' %d\n", v / 10000, NCCL_MAJOR);
    return true;
  }
} rccl;

struct Comm {
  bool ready = false;
  std::atomic<int> rank{-1}, world{0};               // -1 / 0 while no communicator exists; read without the lock (zkt_comm_rank / zkt_comm_world may be called from the all-gather callback)
  ncclComm_t nccl = nullptr;
  zkt_allgather_fn fn = nullptr; void* fn_ctx = nullptr;
  uint32_t *d_send = nullptr, *d_recv = nullptr;     // one slot and world slots: persistent, no allocation per call
  uint8_t *h_send = nullptr, *h_recv = nullptr;      // pinned staging of the callback transport (and of the status words)
  std::mutex mu;                                     // one exchange at a time (the buffers and the communicator are shared)
};
Comm c;
static_assert(ZKT_COMM_ID_BYTES >= sizeof(ncclUniqueId), "ZKT_COMM_ID_BYTES must hold an ncclUniqueId");
constexpr size_t PAYLOAD_WORDS = ZKT_GROTH16_PARTIAL_WORDS;   // the largest payload: A | B | C partials of one proof
constexpr size_t HDR_WORDS = 2;                               // status word (+ padding: payloads stay 8-byte aligned)
constexpr size_t SLOT_WORDS = PAYLOAD_WORDS + HDR_WORDS;
constexpr uint32_t STATUS_UNSET = 0xffffffffu;                // what a send slot holds between exchanges: a rank that could not even upload its status reads as failed everywhere

#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "[zkt] HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); return ZKT_ERR_DEVICE; } } while (0)
#define NCCLCHK(x) do { ncclResult_t _r = (x); if (_r != ncclSuccess) { fprintf(stderr, "[zkt] RCCL error %s at %s:%d\n", rccl.GetErrorString(_r), __FILE__, __LINE__); return ZKT_ERR_DEVICE; } } while (0)

void release_locked() {
  if (c.nccl) { rccl.CommDestroy(c.nccl); c.nccl = nullptr; }
  if (c.d_send) (void)hipFree(c.d_send); if (c.d_recv) (void)hipFree(c.d_recv);
  if (c.h_send) (void)hipHostFree(c.h_send); if (c.h_recv) (void)hipHostFree(c.h_recv);
  c.ready = false; c.rank = -1; c.world = 0; c.fn = nullptr; c.fn_ctx = nullptr;
  c.d_send = c.d_recv = nullptr; c.h_send = c.h_recv = nullptr;
}
int alloc_buffers(int world) {
  HIPCHK(hipMalloc((void**)&c.d_send, SLOT_WORDS * 4));
  HIPCHK(hipMalloc((void**)&c.d_recv, SLOT_WORDS * 4 * (size_t)world));
  HIPCHK(hipHostMalloc((void**)&c.h_send, SLOT_WORDS * 4, hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&c.h_recv, SLOT_WORDS * 4 * (size_t)world, hipHostMallocDefault));
  HIPCHK(hipMemset(c.d_send, 0xff, HDR_WORDS * 4));
  return ZKT_OK;
}
uint32_t* send_payload() { return c.d_send + HDR_WORDS; }
// Slot layout on the wire: [status u32][pad u32][payload: `words` u32].  d_send of every rank -> d_recv[rank * (words + 2) ..]; complete on the
// host's view when this returns, with every rank's status word in h_recv.  Caller holds c.mu.  `local_rc` is this rank's status.
// A HIP error on this rank BEFORE the exchange does not keep it out of the exchange: the slot's status word then still holds STATUS_UNSET (written
// at the end of the previous exchange), the collective runs, every rank sees a failed rank, and this rank returns its own error afterwards.
int all_gather_slots(size_t words, int local_rc, int* first_bad_rc) {
  hipStream_t s = zkt_internal_stream();
  const size_t slot = words + HDR_WORDS;
  const int world = c.world.load();
  const uint32_t st[2] = {(uint32_t)local_rc, 0u};
  int local_dev_err = ZKT_OK;
  if (hipMemcpyAsync(c.d_send, st, 8, hipMemcpyHostToDevice, s) != hipSuccess) { fprintf(stderr, "[zkt] status upload failed before the exchange: taking part with a failed status\n"); local_dev_err = ZKT_ERR_DEVICE; }
  *first_bad_rc = 0;
  if (c.nccl || !c.fn) {
    if (c.nccl) NCCLCHK(rccl.AllGather(c.d_send, c.d_recv, slot, ncclUint32, c.nccl, s));
    else HIPCHK(hipMemcpyAsync(c.d_recv, c.d_send, slot * 4, hipMemcpyDeviceToDevice, s));       // world = 1 without an RCCL library: the exchange is a local copy
    for (int r = 0; r < world; ++r) HIPCHK(hipMemcpyAsync(c.h_recv + (size_t)r * 8, c.d_recv + (size_t)r * slot, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemsetAsync(c.d_send, 0xff, HDR_WORDS * 4, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int r = 0; r < world && !*first_bad_rc; ++r) *first_bad_rc = (int)((const uint32_t*)c.h_recv)[2 * r];
    if (*first_bad_rc == (int)STATUS_UNSET) *first_bad_rc = ZKT_ERR_DEVICE;
    return local_dev_err;
  }
  if (hipMemcpyAsync(c.h_send, c.d_send, slot * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
    local_dev_err = ZKT_ERR_DEVICE; ((uint32_t*)c.h_send)[0] = (uint32_t)ZKT_ERR_DEVICE;            // the callback still runs: the other ranks are waiting in theirs
  }
  if (local_dev_err) ((uint32_t*)c.h_send)[0] = (uint32_t)ZKT_ERR_DEVICE;
  if (c.fn(c.fn_ctx, c.h_send, c.h_recv, slot * 4) != 0) { fprintf(stderr, "[zkt] all-gather callback failed\n"); return ZKT_ERR_DEVICE; }
  (void)hipMemsetAsync(c.d_send, 0xff, HDR_WORDS * 4, s);
  for (int r = 0; r < world && !*first_bad_rc; ++r) *first_bad_rc = (int)((const uint32_t*)c.h_recv)[(size_t)r * slot];
  if (*first_bad_rc == (int)STATUS_UNSET) *first_bad_rc = ZKT_ERR_DEVICE;
  if (*first_bad_rc || local_dev_err) return local_dev_err;
  HIPCHK(hipMemcpyAsync(c.d_recv, c.h_recv, slot * 4 * (size_t)world, hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKT_OK;
}
int partial_words(int grp) { return grp == zkt::G_G1 ? ZKT_G1_PARTIAL_WORDS : grp == zkt::G_G2 ? ZKT_G2_PARTIAL_WORDS : ZKT_SECP_PARTIAL_WORDS; }

// exchange + combine of one group's partial already sitting in the send slot (if local_rc == 0): every rank leaves with the same affine point or the same error
int exchange_and_sum(int grp, int local_rc, void* out) {
  const size_t w = (size_t)partial_words(grp);
  int bad = 0;
  int rc = all_gather_slots(w, local_rc, &bad); if (rc) return rc;
  if (bad) return local_rc ? local_rc : bad;
  return zkt_internal_jac_sum(grp, c.d_recv + HDR_WORDS, (size_t)c.world.load(), w + HDR_WORDS, zkt_internal_stream(), out);
}
}  // namespace

extern "C" {

int zkt_comm_unique_id(uint8_t id[ZKT_COMM_ID_BYTES]) {
  if (!id) return ZKT_ERR_SHAPE;
  if (!rccl.load()) return ZKT_ERR_DEVICE;
  ncclUniqueId u;
  NCCLCHK(rccl.GetUniqueId(&u));
  memset(id, 0, ZKT_COMM_ID_BYTES); memcpy(id, &u, sizeof(u));
  return ZKT_OK;
}
int zkt_comm_init(int rank, int world, const uint8_t id[ZKT_COMM_ID_BYTES]) {
  if (zkt_internal_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (world < 1 || rank < 0 || rank >= world || (world > 1 && !id)) return ZKT_ERR_SHAPE;
  // world = 1 on a machine without an RCCL library (or with ZKT_COMM_LOCAL_WORLD1=1): no communicator, the exchange is a device-to-device copy —
  // a single-GPU consumer has no RCCL dependency.  With the library present a one-rank communicator runs the same calls as an 8-rank one.
  const char* lw = getenv("ZKT_COMM_LOCAL_WORLD1");
  const bool have_rccl = !(world == 1 && lw && *lw == '1') && rccl.load();
  if (!have_rccl && world > 1) return ZKT_ERR_DEVICE;
  std::lock_guard<std::mutex> lk(c.mu);
  if (c.ready) return ZKT_ERR_SHAPE;
  c.fn = nullptr; c.fn_ctx = nullptr; c.nccl = nullptr;
  if (!have_rccl) {
    int rc0 = alloc_buffers(world);
    if (rc0) { release_locked(); return rc0; }
    c.rank = rank; c.world = world; c.ready = true;
    return ZKT_OK;
  }
  ncclUniqueId u;
  if (id) memcpy(&u, id, sizeof(u));
  else { ncclResult_t r = rccl.GetUniqueId(&u); if (r != ncclSuccess) { fprintf(stderr, "[zkt] RCCL error %s in ncclGetUniqueId\n", rccl.GetErrorString(r)); release_locked(); return ZKT_ERR_DEVICE; } }
  // world = 1 too: the one-rank communicator runs the same init and collective code as an 8-rank one
  ncclResult_t r = rccl.CommInitRank(&c.nccl, world, u, rank);          // binds the calling thread's current device = the zkt_init device
  if (r != ncclSuccess) { fprintf(stderr, "[zkt] RCCL error %s in ncclCommInitRank\n", rccl.GetErrorString(r)); c.nccl = nullptr; release_locked(); return ZKT_ERR_DEVICE; }
  int rc = alloc_buffers(world);
  if (rc) { release_locked(); return rc; }
  c.rank = rank; c.world = world; c.ready = true;
  return ZKT_OK;
}
int zkt_comm_init_callback(int rank, int world, zkt_allgather_fn fn, void* ctx) {
  if (zkt_internal_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (world < 1 || rank < 0 || rank >= world || !fn) return ZKT_ERR_SHAPE;
  std::lock_guard<std::mutex> lk(c.mu);
  if (c.ready) return ZKT_ERR_SHAPE;
  c.fn = fn; c.fn_ctx = ctx; c.nccl = nullptr;
  int rc = alloc_buffers(world);
  if (rc) { release_locked(); return rc; }
  c.rank = rank; c.world = world; c.ready = true;
  return ZKT_OK;
}
void zkt_comm_finalize(void) {
  std::lock_guard<std::mutex> lk(c.mu);
  if (!c.ready) return;
  release_locked();
}
// lock-free: an all-gather callback runs with the communicator's mutex held by the collective entry point and may ask for its rank / world
int zkt_comm_rank(void) { return c.rank.load(); }
int zkt_comm_world(void) { return c.world.load(); }
void zkt_comm_shard_range(size_t n, int rank, int world, size_t* lo, size_t* hi) {
  if (world < 1 || rank < 0 || rank >= world) { if (lo) *lo = 0; if (hi) *hi = 0; return; }      // no such shard: the empty range
  const size_t base = n / (size_t)world, extra = n % (size_t)world, r = (size_t)rank;
  const size_t b = r * base + (r < extra ? r : extra);
  if (lo) *lo = b;
  if (hi) *hi = b + base + (r < extra ? 1 : 0);
}

// Polynomial::eval_with_g1_hidings (polynomial.rs:271-281) over an index range per rank: this rank's resident shard, its scalars, one exchange.
// A rank whose local stage fails still takes part in the exchange (status word): every rank returns the error, none blocks.
#define ZKT_SHARDED_API(NAME, GRP, PT)                                                                                              \
  int zkt_##NAME##_msm_sharded_collect(zkt_##NAME##_bases* b, int slot, PT* out) {                                                  \
    if (zkt_internal_ready() != ZKT_OK) return ZKT_ERR_DEVICE;                                                                       \
    std::lock_guard<std::mutex> lk(c.mu);                                                                                            \
    if (!c.ready) return ZKT_ERR_SHAPE;                                                                                              \
    int rc = out ? zkt_##NAME##_msm_collect(b, slot, nullptr, send_payload()) : ZKT_ERR_SHAPE;                                             \
    return exchange_and_sum(GRP, rc, out);                                                                                           \
  }                                                                                                                                  \
  int zkt_##NAME##_msm_sharded(zkt_##NAME##_bases* b, const uint64_t* dev_scalars, size_t n_local, void* stream, PT* out) {          \
    if (zkt_internal_ready() != ZKT_OK) return ZKT_ERR_DEVICE;                                                                       \
    std::lock_guard<std::mutex> lk(c.mu);                                                                                            \
    if (!c.ready) return ZKT_ERR_SHAPE;                                                                                              \
    int rc = out ? zkt_##NAME##_msm_dev(b, dev_scalars, n_local, stream, nullptr, send_payload()) : ZKT_ERR_SHAPE;                         \
    return exchange_and_sum(GRP, rc, out);                                                                                           \
  }
ZKT_SHARDED_API(g1, zkt::G_G1, zkt_g1_affine)
ZKT_SHARDED_API(g2, zkt::G_G2, zkt_g2_affine)
ZKT_SHARDED_API(secp, zkt::G_SECP, zkt_secp_affine)

// Prover::prove (prover.rs:96-147) for one proof sharded over the ranks (BASELINE config 4): the key came from
// zkt_groth16_setup_r1cs_sharded(…, zkt_comm_rank(), zkt_comm_world(), …); one all-gather of 672 B (+ status) per proof, three local combines.
int zkt_groth16_prove_r1cs_sharded(zkt_groth16_pk* pk, const uint64_t* dev_wires, const uint64_t* r, const uint64_t* s, zkt_g1_affine* A, zkt_g2_affine* B, zkt_g1_affine* C) {
  if (zkt_internal_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  std::lock_guard<std::mutex> lk(c.mu);
  if (!c.ready) return ZKT_ERR_SHAPE;
  int local = (A && B && C) ? zkt_groth16_prove_r1cs_partials(pk, dev_wires, r, s, send_payload()) : ZKT_ERR_SHAPE;
  int bad = 0, rc;
  if ((rc = all_gather_slots(PAYLOAD_WORDS, local, &bad))) return rc;
  if (bad) return local ? local : bad;
  hipStream_t st = zkt_internal_stream();
  const size_t W = (size_t)c.world.load(); const uint32_t* pay = c.d_recv + HDR_WORDS;
  if ((rc = zkt_internal_jac_sum(zkt::G_G1, pay, W, SLOT_WORDS, st, A))) return rc;
  if ((rc = zkt_internal_jac_sum(zkt::G_G2, pay + ZKT_G1_PARTIAL_WORDS, W, SLOT_WORDS, st, B))) return rc;
  return zkt_internal_jac_sum(zkt::G_G1, pay + ZKT_G1_PARTIAL_WORDS + ZKT_G2_PARTIAL_WORDS, W, SLOT_WORDS, st, C);
}

}  // extern "C"
