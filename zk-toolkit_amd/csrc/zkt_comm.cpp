// Multi-GPU inside the C ABI: one process per GPU, the MSM / proof index range partitioned over the ranks,
// and ONE exchange step per result — an all-gather of the fixed-size Jacobian partial sums over RCCL (xGMI), followed by a local
// combine (an elliptic-curve sum is not an RCCL reduction op, so "all-reduce of partial sums" is all-gather + local add).
// The payload is 168 B (G1) / 336 B (G2) / 672 B (a Groth16 proof) per rank: latency-bound, bandwidth irrelevant.
//
// Two transports behind the same entry points:
//   * RCCL: zkt_comm_unique_id on rank 0, the id shipped out of band by the host (as with ncclGetUniqueId), zkt_comm_init on every rank;
//   * a host callback (zkt_comm_init_callback) for hosts that bring their own exchange (MPI, gloo, a test harness): the partials go through
//     pinned host memory and `fn` all-gathers bytes_per_rank bytes per rank.  Also what rehearses world > 1 on a one-GPU box, where RCCL
//     refuses two ranks on one device.
// There is no CPU compute here: the partial sums and their combination are HIP kernels (zkt_msm.hip).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <mutex>
#include <cstring>
#include <cstdio>
#include "../../include/zkt.h"
#include "zkt_internal.h"

extern int zkt_internal_ready();                 // zkt_api.cpp: ready + hipSetDevice(the library's device)
extern hipStream_t zkt_internal_stream();
extern "C" int zkt_internal_jac_sum(int grp, const uint32_t* dev_partials, size_t count, size_t stride_words, hipStream_t s, void* out);

namespace {
struct Comm {
  bool ready = false;
  int rank = 0, world = 1;
  ncclComm_t nccl = nullptr;
  zkt_allgather_fn fn = nullptr; void* fn_ctx = nullptr;
  uint32_t *d_send = nullptr, *d_recv = nullptr;     // ZKT_GROTH16_PARTIAL_WORDS and world * that: persistent, no allocation per call
  uint8_t *h_send = nullptr, *h_recv = nullptr;      // pinned staging of the callback transport
  std::mutex mu;                                     // one exchange at a time (the buffers and the communicator are shared)
};
Comm c;
static_assert(ZKT_COMM_ID_BYTES >= sizeof(ncclUniqueId), "ZKT_COMM_ID_BYTES must hold an ncclUniqueId");
constexpr size_t SLOT_WORDS = ZKT_GROTH16_PARTIAL_WORDS;   // the largest payload: A | B | C partials of one proof

#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "[zkt] HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); return ZKT_ERR_DEVICE; } } while (0)
#define NCCLCHK(x) do { ncclResult_t _r = (x); if (_r != ncclSuccess) { fprintf(stderr, "[zkt] RCCL error %s at %s:%d\n", ncclGetErrorString(_r), __FILE__, __LINE__); return ZKT_ERR_DEVICE; } } while (0)

int alloc_buffers() {
  HIPCHK(hipMalloc((void**)&c.d_send, SLOT_WORDS * 4));
  HIPCHK(hipMalloc((void**)&c.d_recv, SLOT_WORDS * 4 * (size_t)c.world));
  if (c.fn) {
    HIPCHK(hipHostMalloc((void**)&c.h_send, SLOT_WORDS * 4, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&c.h_recv, SLOT_WORDS * 4 * (size_t)c.world, hipHostMallocDefault));
  }
  return ZKT_OK;
}
// d_send[0..words) of every rank -> d_recv[rank * words ..], complete on the host's view when this returns.  Caller holds c.mu.
int all_gather_words(size_t words) {
  hipStream_t s = zkt_internal_stream();
  if (c.world == 1) { HIPCHK(hipMemcpyAsync(c.d_recv, c.d_send, words * 4, hipMemcpyDeviceToDevice, s)); HIPCHK(hipStreamSynchronize(s)); return ZKT_OK; }
  if (c.nccl) {
    NCCLCHK(ncclAllGather(c.d_send, c.d_recv, words, ncclUint32, c.nccl, s));
    HIPCHK(hipStreamSynchronize(s));
    return ZKT_OK;
  }
  HIPCHK(hipMemcpyAsync(c.h_send, c.d_send, words * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (c.fn(c.fn_ctx, c.h_send, c.h_recv, words * 4) != 0) { fprintf(stderr, "[zkt] all-gather callback failed\n"); return ZKT_ERR_DEVICE; }
  HIPCHK(hipMemcpyAsync(c.d_recv, c.h_recv, words * 4 * (size_t)c.world, hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKT_OK;
}
int partial_words(int grp) { return grp == zkt::G_G1 ? ZKT_G1_PARTIAL_WORDS : grp == zkt::G_G2 ? ZKT_G2_PARTIAL_WORDS : ZKT_SECP_PARTIAL_WORDS; }

// exchange + combine of one group's partial already sitting in d_send: every rank leaves with the same affine point
int exchange_and_sum(int grp, void* out) {
  const size_t w = (size_t)partial_words(grp);
  int rc = all_gather_words(w); if (rc) return rc;
  return zkt_internal_jac_sum(grp, c.d_recv, (size_t)c.world, w, zkt_internal_stream(), out);
}
}  // namespace

extern "C" {

int zkt_comm_unique_id(uint8_t id[ZKT_COMM_ID_BYTES]) {
  if (!id) return ZKT_ERR_SHAPE;
  ncclUniqueId u;
  NCCLCHK(ncclGetUniqueId(&u));
  memset(id, 0, ZKT_COMM_ID_BYTES); memcpy(id, &u, sizeof(u));
  return ZKT_OK;
}
int zkt_comm_init(int rank, int world, const uint8_t id[ZKT_COMM_ID_BYTES]) {
  if (zkt_internal_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (world < 1 || rank < 0 || rank >= world || (world > 1 && !id)) return ZKT_ERR_SHAPE;
  std::lock_guard<std::mutex> lk(c.mu);
  if (c.ready) return ZKT_ERR_SHAPE;
  c.rank = rank; c.world = world; c.fn = nullptr;
  if (world > 1) {
    ncclUniqueId u; memcpy(&u, id, sizeof(u));
    NCCLCHK(ncclCommInitRank(&c.nccl, world, u, rank));          // binds the calling thread's current device = the zkt_init device
  }
  int rc = alloc_buffers(); if (rc) return rc;
  c.ready = true;
  return ZKT_OK;
}
int zkt_comm_init_callback(int rank, int world, zkt_allgather_fn fn, void* ctx) {
  if (zkt_internal_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (world < 1 || rank < 0 || rank >= world || !fn) return ZKT_ERR_SHAPE;
  std::lock_guard<std::mutex> lk(c.mu);
  if (c.ready) return ZKT_ERR_SHAPE;
  c.rank = rank; c.world = world; c.fn = fn; c.fn_ctx = ctx; c.nccl = nullptr;
  int rc = alloc_buffers(); if (rc) return rc;
  c.ready = true;
  return ZKT_OK;
}
void zkt_comm_finalize(void) {
  std::lock_guard<std::mutex> lk(c.mu);
  if (!c.ready) return;
  if (c.nccl) ncclCommDestroy(c.nccl);
  if (c.d_send) hipFree(c.d_send); if (c.d_recv) hipFree(c.d_recv);
  if (c.h_send) hipHostFree(c.h_send); if (c.h_recv) hipHostFree(c.h_recv);
  c.ready = false; c.rank = 0; c.world = 1; c.nccl = nullptr; c.fn = nullptr; c.fn_ctx = nullptr;
  c.d_send = c.d_recv = nullptr; c.h_send = c.h_recv = nullptr;
}
int zkt_comm_rank(void) { return c.ready ? c.rank : -1; }
int zkt_comm_world(void) { return c.ready ? c.world : 0; }
void zkt_comm_shard_range(size_t n, int rank, int world, size_t* lo, size_t* hi) {
  const size_t base = n / (size_t)world, extra = n % (size_t)world, r = (size_t)rank;
  const size_t b = r * base + (r < extra ? r : extra);
  if (lo) *lo = b;
  if (hi) *hi = b + base + (r < extra ? 1 : 0);
}

// Polynomial::eval_with_g1_hidings (polynomial.rs:271-281) over an index range per rank: this rank's resident shard, its scalars, one exchange
#define ZKT_SHARDED_API(NAME, GRP, PT)                                                                                              \
  int zkt_##NAME##_msm_sharded_collect(zkt_##NAME##_bases* b, int slot, PT* out) {                                                  \
    if (zkt_internal_ready() != ZKT_OK) return ZKT_ERR_DEVICE;                                                                       \
    if (!c.ready || !out) return ZKT_ERR_SHAPE;                                                                                      \
    std::lock_guard<std::mutex> lk(c.mu);                                                                                            \
    int rc = zkt_##NAME##_msm_collect(b, slot, nullptr, c.d_send); if (rc) return rc;                                                \
    return exchange_and_sum(GRP, out);                                                                                               \
  }                                                                                                                                  \
  int zkt_##NAME##_msm_sharded(zkt_##NAME##_bases* b, const uint64_t* dev_scalars, size_t n_local, void* stream, PT* out) {          \
    if (zkt_internal_ready() != ZKT_OK) return ZKT_ERR_DEVICE;                                                                       \
    if (!c.ready || !out) return ZKT_ERR_SHAPE;                                                                                      \
    std::lock_guard<std::mutex> lk(c.mu);                                                                                            \
    int rc = zkt_##NAME##_msm_dev(b, dev_scalars, n_local, stream, nullptr, c.d_send); if (rc) return rc;                            \
    return exchange_and_sum(GRP, out);                                                                                               \
  }
ZKT_SHARDED_API(g1, zkt::G_G1, zkt_g1_affine)
ZKT_SHARDED_API(g2, zkt::G_G2, zkt_g2_affine)
ZKT_SHARDED_API(secp, zkt::G_SECP, zkt_secp_affine)

// Prover::prove (prover.rs:96-147) for one proof sharded over the ranks (BASELINE config 4): the key came from
// zkt_groth16_setup_r1cs_sharded(…, zkt_comm_rank(), zkt_comm_world(), …); one all-gather of 672 B per proof, three local combines.
int zkt_groth16_prove_r1cs_sharded(zkt_groth16_pk* pk, const uint64_t* dev_wires, const uint64_t* r, const uint64_t* s, zkt_g1_affine* A, zkt_g2_affine* B, zkt_g1_affine* C) {
  if (zkt_internal_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (!c.ready || !A || !B || !C) return ZKT_ERR_SHAPE;
  std::lock_guard<std::mutex> lk(c.mu);
  int rc = zkt_groth16_prove_r1cs_partials(pk, dev_wires, r, s, c.d_send); if (rc) return rc;
  if ((rc = all_gather_words(SLOT_WORDS))) return rc;
  hipStream_t st = zkt_internal_stream();
  if ((rc = zkt_internal_jac_sum(zkt::G_G1, c.d_recv, (size_t)c.world, SLOT_WORDS, st, A))) return rc;
  if ((rc = zkt_internal_jac_sum(zkt::G_G2, c.d_recv + ZKT_G1_PARTIAL_WORDS, (size_t)c.world, SLOT_WORDS, st, B))) return rc;
  return zkt_internal_jac_sum(zkt::G_G1, c.d_recv + ZKT_G1_PARTIAL_WORDS + ZKT_G2_PARTIAL_WORDS, (size_t)c.world, SLOT_WORDS, st, C);
}

}  // extern "C"
