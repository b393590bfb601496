// Pinocchio (protocol 2 of eprint 2013/279) over the batched kernels — SURVEY §8 row f-4:
//   src/zk/w_trusted_setup/pinocchio/{crs.rs:49-161, prover.rs:98-170, verifier.rs:31-85}
// Random values the reference draws from OS entropy (crs.rs:58-64,82; prover.rs:104-105) are arguments.
// Setup is one Fr scalar stage on the device followed by fixed-base multiplications of the two generators (every CRS element
// is generator * scalar, whatever chain of multiplications the reference writes); prove is 7 G1 MSMs + 2 G2 MSMs over the mid
// wires plus a handful of single operations; verify is five pairing-product equalities, the first four in one launch.
#include <vector>
#include <memory>
#include <mutex>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include "abi.h"
#include "zkt_internal.h"
#include "../../include/zkt.h"

namespace zkt {
typedef FrC C;
// out[i] = P_i(x) for `rows` dense polynomials of n coefficients, Montgomery (Polynomial::eval_at, polynomial.rs:240-249)
__global__ void __launch_bounds__(64) k_pin_poly_eval(const uint32_t* __restrict__ P, size_t rows, size_t n, const uint32_t* __restrict__ x, uint32_t* __restrict__ out_mont) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= rows) return;
  Fp<C> xm = ld_fp<C>(x), acc = fp_zero<C>();
  for (size_t k = n; k-- > 0;) acc = fp_add(fp_mul(acc, xm), ld_fp<C>(P + (i * n + k) * 8));
  st_raw<C>(out_mont + i * 8, acc);
}
// per wire i: the seven scalars of crs.rs:86-108,122-124 (canonical); rnd = r_v, r_w, alpha_v, alpha_w, alpha_y, beta, gamma, s
//   col 0 r_v v_i   1 r_w w_i   2 r_y y_i   3 r_v alpha_v v_i   4 r_w alpha_w w_i   5 r_y alpha_y y_i   6 beta (r_v v_i + r_w w_i + r_y y_i)
// thread 0 also writes the single scalars of crs.rs:110-140: 1, alpha_v, alpha_w, alpha_y, gamma, gamma beta, T = r_y t(s), T alpha_v, T alpha_y, T beta
__global__ void __launch_bounds__(64) k_pin_scalars(const uint32_t* __restrict__ ve, const uint32_t* __restrict__ we, const uint32_t* __restrict__ ye,
                                                    const uint32_t* __restrict__ rnd, size_t n, size_t rows, uint32_t* __restrict__ cols, uint32_t* __restrict__ singles) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  Fp<C> r_v = ld_fp<C>(rnd), r_w = ld_fp<C>(rnd + 8), a_v = ld_fp<C>(rnd + 16), a_w = ld_fp<C>(rnd + 24), a_y = ld_fp<C>(rnd + 32), beta = ld_fp<C>(rnd + 40);
  Fp<C> r_y = fp_mul(r_v, r_w);                                            // crs.rs:67
  if (i < rows) {
    Fp<C> v = fp_mul(r_v, ld_raw<C>(ve + i * 8)), w = fp_mul(r_w, ld_raw<C>(we + i * 8)), y = fp_mul(r_y, ld_raw<C>(ye + i * 8));
    st_fp<C>(cols + (0 * rows + i) * 8, v); st_fp<C>(cols + (1 * rows + i) * 8, w); st_fp<C>(cols + (2 * rows + i) * 8, y);
    st_fp<C>(cols + (3 * rows + i) * 8, fp_mul(a_v, v)); st_fp<C>(cols + (4 * rows + i) * 8, fp_mul(a_w, w)); st_fp<C>(cols + (5 * rows + i) * 8, fp_mul(a_y, y));
    st_fp<C>(cols + (6 * rows + i) * 8, fp_mul(beta, fp_add(fp_add(v, w), y)));
  }
  if (i == 0) {
    Fp<C> gamma = ld_fp<C>(rnd + 48), s = ld_fp<C>(rnd + 56), one = fp_one<C>(), t = one, ii = fp_zero<C>();
    for (size_t k = 1; k <= n; ++k) { ii = fp_add(ii, one); t = fp_mul(t, fp_sub(s, ii)); }     // QAP::build_t(f,n).eval_at(s), qap.rs:115-135
    Fp<C> T = fp_mul(r_y, t);
    st_fp<C>(singles, one); st_fp<C>(singles + 8, a_v); st_fp<C>(singles + 16, a_w); st_fp<C>(singles + 24, a_y); st_fp<C>(singles + 32, gamma);
    st_fp<C>(singles + 40, fp_mul(gamma, beta)); st_fp<C>(singles + 48, T); st_fp<C>(singles + 56, fp_mul(T, a_v)); st_fp<C>(singles + 64, fp_mul(T, a_y));
    st_fp<C>(singles + 72, fp_mul(T, beta));
  }
}
// out[k] = s^k, canonical (pow_seq, prime_field_elem.rs:346-361)
__global__ void __launch_bounds__(256) k_pin_powseq(const uint32_t* __restrict__ base, size_t n, uint32_t* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  Fp<C> b = ld_fp<C>(base), r = fp_one<C>();
  for (size_t e = i; e; e >>= 1) { if (e & 1) r = fp_mul(r, b); b = fp_sqr(b); }
  st_fp<C>(out + i * 8, r);
}
// Verifier: the handful of point additions between the statement products and the pairings, one lane per output and one normalisation each
// (out = fix[0] + .. + fix[nfix-1] + terms[0] + .. + terms[nterms-1], complete additions; a launch per addition costs ~0.3 ms of latency each).
struct PinSum { const uint32_t* fix[3]; int nfix; const uint32_t* terms; int nterms; uint32_t* out; };
struct PinSums { PinSum s[3]; int n; };
template <class F>
__global__ void __launch_bounds__(64) k_pin_sums(PinSums p) {
  const int j = threadIdx.x;
  if (j >= p.n) return;
  constexpr int W = PtIO<F>::WORDS;
  const PinSum& q = p.s[j];
  Jac<F> acc = jac_inf<F>();
  for (int k = 0; k < q.nfix; ++k) acc = jac_add_aff(acc, PtIO<F>::ld(q.fix[k]));
  for (int t = 0; t < q.nterms; ++t) acc = jac_add_aff(acc, PtIO<F>::ld(q.terms + (size_t)t * W));
  PtIO<F>::st(q.out, jac_to_aff(acc));
}
}  // namespace zkt

using namespace zkt;

namespace {
struct Dev {
  void* p = nullptr;
  explicit Dev(size_t bytes) { if (hipMalloc(&p, bytes ? bytes : 4) != hipSuccess) p = nullptr; }
  ~Dev() { if (p) hipFree(p); }
  uint32_t* w() const { return (uint32_t*)p; }
  Dev(const Dev&) = delete; Dev& operator=(const Dev&) = delete;
};
#define QCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "[zkt] HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); return ZKT_ERR_DEVICE; } } while (0)
#define ZRC(x) do { int _rc = (x); if (_rc != ZKT_OK) return _rc; } while (0)
const size_t G1B = sizeof(zkt_g1_affine), G2B = sizeof(zkt_g2_affine), FRB = 32;
const uint64_t G1_GEN[13] = {0xfb3af00adb22c6bbull, 0x6c55e83ff97a1aefull, 0xa14e3a3f171bac58ull, 0xc3688c4f9774b905ull, 0x2695638c4fa9ac0full, 0x17f1d3a73197d794ull,
                             0x0caa232946c5e7e1ull, 0xd03cc744a2888ae4ull, 0x00db18cb2c04b3edull, 0xfcf5e095d5d00af6ull, 0xa09e30ed741d8ae4ull, 0x08b3f481e3aaa0f1ull, 0};   // g1_point.rs:38-47
const uint64_t G2_GEN[25] = {0xe5ac7d055d042b7eull, 0x334cf11213945d57ull, 0xb5da61bbdc7f5049ull, 0x596bd0d09920b61aull, 0x7dacd3a088274f65ull, 0x13e02b6052719f60ull,
                             0xd48056c8c121bdb8ull, 0x0bac0326a805bbefull, 0xb4510b647ae3d177ull, 0xc6e47ad4fa403b02ull, 0x260805272dc51051ull, 0x024aa2b2f08f0a91ull,
                             0xaaa9075ff05f79beull, 0x3f370d275cec1da1ull, 0x267492ab572e99abull, 0xcb3e287e85a763afull, 0x32acd2b02bc28b99ull, 0x0606c4a02ea734ccull,
                             0xe193548608b82801ull, 0x923ac9cc3baca289ull, 0x6d429a695160d12cull, 0xadfd9baa8cbdd3a7ull, 0x8cc9cdc6da2e351aull, 0x0ce5d527727d6e11ull, 0};   // g2_point.rs:36-46
}  // namespace

extern int zkt_internal_ready();   // zkt_api.cpp
extern void zkt_internal_set_error_index(size_t i);

// ---- verification with the key's io points as fixed-base tables ------------------------------------------------------------------
// A verifier checks many proofs against ONE key.  The statement sums sum_i io_i * {vk_io, yk_io, wk_io}[i] (verifier.rs:70-76) through the one-shot
// MSM entry points cost 2.6 + 2.8 + 7.1 ms for a handful of wires — a 255-step doubling chain each, however few points.  Here the key's io points get
// fixed-base tables once (launch_fixed_tables, kept for the last two keys, keyed by the points' bytes), a statement sum is one wave per wire and the
// additions are one lane each, and the five equalities are ONE launch of the lane-distributed product kernel (K = 3 with a pair count per element):
// ~25 ms -> ~6.5 ms per verification, same decisions in the same order.  Everything runs on the stream the other protocol calls use: every queue that runs
// a pairing kernel keeps 5-6 GiB of scratch (see GuardStreams, zkt_pairing.hip), and two verifier streams of their own exhausted the process's scratch pool.
// An element that does not fit the short Miller loop (a point off its curve or outside G2) sends the whole verification to the table-free path below.
namespace {
constexpr size_t PIN_FAST_IO = 12;                  // launch_fixed_tables takes twelve points per launch
struct PinTables { std::vector<uint8_t> key; std::shared_ptr<void> mem; uint64_t stamp = 0; };
struct PinState {
  std::mutex mu;                                    // the table cache; a verification holds it from look-up to completion, so an entry is never released under a running launch
  PinTables tab[2]; uint64_t clock = 0;
} g_pin;
// tables of [vk_io | yk_io] (G1, 2 n_io points) followed by wk_io (G2, n_io points); dio1 / dio2 = the same points already in HBM.  Caller holds g_pin.mu.
std::shared_ptr<void> pin_tables_for(const zkt_pinocchio_crs* c, const uint32_t* dio1, const uint32_t* dio2, hipStream_t s) {
  const size_t nio = c->n_io, k1 = nio * G1B, k2 = nio * G2B;
  PinTables* victim = &g_pin.tab[0];
  for (PinTables& e : g_pin.tab) {
    if (e.mem && e.key.size() == 2 * k1 + k2 && memcmp(e.key.data(), c->vk_io, k1) == 0 && memcmp(e.key.data() + k1, c->yk_io, k1) == 0 &&
        memcmp(e.key.data() + 2 * k1, c->wk_io, k2) == 0) { e.stamp = ++g_pin.clock; return e.mem; }
    if (e.stamp < victim->stamp) victim = &e;
  }
  void* mem = nullptr;
  if (hipMalloc(&mem, 64 * (2 * k1 + k2)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  std::shared_ptr<void> t(mem, [](void* q) { if (q) hipFree(q); });
  uint32_t* t1 = (uint32_t*)mem; uint32_t* t2 = t1 + 2 * nio * 64 * (G1B / 4);
  for (int half = 0; half < 2; ++half) {
    FixedTables ft{}; ft.n = (int)nio;
    for (size_t j = 0; j < nio; ++j) { ft.point[j] = dio1 + (half * nio + j) * (G1B / 4); ft.table[j] = t1 + (half * nio + j) * 64 * (G1B / 4); }
    if (launch_fixed_tables(G_G1, ft, s) != hipSuccess) return nullptr;
  }
  FixedTables ft{}; ft.n = (int)nio;
  for (size_t j = 0; j < nio; ++j) { ft.point[j] = dio2 + j * (G2B / 4); ft.table[j] = t2 + j * 64 * (G2B / 4); }
  if (launch_fixed_tables(G_G2, ft, s) != hipSuccess) return nullptr;
  victim->mem = t; victim->stamp = ++g_pin.clock;
  victim->key.resize(2 * k1 + k2);
  memcpy(victim->key.data(), c->vk_io, k1); memcpy(victim->key.data() + k1, c->yk_io, k1); memcpy(victim->key.data() + 2 * k1, c->wk_io, k2);
  return t;
}
// The decision of Verifier::verify from the outcomes of its five checks, in the reference's order (verifier.rs:43-84): a rejection by an earlier check
// wins over a panic (argument at infinity) of a later one.  rc4 / inf4: status and first-infinity index of the four 2-pair checks; rc1: the 3-pair check.
int pin_decide(int rc4, size_t inf4, const uint32_t* ok4, int rc1, uint32_t ok1) {
  if (rc4 != ZKT_OK && rc4 != ZKT_ERR_INFINITY) return -rc4;
  if (rc4 != ZKT_ERR_INFINITY) inf4 = (size_t)-1;
  for (size_t k = 0; k < 4; ++k) {
    if (k == inf4) return -ZKT_ERR_INFINITY;        // the first check with an argument at infinity: the reference panics here (later lanes may also be at infinity)
    if (!ok4[k]) return 0;
  }
  if (rc1 != ZKT_OK) return -rc1;
  return ok1 ? 1 : 0;
}
constexpr int PIN_FALL_BACK = -1000000;             // pin_verify_fast: an element does not fit the short Miller loop (a point off its curve or outside G2) — evaluate the reference way
int pin_verify_fast(const zkt_pinocchio_crs* c, const zkt_pinocchio_proof* pf, const uint64_t* io_wires) {
  std::lock_guard<std::mutex> lk(g_pin.mu);
  hipStream_t s = nullptr;                          // the stream every protocol call uses (scratch budget: DESIGN.md §4)
  const size_t nio = c->n_io;
  const int W1 = (int)(G1B / 4), W2 = (int)(G2B / 4);
  // Five equalities lhs == rhs as products == 1 (verifier.rs:43-84), packed as 5 elements x 3 pair slots for ONE launch: the four 2-pair checks repeat
  // their pair 0 in slot 2 (kcount = 2), the divisibility check uses all three.  Slots marked * are filled on the device.
  //   0: (beta_vwy_mid_s, gamma)   (*v_mid_s + w_mid_s + y_mid_s, beta_gamma)        :43-48
  //   1: (alpha_v_mid_s, one_g2)   (v_mid_s, alpha_v)                                :50-55
  //   2: (alpha_w_mid_s, one_g2)   (alpha_w, g2_w_mid_s)                             :56-61
  //   3: (alpha_y_mid_s, one_g2)   (y_mid_s, alpha_y)                                :62-66
  //   4: (*v_s, *w_s)              (t, h_s)                 (*y_s, one_g2)           :69-84
  zkt_g1_affine g1p[15] = {*pf->beta_vwy_mid_s, *pf->g1_w_mid_s /* operand of the sum that replaces it */, *pf->beta_vwy_mid_s,
                           *pf->alpha_v_mid_s, *pf->v_mid_s, *pf->alpha_v_mid_s,
                           *pf->alpha_w_mid_s, *c->alpha_w, *pf->alpha_w_mid_s,
                           *pf->alpha_y_mid_s, *pf->y_mid_s, *pf->alpha_y_mid_s,
                           *pf->v_mid_s, *c->t, *pf->y_mid_s};
  zkt_g2_affine g2p[15] = {*c->gamma, *c->beta_gamma, *c->gamma,
                           *c->one_g2, *c->alpha_v, *c->one_g2,
                           *c->one_g2, *pf->g2_w_mid_s, *c->one_g2,
                           *c->one_g2, *c->alpha_y, *c->one_g2,
                           *pf->g2_w_mid_s, *pf->h_s, *c->one_g2};
  const uint8_t kcount[8] = {2, 2, 2, 2, 3, 0, 0, 0};
  Dev d1(sizeof g1p), d2(sizeof g2p), dio1(2 * nio * G1B), dio2(nio * G2B), dk(nio * FRB), dp1(2 * nio * G1B), dp2(nio * G2B), dok(5 * 4), derr(8), dcnt(8);
  if (!d1.p || !d2.p || !dio1.p || !dio2.p || !dk.p || !dp1.p || !dp2.p || !dok.p || !derr.p || !dcnt.p) return -ZKT_ERR_DEVICE;
#define VCHK(x) do { if ((x) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(s); return -ZKT_ERR_DEVICE; } } while (0)
  const unsigned long long noerr = ~0ull;
  VCHK(hipMemcpyAsync(d1.p, g1p, sizeof g1p, hipMemcpyHostToDevice, s)); VCHK(hipMemcpyAsync(d2.p, g2p, sizeof g2p, hipMemcpyHostToDevice, s));
  VCHK(hipMemcpyAsync(derr.p, &noerr, 8, hipMemcpyHostToDevice, s)); VCHK(hipMemcpyAsync(dcnt.p, kcount, 8, hipMemcpyHostToDevice, s));
  std::shared_ptr<void> tabs;
  if (nio) {
    VCHK(hipMemcpyAsync(dio1.p, c->vk_io, nio * G1B, hipMemcpyHostToDevice, s)); VCHK(hipMemcpyAsync(dio1.w() + nio * W1, c->yk_io, nio * G1B, hipMemcpyHostToDevice, s));
    VCHK(hipMemcpyAsync(dio2.p, c->wk_io, nio * G2B, hipMemcpyHostToDevice, s)); VCHK(hipMemcpyAsync(dk.p, io_wires, nio * FRB, hipMemcpyHostToDevice, s));
    tabs = pin_tables_for(c, dio1.w(), dio2.w(), s);
    if (!tabs) { (void)hipStreamSynchronize(s); return -ZKT_ERR_DEVICE; }
    const uint32_t* t1 = (const uint32_t*)tabs.get(); const uint32_t* t2 = t1 + 2 * nio * 64 * W1;
    // products io_i * point_i: out[j] for table j with scalar k[j] (n = 1 "proof", n_pts tables)
    VCHK(launch_fixed_muls_batch(G_G1, t1, dk.w(), dp1.w(), 1, (int)nio, s));
    VCHK(launch_fixed_muls_batch(G_G1, t1 + nio * 64 * W1, dk.w(), dp1.w() + nio * W1, 1, (int)nio, s));
    VCHK(launch_fixed_muls_batch(G_G2, t2, dk.w(), dp2.w(), 1, (int)nio, s));
  }
  {
    uint32_t* G1 = d1.w(); uint32_t* G2 = d2.w();
    PinSums p1{}; p1.n = 3;
    p1.s[0] = PinSum{{G1 + 4 * W1, G1 + 1 * W1, G1 + 10 * W1}, 3, nullptr, 0, G1 + 1 * W1};                    // (v_mid_s + w_mid_s) + y_mid_s (:44), in place of w_mid_s
    p1.s[1] = PinSum{{G1 + 4 * W1, nullptr, nullptr}, 1, dp1.w(), (int)nio, G1 + 12 * W1};                      // v_s = v_mid_s + v_io (:70-72)
    p1.s[2] = PinSum{{G1 + 10 * W1, nullptr, nullptr}, 1, dp1.w() + nio * W1, (int)nio, G1 + 14 * W1};          // y_s = y_mid_s + y_io (:75-76)
    hipLaunchKernelGGL(k_pin_sums<FqOps>, dim3(1), dim3(64), 0, s, p1);
    PinSums p2{}; p2.n = 1;
    p2.s[0] = PinSum{{G2 + 7 * W2, nullptr, nullptr}, 1, dp2.w(), (int)nio, G2 + 12 * W2};                      // w_s = g2_w_mid_s + w_io (:73-74)
    hipLaunchKernelGGL(k_pin_sums<Fq2Ops>, dim3(1), dim3(64), 0, s, p2);
    PairArgs a{};
    for (int j = 0; j < 3; ++j) { a.g1[j] = G1 + j * W1; a.g2[j] = G2 + j * W2; a.s1[j] = 3 * W1; a.s2[j] = 3 * W2; a.neg[j] = j ? 1 : 0; }
    const hipError_t le = launch_pairing_product_check_counts(a, 3, (const uint8_t*)dcnt.p, dok.w(), 5, (unsigned long long*)derr.p, s);
    if (le == hipErrorInvalidValue) { (void)hipGetLastError(); (void)hipStreamSynchronize(s); return PIN_FALL_BACK; }      // small-batch kernels switched off (ZKT_DPRODUCT_MAX)
    VCHK(le);
  }
  uint32_t ok[5] = {0, 0, 0, 0, 0}; unsigned long long e = ~0ull;
  VCHK(hipMemcpyAsync(ok, dok.p, sizeof ok, hipMemcpyDeviceToHost, s)); VCHK(hipMemcpyAsync(&e, derr.p, sizeof e, hipMemcpyDeviceToHost, s));
  VCHK(hipStreamSynchronize(s));
#undef VCHK
  for (uint32_t v : ok) if (v > 1) return PIN_FALL_BACK;
  // one error word for the five elements: the smallest index with an argument at infinity.  If it is among the first four, the decision is taken there or before.
  return pin_decide(e < 4 ? ZKT_ERR_INFINITY : ZKT_OK, (size_t)e, ok, e == 4 ? ZKT_ERR_INFINITY : ZKT_OK, ok[4]);
}
}  // namespace
extern "C" void zkt_pinocchio_clear_caches() {      // zkt_shutdown (through zkt_internal_clear_caches)
  std::lock_guard<std::mutex> lk(g_pin.mu);
  for (PinTables& t : g_pin.tab) { t.mem.reset(); t.key.clear(); t.stamp = 0; }
}

extern "C" {

// CRS::new (crs.rs:49-161)
int zkt_pinocchio_setup(zkt_pinocchio_crs* c, const uint64_t* vi, const uint64_t* wi, const uint64_t* yi, const uint64_t* rnd) {
  if (zkt_internal_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (!c || !vi || !wi || !yi || !rnd || c->n == 0 || c->max_degree == 0) return ZKT_ERR_SHAPE;
  for (int k = 0; k < 8; ++k) { bool z = true; for (int j = 0; j < 4; ++j) z = z && rnd[4 * k + j] == 0; if (z) return ZKT_ERR_INV_ZERO; }   // rand_elem(true) (crs.rs:58-64,82)
  const size_t n = c->n, nio = c->n_io, nmid = c->n_mid, rows = nio + nmid, deg = c->max_degree;
  if (rows == 0) return ZKT_ERR_SHAPE;
  hipStream_t s = nullptr;
  Dev dP(rows * n * FRB), drnd(256), dve(rows * FRB), dwe(rows * FRB), dye(rows * FRB), dcols(7 * rows * FRB), dsing(10 * FRB), dpow(deg * FRB);
  Dev dgen1(G1B), dgen2(G2B), d1(7 * rows * G1B), d2((rows + deg) * G2B), ds1(10 * G1B), ds2(10 * G2B);
  if (!dP.p || !dcols.p || !d1.p || !d2.p || !ds1.p || !ds2.p || !dpow.p) return ZKT_ERR_DEVICE;
  QCHK(hipMemcpyAsync(drnd.p, rnd, 256, hipMemcpyHostToDevice, s));
  QCHK(hipMemcpyAsync(dgen1.p, G1_GEN, G1B, hipMemcpyHostToDevice, s)); QCHK(hipMemcpyAsync(dgen2.p, G2_GEN, G2B, hipMemcpyHostToDevice, s));
  const uint64_t* polys[3] = {vi, wi, yi}; Dev* ev[3] = {&dve, &dwe, &dye};
  for (int k = 0; k < 3; ++k) {
    QCHK(hipMemcpyAsync(dP.p, polys[k], rows * n * FRB, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_pin_poly_eval, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, s, (const uint32_t*)dP.w(), rows, n, (const uint32_t*)(drnd.w() + 56), ev[k]->w());
  }
  hipLaunchKernelGGL(k_pin_scalars, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, s, (const uint32_t*)dve.w(), (const uint32_t*)dwe.w(), (const uint32_t*)dye.w(),
                     (const uint32_t*)drnd.w(), n, rows, dcols.w(), dsing.w());
  hipLaunchKernelGGL(k_pin_powseq, dim3((unsigned)((deg + 255) / 256)), dim3(256), 0, s, (const uint32_t*)(drnd.w() + 56), deg, dpow.w());
  QCHK(hipGetLastError());
  // generator * scalar for every element
  QCHK(launch_generator_mul(G_G1, dgen1.w(), dcols.w(), d1.w(), 7 * rows, s));
  QCHK(launch_generator_mul(G_G2, dgen2.w(), dcols.w() + rows * 8, d2.w(), rows, s));                 // g2_w * w_i(s)
  QCHK(launch_generator_mul(G_G2, dgen2.w(), dpow.w(), d2.w() + rows * 50, deg, s));                   // si
  QCHK(launch_generator_mul(G_G1, dgen1.w(), dsing.w(), ds1.w(), 10, s));
  QCHK(launch_generator_mul(G_G2, dgen2.w(), dsing.w(), ds2.w(), 10, s));
  auto g1col = [&](int col, size_t from) { return d1.w() + ((size_t)col * rows + from) * 26; };
  auto dl = [&](void* h, const void* d, size_t bytes) -> int { if (bytes) QCHK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s)); return ZKT_OK; };
  zkt_g1_affine* mid1[7] = {c->vk_mid, c->g1_wk_mid, c->yk_mid, c->alpha_vk_mid, c->alpha_wk_mid, c->alpha_yk_mid, c->beta_vwy_k_mid};
  for (int col = 0; col < 7; ++col) ZRC(dl(mid1[col], g1col(col, nio), nmid * G1B));
  ZRC(dl(c->g2_wk_mid, d2.w() + nio * 50, nmid * G2B)); ZRC(dl(c->si, d2.w() + rows * 50, deg * G2B));
  ZRC(dl(c->vk_io, g1col(0, 0), nio * G1B)); ZRC(dl(c->yk_io, g1col(2, 0), nio * G1B)); ZRC(dl(c->wk_io, d2.w(), nio * G2B));
  // singles: 0 one, 1 alpha_v, 2 alpha_w, 3 alpha_y, 4 gamma, 5 gamma beta, 6 T, 7 T alpha_v, 8 T alpha_y, 9 T beta
  ZRC(dl(c->one_g1, ds1.w(), G1B)); ZRC(dl(c->one_g2, ds2.w(), G2B)); ZRC(dl(c->alpha_v, ds2.w() + 1 * 50, G2B)); ZRC(dl(c->alpha_w, ds1.w() + 2 * 26, G1B));
  ZRC(dl(c->alpha_y, ds2.w() + 3 * 50, G2B)); ZRC(dl(c->gamma, ds2.w() + 4 * 50, G2B)); ZRC(dl(c->beta_gamma, ds2.w() + 5 * 50, G2B));
  ZRC(dl(c->t, ds1.w() + 6 * 26, G1B)); ZRC(dl(c->alpha_v_t, ds1.w() + 7 * 26, G1B)); ZRC(dl(c->alpha_y_t, ds1.w() + 8 * 26, G1B)); ZRC(dl(c->beta_t, ds1.w() + 9 * 26, G1B));
  QCHK(hipStreamSynchronize(s));
  return ZKT_OK;
}

// Prover::prove (prover.rs:98-170): wires = all (n_io + n_mid) witness values; h = quotient coefficients (prover.rs:143-146)
int zkt_pinocchio_prove(const zkt_pinocchio_crs* c, const uint64_t* wires, const uint64_t* h, size_t h_len, const uint64_t* delta_v, const uint64_t* delta_y,
                        zkt_pinocchio_proof* pf) {
  if (zkt_internal_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (!c || !wires || !delta_v || !delta_y || !pf || (h_len && !h) || h_len > c->max_degree) return ZKT_ERR_SHAPE;   // eval_with_g2_hidings would index-panic (polynomial.rs:289-291)
  const size_t nio = c->n_io, nmid = c->n_mid;
  const uint64_t* wmid = wires + nio * 4;
  // the eight running sums of prover.rs:133-146 are MSMs over the mid wires
  zkt_g1_affine sv, sw1, sy, sav, saw, say, sb; zkt_g2_affine sw2;
  const zkt_g1_affine* bases1[7] = {c->vk_mid, c->g1_wk_mid, c->yk_mid, c->alpha_vk_mid, c->alpha_wk_mid, c->alpha_yk_mid, c->beta_vwy_k_mid};
  zkt_g1_affine* sums1[7] = {&sv, &sw1, &sy, &sav, &saw, &say, &sb};
  for (int k = 0; k < 7; ++k) ZRC(zkt_g1_msm(bases1[k], wmid, nmid, sums1[k]));
  ZRC(zkt_g2_msm(c->g2_wk_mid, wmid, nmid, &sw2));
  // randomisation terms (prover.rs:124-131): t dv, t dy, alpha_v_t dv, alpha_y_t dy, beta_t dv, beta_t dy
  zkt_g1_affine pts[6] = {*c->t, *c->t, *c->alpha_v_t, *c->alpha_y_t, *c->beta_t, *c->beta_t}, rnd[6];
  uint64_t sc[24];
  const uint64_t* which[6] = {delta_v, delta_y, delta_v, delta_y, delta_v, delta_y};
  for (int k = 0; k < 6; ++k) memcpy(sc + 4 * k, which[k], 32);
  ZRC(zkt_g1_mul_batch(pts, sc, 4, rnd, 6));
  zkt_g1_affine bsum;
  ZRC(zkt_g1_add_batch(&rnd[4], &rnd[5], &bsum, 1));
  zkt_g1_affine lhs[4] = {rnd[0], rnd[1], rnd[2], rnd[3]}, rhs[4] = {sv, sy, sav, say}, out4[4];
  ZRC(zkt_g1_add_batch(lhs, rhs, out4, 4));
  *pf->v_mid_s = out4[0]; *pf->y_mid_s = out4[1]; *pf->alpha_v_mid_s = out4[2]; *pf->alpha_y_mid_s = out4[3];
  ZRC(zkt_g1_add_batch(&bsum, &sb, pf->beta_vwy_mid_s, 1));
  *pf->g1_w_mid_s = sw1; *pf->g2_w_mid_s = sw2; *pf->alpha_w_mid_s = saw;
  // adjusted h(s) (prover.rs:148-161): h_s + w_s delta_v - one_g2 delta_y
  zkt_g2_affine h_s, w_io, w_s, wdv, ody, nody, t2;
  ZRC(zkt_g2_msm(c->si, h, h_len, &h_s));
  ZRC(zkt_g2_msm(c->wk_io, wires, nio, &w_io));
  ZRC(zkt_g2_add_batch(&sw2, &w_io, &w_s, 1));
  ZRC(zkt_g2_mul_batch(&w_s, delta_v, 4, &wdv, 1)); ZRC(zkt_g2_mul_batch(c->one_g2, delta_y, 4, &ody, 1)); ZRC(zkt_g2_neg_batch(&ody, &nody, 1));
  ZRC(zkt_g2_add_batch(&h_s, &wdv, &t2, 1)); ZRC(zkt_g2_add_batch(&t2, &nody, pf->h_s, 1));
  return ZKT_OK;
}

}  // extern "C"
// ---- the prover with the evaluation key resident (round 3) --------------------------------------------------------------------------------------
// Prover::prove (prover.rs:98-170) is ten multi-scalar multiplications: seven G1 and one G2 sum over the SAME mid-wire values (prover.rs:133-141), the quotient
// h over the powers s^i in G2 (:143-146) and the io wires over wk_io in G2 (:151-156).  zkt_pinocchio_prove hands each of them to the one-shot entry points, which
// upload the bases and build their plan on every call (50 ms at 32 constraints, seconds at 2^16).  An evaluation key is long-lived, so this handle keeps its ten
// base sets in HBM with their window-multiple tables (zkt_g1_bases / zkt_g2_bases), uploads the wire values ONCE per proof and runs the ten sums through the
// pipelined submit/collect interface; the G1 sets share one group of streams, the G2 sets another (zkt_internal_bases_share_streams: a stream is a hardware queue).
extern "C" int zkt_internal_bases_share_streams(void* dst, void* src, int share_acc, int tail_base, int tail_span);
struct zkt_pinocchio_pk {
  size_t n_io = 0, n_mid = 0, max_degree = 0;
  zkt_g1_bases* g1[7] = {};                       // vk, g1_wk, yk, alpha_vk, alpha_wk, alpha_yk, beta_vwy_k (mid)
  zkt_g2_bases *g2_wk = nullptr, *si = nullptr, *wk_io = nullptr;
  zkt_g1_affine t{}, alpha_v_t{}, alpha_y_t{}, beta_t{}; zkt_g2_affine one_g2{};
  void *d_wires = nullptr, *d_h = nullptr;
  std::mutex mu;                                  // a handle serves one proof at a time
  ~zkt_pinocchio_pk() {
    if (wk_io) zkt_g2_bases_free(wk_io);
    if (si) zkt_g2_bases_free(si);
    for (int k = 6; k >= 1; --k) if (g1[k]) zkt_g1_bases_free(g1[k]);
    if (g2_wk) zkt_g2_bases_free(g2_wk);          // owner of the G2 group's streams: after si and wk_io
    if (g1[0]) zkt_g1_bases_free(g1[0]);          // owner of the G1 group's streams: last
    if (d_wires) hipFree(d_wires);
    if (d_h) hipFree(d_h);
  }
};
extern "C" {
int zkt_pinocchio_pk_create(const zkt_pinocchio_crs* c, zkt_pinocchio_pk** out) {
  if (zkt_internal_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (!c || !out || !c->t || !c->alpha_v_t || !c->alpha_y_t || !c->beta_t || !c->one_g2) return ZKT_ERR_SHAPE;
  std::unique_ptr<zkt_pinocchio_pk> pk(new zkt_pinocchio_pk);
  pk->n_io = c->n_io; pk->n_mid = c->n_mid; pk->max_degree = c->max_degree;
  pk->t = *c->t; pk->alpha_v_t = *c->alpha_v_t; pk->alpha_y_t = *c->alpha_y_t; pk->beta_t = *c->beta_t; pk->one_g2 = *c->one_g2;
  const zkt_g1_affine* b1[7] = {c->vk_mid, c->g1_wk_mid, c->yk_mid, c->alpha_vk_mid, c->alpha_wk_mid, c->alpha_yk_mid, c->beta_vwy_k_mid};
  if (c->n_mid) {
    for (int k = 0; k < 7; ++k) {
      if (!b1[k]) return ZKT_ERR_SHAPE;
      ZRC(zkt_g1_bases_upload(b1[k], c->n_mid, &pk->g1[k]));
      if (k) ZRC(zkt_internal_bases_share_streams(pk->g1[k], pk->g1[0], 1, k % 4, 1));      // below 2^19 terms a slot runs its whole MSM on its reduce stream: four side by side
    }
    if (!c->g2_wk_mid) return ZKT_ERR_SHAPE;
    ZRC(zkt_g2_bases_upload(c->g2_wk_mid, c->n_mid, &pk->g2_wk));
  }
  if (c->max_degree) {
    if (!c->si) return ZKT_ERR_SHAPE;
    ZRC(zkt_g2_bases_upload(c->si, c->max_degree, &pk->si));
    if (pk->g2_wk) ZRC(zkt_internal_bases_share_streams(pk->si, pk->g2_wk, 1, 2, 1));
  }
  if (c->n_io) {
    if (!c->wk_io) return ZKT_ERR_SHAPE;
    ZRC(zkt_g2_bases_upload(c->wk_io, c->n_io, &pk->wk_io));
    if (pk->g2_wk) ZRC(zkt_internal_bases_share_streams(pk->wk_io, pk->g2_wk, 1, 3, 1));
  }
  if (hipMalloc(&pk->d_wires, (c->n_io + c->n_mid ? c->n_io + c->n_mid : 1) * FRB) != hipSuccess || hipMalloc(&pk->d_h, (c->max_degree ? c->max_degree : 1) * FRB) != hipSuccess) return ZKT_ERR_DEVICE;
  *out = pk.release();
  return ZKT_OK;
}
void zkt_pinocchio_pk_free(zkt_pinocchio_pk* pk) { delete pk; }
// Prover::prove (prover.rs:98-170) on a resident key: same arguments and the same nine proof points as zkt_pinocchio_prove
int zkt_pinocchio_prove_resident(zkt_pinocchio_pk* pk, const uint64_t* wires, const uint64_t* h, size_t h_len, const uint64_t* delta_v, const uint64_t* delta_y,
                                 zkt_pinocchio_proof* pf) {
  if (zkt_internal_ready() != ZKT_OK) return ZKT_ERR_DEVICE;
  if (!pk || !wires || !delta_v || !delta_y || !pf || (h_len && !h) || h_len > pk->max_degree) return ZKT_ERR_SHAPE;
  std::lock_guard<std::mutex> lk(pk->mu);
  const size_t nio = pk->n_io, nmid = pk->n_mid;
  hipStream_t s = nullptr;
  QCHK(hipMemcpyAsync(pk->d_wires, wires, (nio + nmid) * FRB, hipMemcpyHostToDevice, s));
  // the quotient is padded with zeros to the key's max_degree terms (the resident set has a fixed length; a zero scalar adds nothing)
  if (h_len) QCHK(hipMemcpyAsync(pk->d_h, h, h_len * FRB, hipMemcpyHostToDevice, s));
  if (h_len < pk->max_degree) QCHK(hipMemsetAsync((char*)pk->d_h + h_len * FRB, 0, (pk->max_degree - h_len) * FRB, s));
  const uint64_t* d_mid = (const uint64_t*)pk->d_wires + nio * 4;
  // every sum in flight at once; whatever was submitted is collected below even after an error (a slot left pending would block the next proof)
  int rc = ZKT_OK, r2; bool sub1[7] = {}, sub_w = false, sub_h = false, sub_io = false;
  if (nmid) {
    for (int k = 0; k < 7 && rc == ZKT_OK; ++k) { rc = zkt_g1_msm_submit(pk->g1[k], d_mid, nmid, s, 0); sub1[k] = rc == ZKT_OK; }
    if (rc == ZKT_OK) { rc = zkt_g2_msm_submit(pk->g2_wk, d_mid, nmid, s, 0); sub_w = rc == ZKT_OK; }
  }
  if (rc == ZKT_OK && h_len) { rc = zkt_g2_msm_submit(pk->si, (const uint64_t*)pk->d_h, pk->max_degree, s, 0); sub_h = rc == ZKT_OK; }
  if (rc == ZKT_OK && nio) { rc = zkt_g2_msm_submit(pk->wk_io, (const uint64_t*)pk->d_wires, nio, s, 0); sub_io = rc == ZKT_OK; }
  // randomisation terms (prover.rs:124-131) while the sums run: t dv, t dy, alpha_v_t dv, alpha_y_t dy, beta_t dv, beta_t dy
  zkt_g1_affine pts[6] = {pk->t, pk->t, pk->alpha_v_t, pk->alpha_y_t, pk->beta_t, pk->beta_t}, rnd[6];
  uint64_t sc[24];
  const uint64_t* which[6] = {delta_v, delta_y, delta_v, delta_y, delta_v, delta_y};
  for (int k = 0; k < 6; ++k) memcpy(sc + 4 * k, which[k], 32);
  const int rc_r = rc == ZKT_OK ? zkt_g1_mul_batch(pts, sc, 4, rnd, 6) : ZKT_OK;
  zkt_g1_affine sums[7]; zkt_g2_affine sw2, h_s, w_io;
  memset(sums, 0, sizeof(sums)); memset(&sw2, 0, sizeof(sw2)); memset(&h_s, 0, sizeof(h_s)); memset(&w_io, 0, sizeof(w_io));
  for (int k = 0; k < 7; ++k) sums[k].is_infinity = 1;
  sw2.is_infinity = 1; h_s.is_infinity = 1; w_io.is_infinity = 1;
  for (int k = 0; k < 7; ++k) if (sub1[k] && (r2 = zkt_g1_msm_collect(pk->g1[k], 0, &sums[k], nullptr)) != ZKT_OK) rc = r2;
  if (sub_w && (r2 = zkt_g2_msm_collect(pk->g2_wk, 0, &sw2, nullptr)) != ZKT_OK) rc = r2;
  if (sub_h && (r2 = zkt_g2_msm_collect(pk->si, 0, &h_s, nullptr)) != ZKT_OK) rc = r2;
  if (sub_io && (r2 = zkt_g2_msm_collect(pk->wk_io, 0, &w_io, nullptr)) != ZKT_OK) rc = r2;
  if (rc != ZKT_OK) return rc;
  if (rc_r != ZKT_OK) return rc_r;
  const zkt_g1_affine &sv = sums[0], &sw1 = sums[1], &sy = sums[2], &sav = sums[3], &saw = sums[4], &say = sums[5], &sb = sums[6];
  zkt_g1_affine bsum;
  ZRC(zkt_g1_add_batch(&rnd[4], &rnd[5], &bsum, 1));
  zkt_g1_affine lhs[5] = {rnd[0], rnd[1], rnd[2], rnd[3], bsum}, rhs[5] = {sv, sy, sav, say, sb}, out5[5];
  ZRC(zkt_g1_add_batch(lhs, rhs, out5, 5));
  *pf->v_mid_s = out5[0]; *pf->y_mid_s = out5[1]; *pf->alpha_v_mid_s = out5[2]; *pf->alpha_y_mid_s = out5[3]; *pf->beta_vwy_mid_s = out5[4];
  *pf->g1_w_mid_s = sw1; *pf->g2_w_mid_s = sw2; *pf->alpha_w_mid_s = saw;
  // adjusted h(s) (prover.rs:148-161): h_s + w_s delta_v - one_g2 delta_y
  zkt_g2_affine w_s, two[2], muls[2], nody, t2;
  ZRC(zkt_g2_add_batch(&sw2, &w_io, &w_s, 1));
  two[0] = w_s; two[1] = pk->one_g2;
  uint64_t sc2[8]; memcpy(sc2, delta_v, 32); memcpy(sc2 + 4, delta_y, 32);
  ZRC(zkt_g2_mul_batch(two, sc2, 4, muls, 2));
  ZRC(zkt_g2_neg_batch(&muls[1], &nody, 1));
  ZRC(zkt_g2_add_batch(&h_s, &muls[0], &t2, 1)); ZRC(zkt_g2_add_batch(&t2, &nody, pf->h_s, 1));
  return ZKT_OK;
}

// Verifier::verify (verifier.rs:31-85): 1 accept, 0 reject, negative = -status (a tate() argument at infinity panics in the reference).
// The checks are evaluated in the reference's order, so a rejection by an earlier check wins over a panic of a later one.
int zkt_pinocchio_verify(const zkt_pinocchio_crs* c, const zkt_pinocchio_proof* pf, const uint64_t* io_wires) {
  if (zkt_internal_ready() != ZKT_OK) return -ZKT_ERR_DEVICE;
  if (!c || !pf || (c->n_io && !io_wires)) return -ZKT_ERR_SHAPE;
  static const bool fast = [] { const char* e = getenv("ZKT_PINOCCHIO_FAST_VERIFY"); return !e || atoi(e) != 0; }();
  if (fast && c->n_io <= PIN_FAST_IO) { const int v = pin_verify_fast(c, pf, io_wires); if (v != PIN_FALL_BACK) return v; }
  int rc;
  zkt_g1_affine t1, vwy;
  if ((rc = zkt_g1_add_batch(pf->v_mid_s, pf->g1_w_mid_s, &t1, 1)) || (rc = zkt_g1_add_batch(&t1, pf->y_mid_s, &vwy, 1))) return -rc;      // :44
  // four two-pair equalities lhs == rhs as tate(lhs) * tate(-rhs) == 1   (:43-66)
  zkt_g1_affine g1s[8] = {*pf->beta_vwy_mid_s, vwy, *pf->alpha_v_mid_s, *pf->v_mid_s, *pf->alpha_w_mid_s, *c->alpha_w, *pf->alpha_y_mid_s, *pf->y_mid_s};
  zkt_g2_affine g2s[8] = {*c->gamma, *c->beta_gamma, *c->one_g2, *c->alpha_v, *c->one_g2, *pf->g2_w_mid_s, *c->one_g2, *c->alpha_y};
  // QAP divisibility check (:69-84): v_s, w_s, y_s add the io part
  zkt_g1_affine v_io, y_io, v_s, y_s; zkt_g2_affine w_io, w_s;
  if ((rc = zkt_g1_msm(c->vk_io, io_wires, c->n_io, &v_io)) || (rc = zkt_g1_msm(c->yk_io, io_wires, c->n_io, &y_io)) || (rc = zkt_g2_msm(c->wk_io, io_wires, c->n_io, &w_io))) return -rc;
  if ((rc = zkt_g1_add_batch(pf->v_mid_s, &v_io, &v_s, 1)) || (rc = zkt_g1_add_batch(pf->y_mid_s, &y_io, &y_s, 1)) || (rc = zkt_g2_add_batch(pf->g2_w_mid_s, &w_io, &w_s, 1))) return -rc;
  zkt_g1_affine q1[3] = {v_s, *c->t, y_s}; zkt_g2_affine q2[3] = {w_s, *pf->h_s, *c->one_g2};
  const uint8_t neg2[2] = {0, 1}, neg3[3] = {0, 1, 1};
  uint32_t ok4[4] = {0, 0, 0, 0}, ok1 = 0;
  int rc4 = zkt_pairing_product_check_batch(g1s, g2s, neg2, 2, 4, ok4);
  size_t inf4 = rc4 == ZKT_ERR_INFINITY ? zkt_last_error_index() : (size_t)-1;
  if (rc4 != ZKT_OK && rc4 != ZKT_ERR_INFINITY) return -rc4;
  for (size_t k = 0; k < 4; ++k) {
    if (k == inf4) return -ZKT_ERR_INFINITY;        // the first check with an argument at infinity: the reference panics here (later lanes may also be at infinity)
    if (inf4 != (size_t)-1 && k > inf4) break;
    if (!ok4[k]) return 0;
  }
  if (inf4 != (size_t)-1) return -ZKT_ERR_INFINITY;
  int rc1 = zkt_pairing_product_check_batch(q1, q2, neg3, 3, 1, &ok1);
  if (rc1 != ZKT_OK) return -rc1;
  return ok1 ? 1 : 0;
}

}  // extern "C"
