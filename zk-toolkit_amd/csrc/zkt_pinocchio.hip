// Pinocchio (protocol 2 of eprint 2013/279) over the batched kernels — SURVEY §8 row f-4:
//   src/zk/w_trusted_setup/pinocchio/{crs.rs:49-161, prover.rs:98-170, verifier.rs:31-85}
// Random values the reference draws from OS entropy (crs.rs:58-64,82; prover.rs:104-105) are arguments.
// Setup is one Fr scalar stage on the device followed by fixed-base multiplications of the two generators (every CRS element
// is generator * scalar, whatever chain of multiplications the reference writes); prove is 7 G1 MSMs + 2 G2 MSMs over the mid
// wires plus a handful of single operations; verify is five pairing-product equalities, the first four in one launch.
#include <vector>
#include <cstring>
#include <cstdio>
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
  QCHK(launch_group_mul(G_G1, dgen1.w(), dcols.w(), 8, d1.w(), 7 * rows, s, true));
  QCHK(launch_group_mul(G_G2, dgen2.w(), dcols.w() + rows * 8, 8, d2.w(), rows, s, true));                 // g2_w * w_i(s)
  QCHK(launch_group_mul(G_G2, dgen2.w(), dpow.w(), 8, d2.w() + rows * 50, deg, s, true));                   // si
  QCHK(launch_group_mul(G_G1, dgen1.w(), dsing.w(), 8, ds1.w(), 10, s, true));
  QCHK(launch_group_mul(G_G2, dgen2.w(), dsing.w(), 8, ds2.w(), 10, s, true));
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

// Verifier::verify (verifier.rs:31-85): 1 accept, 0 reject, negative = -status (a tate() argument at infinity panics in the reference).
// The checks are evaluated in the reference's order, so a rejection by an earlier check wins over a panic of a later one.
int zkt_pinocchio_verify(const zkt_pinocchio_crs* c, const zkt_pinocchio_proof* pf, const uint64_t* io_wires) {
  if (zkt_internal_ready() != ZKT_OK) return -ZKT_ERR_DEVICE;
  if (!c || !pf || (c->n_io && !io_wires)) return -ZKT_ERR_SHAPE;
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
