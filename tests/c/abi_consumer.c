/* A plain C99 consumer of include/zkt.h — what a cgo / Rust-FFI binding sees: no Python, no torch, only the C ABI.
 * Checks on the device: Fq (a*b)*b^-1 == a, inverse of zero -> status + index, 2G via add == 2G via scalar mul == MSM of
 * {G, G} with scalars {1, 1}, tate(2G, H) == tate(G, H)^2 for H = 3 G2 (bilinearity as in pairing.rs:107-123), no-init error. */
#include <stdio.h>
#include <string.h>
#include "zkt.h"

static const uint64_t G1X[6] = {0xfb3af00adb22c6bbull, 0x6c55e83ff97a1aefull, 0xa14e3a3f171bac58ull, 0xc3688c4f9774b905ull, 0x2695638c4fa9ac0full, 0x17f1d3a73197d794ull};   /* g1_point.rs:38-47 */
static const uint64_t G1Y[6] = {0x0caa232946c5e7e1ull, 0xd03cc744a2888ae4ull, 0x00db18cb2c04b3edull, 0xfcf5e095d5d00af6ull, 0xa09e30ed741d8ae4ull, 0x08b3f481e3aaa0f1ull};
static const uint64_t G2XY[24] = {0xe5ac7d055d042b7eull, 0x334cf11213945d57ull, 0xb5da61bbdc7f5049ull, 0x596bd0d09920b61aull, 0x7dacd3a088274f65ull, 0x13e02b6052719f60ull,
                                   0xd48056c8c121bdb8ull, 0x0bac0326a805bbefull, 0xb4510b647ae3d177ull, 0xc6e47ad4fa403b02ull, 0x260805272dc51051ull, 0x024aa2b2f08f0a91ull,
                                   0xaaa9075ff05f79beull, 0x3f370d275cec1da1ull, 0x267492ab572e99abull, 0xcb3e287e85a763afull, 0x32acd2b02bc28b99ull, 0x0606c4a02ea734ccull,
                                   0xe193548608b82801ull, 0x923ac9cc3baca289ull, 0x6d429a695160d12cull, 0xadfd9baa8cbdd3a7ull, 0x8cc9cdc6da2e351aull, 0x0ce5d527727d6e11ull};   /* g2_point.rs:36-46 */
#define CHECK(c) do { if (!(c)) { printf("FAIL line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main(void) {
  uint64_t a[6] = {5, 0, 0, 0, 0, 1}, b[6] = {0xdeadbeef, 7, 0, 0, 0, 0}, ab[6], binv[6], back[6], zero[12] = {0};
  CHECK(zkt_init(-1) == ZKT_OK);
  CHECK(zkt_fq_mul_batch(a, b, ab, 1) == ZKT_OK);
  CHECK(zkt_fq_inv_batch(b, binv, 1) == ZKT_OK);
  CHECK(zkt_fq_mul_batch(ab, binv, back, 1) == ZKT_OK);
  CHECK(memcmp(back, a, sizeof a) == 0);
  memcpy(zero, b, sizeof b);                                  /* {b, 0}: the second element has no inverse */
  { uint64_t out[12]; CHECK(zkt_fq_inv_batch(zero, out, 2) == ZKT_ERR_INV_ZERO); CHECK(zkt_last_error_index() == 1); }

  zkt_g1_affine g, two_add, two_mul, two_msm, gg[2];
  memset(&g, 0, sizeof g); memcpy(g.x, G1X, sizeof G1X); memcpy(g.y, G1Y, sizeof G1Y);
  uint64_t k2[4] = {2, 0, 0, 0}, ones[8] = {1, 0, 0, 0, 1, 0, 0, 0};
  CHECK(zkt_g1_add_batch(&g, &g, &two_add, 1) == ZKT_OK);
  CHECK(zkt_g1_mul_batch(&g, k2, 4, &two_mul, 1) == ZKT_OK);
  gg[0] = g; gg[1] = g;
  CHECK(zkt_g1_msm(gg, ones, 2, &two_msm) == ZKT_OK);
  CHECK(memcmp(&two_add, &two_mul, sizeof g) == 0 && memcmp(&two_add, &two_msm, sizeof g) == 0 && two_add.is_infinity == 0);

  zkt_g2_affine h2, h;
  memset(&h2, 0, sizeof h2); memcpy(h2.x, G2XY, 96); memcpy(h2.y, G2XY + 12, 96);
  uint64_t k3[4] = {3, 0, 0, 0};
  CHECK(zkt_g2_mul_batch(&h2, k3, 4, &h, 1) == ZKT_OK);
  zkt_g1_affine ps[2]; zkt_g2_affine qs[2]; uint64_t e[2][72], sq[72];
  ps[0] = g; ps[1] = two_add; qs[0] = h; qs[1] = h;
  CHECK(zkt_tate_batch(ps, qs, &e[0][0], 2) == ZKT_OK);
  CHECK(zkt_fq12_mul_batch(e[0], e[0], sq, 1) == ZKT_OK);
  CHECK(zkt_gt_eq(sq, e[1]) == 1 && zkt_gt_eq(e[0], e[1]) == 0);
  printf("abi_consumer ok\n");
  return 0;
}
