// TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See zkt_oracle.hpp for the header
// note: CPU restatement of the reference algorithm, pinned by the reference's
// own KATs (tests/golden/ref_kats.json).  Paths below are relative to
// /root/reference/.
#include "zkt_oracle.hpp"
#include <mutex>
#include <string>

namespace zkto {

FieldParams FqTag::P, FrTag::P, SpTag::P, SnTag::P, DynTag::P;

static void parse_hex(const char* s, uint64_t* out, int n) {
  for (int i = 0; i < n; ++i) out[i] = 0;
  int len = (int)strlen(s);
  for (int i = 0; i < len; ++i) {
    char c = s[len - 1 - i];
    uint64_t v = (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : c - 'A' + 10;
    if (i / 16 < n) out[i / 16] |= v << (4 * (i % 16));
  }
}
static Fq1 fq_hex(const char* s) { uint64_t l[MAXL]; parse_hex(s, l, MAXL); return Fq1::from_limbs(l, MAXL); }

static std::once_flag g_once;
void init_fields() {
  std::call_once(g_once, [] {
    uint64_t m[MAXL];
    // curves/bls12_381/params.rs:9
    parse_hex("1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab", m, MAXL);
    FqTag::P.init_from_limbs(m, MAXL);
    // curves/bls12_381/params.rs:14
    parse_hex("73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001", m, MAXL);
    FrTag::P.init_from_limbs(m, MAXL);
    // curves/secp256k1/affine_point.rs:30-47
    parse_hex("FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F", m, MAXL);
    SpTag::P.init_from_limbs(m, MAXL);
    parse_hex("FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141", m, MAXL);
    SnTag::P.init_from_limbs(m, MAXL);
    DynTag::P = FrTag::P;
  });
}

G1Point g1_generator() {   // g1_point.rs:38-47
  return G1Point(
      fq_hex("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"),
      fq_hex("08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1"));
}
G2Point g2_generator() {   // g2_point.rs:36-46; Fq2::new(u1, u0)
  Fq1 x1 = fq_hex("13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e");
  Fq1 x0 = fq_hex("024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8");
  Fq1 y1 = fq_hex("0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be");
  Fq1 y0 = fq_hex("0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801");
  return G2Point(Fq2(x1, x0), Fq2(y1, y0));
}
SecpPoint secp_generator() {   // secp256k1/affine_point.rs:30-47
  uint64_t gx[MAXL], gy[MAXL];
  parse_hex("79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798", gx, MAXL);
  parse_hex("483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8", gy, MAXL);
  return SecpPoint(Fp<SpTag>::from_limbs(gx, 4), Fp<SpTag>::from_limbs(gy, 4));
}

// ---------------------------------------------------------------------------
// G12Point (g12_point.rs)
// ---------------------------------------------------------------------------
G12Point g12_from_g1(const G1Point& p) {          // :29-44
  G12Point r; r.inf = p.inf;
  if (!p.inf) { r.x = Fq12::from_fq(p.x); r.y = Fq12::from_fq(p.y); }
  return r;
}
G12Point g12_from_g2(const G2Point& p) {          // :47-68 — recomputed (2 Fq12 inv) on every call, as the reference does
  G12Point r; r.inf = p.inf;
  if (p.inf) return r;
  Fq2 one = Fq2::from_u64(1);
  Fq6 root(Fq2::zero(), one, Fq2::zero());
  Fq6 x6_w0(Fq2::zero(), Fq2::zero(), p.x);
  Fq6 y6_w0(Fq2::zero(), Fq2::zero(), p.y);
  r.x = Fq12(Fq6::zero(), x6_w0) * Fq12(Fq6::zero(), root).inv();
  r.y = Fq12(Fq6::zero(), y6_w0) * Fq12(root, Fq6::zero()).inv();
  return r;
}

// ---------------------------------------------------------------------------
// RationalFunction (rational_function.rs:20-101)
// ---------------------------------------------------------------------------
static RationalFunction handle_tangent(const Fq12& x, const Fq12& y) {        // :70-83
  Fq12 two = Fq12::from_u64(2), three = Fq12::from_u64(3);
  RationalFunction rf; rf.vertical = false; rf.x = x; rf.y = y;
  rf.slope = three * x * x * (two * y).inv();
  return rf;
}
static RationalFunction handle_vertical(const Fq12& x) {                      // :85-89
  RationalFunction rf; rf.vertical = true; rf.x = x; return rf;
}
static RationalFunction handle_others(const Fq12& x1, const Fq12& y1, const Fq12& x2, const Fq12& y2) {  // :91-101
  RationalFunction rf; rf.vertical = false; rf.x = x1; rf.y = y1;
  rf.slope = (y2 - y1) * (x2 - x1).inv();
  return rf;
}
template <class P>
static RationalFunction rf_new(const P& p, const P& q, G12Point (*conv)(const P&)) {   // :20-43
  G12Point p12 = conv(p), q12 = conv(q);
  if (p12.inf || q12.inf) throw std::domain_error("Both points need to be rational");
  if (p == q) return handle_tangent(p12.x, p12.y);
  if (q == p.neg()) return handle_vertical(p12.x);
  return handle_others(p12.x, p12.y, q12.x, q12.y);
}
RationalFunction RationalFunction::new_g1(const G1Point& p, const G1Point& q) { return rf_new<G1Point>(p, q, g12_from_g1); }
RationalFunction RationalFunction::new_g2(const G2Point& p, const G2Point& q) { return rf_new<G2Point>(p, q, g12_from_g2); }

static Fq12 rf_eval(const RationalFunction& rf, const G12Point& q12) {        // :45-66
  if (q12.inf) throw std::domain_error("cannot evaluate with point at infinity");
  if (rf.vertical) return q12.x + (-rf.x);
  return (-rf.slope) * q12.x + q12.y + (-rf.y) + rf.slope * rf.x;
}
Fq12 RationalFunction::eval_with_g1(const G1Point& q) const { return rf_eval(*this, g12_from_g1(q)); }
Fq12 RationalFunction::eval_with_g2(const G2Point& q) const { return rf_eval(*this, g12_from_g2(q)); }

// ---------------------------------------------------------------------------
// mini big-integer (u32 limbs) for the one-off (q^12-1)/r of pairing.rs:94-97
// ---------------------------------------------------------------------------
typedef std::vector<uint32_t> BigV;
static void bv_trim(BigV& a) { while (!a.empty() && a.back() == 0) a.pop_back(); }
static BigV bv_mul(const BigV& a, const BigV& b) {
  BigV r(a.size() + b.size(), 0);
  for (size_t i = 0; i < a.size(); ++i) {
    uint64_t c = 0;
    for (size_t j = 0; j < b.size(); ++j) { c += (uint64_t)a[i] * b[j] + r[i + j]; r[i + j] = (uint32_t)c; c >>= 32; }
    r[i + b.size()] = (uint32_t)c;
  }
  bv_trim(r); return r;
}
static int bv_cmp(const BigV& a, const BigV& b) {
  if (a.size() != b.size()) return a.size() < b.size() ? -1 : 1;
  for (size_t i = a.size(); i-- > 0;) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return 0;
}
static void bv_sub_inplace(BigV& a, const BigV& b) {   // a >= b
  int64_t borrow = 0;
  for (size_t i = 0; i < a.size(); ++i) {
    int64_t d = (int64_t)a[i] - (i < b.size() ? b[i] : 0) - borrow;
    borrow = d < 0; a[i] = (uint32_t)(d + (borrow ? ((int64_t)1 << 32) : 0));
  }
  bv_trim(a);
}
static BigV bv_divexact_check(const BigV& num, const BigV& den, bool& exact) {
  BigV q(num.size(), 0), rem;
  for (int bit = (int)num.size() * 32 - 1; bit >= 0; --bit) {
    // rem = rem*2 + bit
    uint32_t carry = (num[bit / 32] >> (bit % 32)) & 1;
    for (size_t i = 0; i < rem.size(); ++i) { uint32_t nc = rem[i] >> 31; rem[i] = (rem[i] << 1) | carry; carry = nc; }
    if (carry) rem.push_back(carry);
    bv_trim(rem);
    if (bv_cmp(rem, den) >= 0) { bv_sub_inplace(rem, den); q[bit / 32] |= 1u << (bit % 32); }
  }
  exact = rem.empty();
  bv_trim(q); return q;
}
static BigV bv_from_limbs64(const uint64_t* l, int n) {
  BigV r; for (int i = 0; i < n; ++i) { r.push_back((uint32_t)l[i]); r.push_back((uint32_t)(l[i] >> 32)); }
  bv_trim(r); return r;
}

// ---------------------------------------------------------------------------
// Pairing (pairing.rs)
// ---------------------------------------------------------------------------
Pairing::Pairing() {
  init_fields();
  // :58-73 — bits of r-1, MSB first, leading 1 dropped
  uint64_t l[4]; uint64_t one[4] = {1, 0, 0, 0};
  limb_sub(l, FrTag::P.p, one, 4);
  std::vector<bool> bits;
  int top = 256; while (top > 0 && !((l[(top - 1) / 64] >> ((top - 1) % 64)) & 1)) --top;
  for (int i = 0; i < top; ++i) bits.push_back((l[i / 64] >> (i % 64)) & 1);   // LSB first
  l_bits.assign(bits.rbegin(), bits.rend());
  l_bits.erase(l_bits.begin());
  // :94-97 — exp = (q^12 - 1) / r
  BigV q = bv_from_limbs64(FqTag::P.p, 6), r = bv_from_limbs64(FrTag::P.p, 4);
  BigV q12 = q;
  for (int i = 1; i < 12; ++i) q12 = bv_mul(q12, q);   // embedding_degree 12, params.rs:27-29
  BigV onev(1, 1); bv_sub_inplace(q12, onev);
  bool exact = false;
  final_exp = bv_divexact_check(q12, r, exact);
  // BigUint `/` floors; r | q^12-1 so the division is exact — assert our reading
  if (!exact) throw std::logic_error("oracle: r does not divide q^12-1");
}

// impl_miller_algorithm! (pairing.rs:20-53)
template <class P1, class P2>
static Fq12 miller(const std::vector<bool>& l_bits, const P1& p, const P2& q,
                   RationalFunction (*mk)(const P1&, const P1&),
                   Fq12 (RationalFunction::*eval_at)(const P2&) const) {
  Fq12 f = Fq12::from_u64(1);
  P1 V = p;
  for (bool bit : l_bits) {
    {
      P1 v2 = affine_add(V, V);
      RationalFunction g_num = mk(V, V);
      RationalFunction g_deno = mk(v2, v2.neg());
      f = (f * f) * (g_num.*eval_at)(q) * (g_deno.*eval_at)(q).inv();
    }
    V = affine_add(V, V);
    if (bit) {
      {
        P1 v_plus_p = affine_add(V, p);
        RationalFunction g_num = mk(V, p);
        RationalFunction g_deno = mk(v_plus_p, v_plus_p.neg());
        f = f * (g_num.*eval_at)(q) * (g_deno.*eval_at)(q).inv();
      }
      V = affine_add(V, p);
    }
  }
  return f;
}
Fq12 Pairing::calc_g1_g2(const G1Point& p, const G2Point& q) const {
  return miller<G1Point, G2Point>(l_bits, p, q, RationalFunction::new_g1, &RationalFunction::eval_with_g2);
}
Fq12 Pairing::calc_g2_g1(const G2Point& p, const G1Point& q) const {
  return miller<G2Point, G1Point>(l_bits, p, q, RationalFunction::new_g2, &RationalFunction::eval_with_g1);
}
Fq12 Pairing::weil(const G1Point& p1, const G2Point& p2) const {   // :75-84
  Fq12 num = calc_g1_g2(p1, p2);
  Fq12 deno = calc_g2_g1(p2, p1);
  return num * deno.inv();
}
Fq12 Pairing::tate(const G1Point& p1, const G2Point& p2) const {   // :86-100
  Fq12 intmed = calc_g1_g2(p1, p2);
  return intmed.pow_bits(final_exp);
}

}  // namespace zkto
