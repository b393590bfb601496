/* zkt.h — C ABI of the MI355X-native engine for zk-toolkit's data-parallel hot path.
 *
 * The reference (exfinen/zk-toolkit, paths below relative to /root/reference/) has no
 * FFI of its own: its public surface is value types with operator overloads.  The one
 * precedent for a native backend is src/building_block/mcl/ (POD handle in a newtype,
 * free functions with out-params, one global init behind Once: mcl_g1.rs:77-93,
 * mcl/pairing.rs:11-17, mcl_initializer.rs:4-15).  This header is what a Rust `extern "C"`
 * block for the zktoolkit_based path binds (INTEGRATION.md shows the stub); every entry
 * point is batch-first and names the reference symbol it replaces.
 *
 * Data layout (all little-endian, plain arrays of uint64_t, caller-owned):
 *   Fq   6 limbs, canonical residue in [0,q)      (PrimeFieldElem.e, prime_field_elem.rs:263-272)
 *   Fr   4 limbs, canonical residue in [0,r)
 *   scalars for point multiplication: `scalar_limbs` limbs each (4 or 6), used as-is,
 *        NOT reduced mod r (macros.rs:10-21)
 *   Fq2  {u1,u0} (fq2.rs:16-19) = 12 limbs; Fq6 {v2,v1,v0} (fq6.rs:16-20) = 36 limbs;
 *   Fq12 {w1,w0} (fq12.rs:18-21) = 72 limbs = the order of the tests' to_strs (fq12.rs:179-195)
 *   zkt_g1_affine / zkt_g2_affine / zkt_secp_affine: {x, y, is_infinity} mirroring
 *        enum {Rational{x,y}, AtInfinity} (g1_point.rs:32-36, g2_point.rs:30-34,
 *        secp256k1/affine_point.rs:23-27); x = y = 0 when is_infinity != 0.
 * Montgomery form, projective coordinates and the limb layout (28-bit limbs for Fq, 32-bit for the 256-bit fields) are internal to the kernels.
 *
 * Errors: the reference panics (inverse of zero prime_field_elem.rs:380-382,434-436;
 * line through / evaluation at infinity rational_function.rs:36,59; index mismatch
 * polynomial.rs:277-279).  Here every call returns a status and zkt_last_error_index()
 * gives the first offending element; a Rust shim turns non-OK into panic!.
 *
 * Threading: zkt_init once (idempotent) binds the library to ONE device; every entry point selects that device for the calling thread.
 * Calls are blocking unless documented otherwise and may come from any thread: independent batch calls are serialised on the library's
 * staging stream; calls on one handle (bases, proving key, context) are serialised by a per-handle lock (submit / collect of different
 * slots of one handle may be issued from different threads).  No host pointer is retained past return.  There is NO CPU fallback:
 * without a HIP device every compute entry point returns ZKT_ERR_DEVICE.
 *
 * `_dev` variants take DEVICE pointers in the same layouts plus a hipStream_t (as void*)
 * and are asynchronous on that stream except where they return a host result.
 */
#ifndef ZKT_H
#define ZKT_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ZKT_OK 0
#define ZKT_ERR_INV_ZERO 1   /* inverse of zero */
#define ZKT_ERR_INFINITY 2   /* pairing argument at infinity */
#define ZKT_ERR_SHAPE 3      /* bad size / null pointer / index mismatch (polynomial.rs:277-279) */
#define ZKT_ERR_DEVICE 4     /* no HIP device, HIP error, or library not initialised */

typedef struct { uint64_t x[6], y[6]; uint32_t is_infinity, _pad; } zkt_g1_affine;    /* 104 B */
typedef struct { uint64_t x[12], y[12]; uint32_t is_infinity, _pad; } zkt_g2_affine;  /* 200 B; x = {u1,u0} */
typedef struct { uint64_t x[4], y[4]; uint32_t is_infinity, _pad; } zkt_secp_affine;  /*  72 B */

/* lifecycle — mcl_initializer.rs:4-15 (init once, panic on failure) */
int zkt_init(int device);                 /* device = HIP ordinal, -1 = current */
void zkt_shutdown(void);                  /* releases everything bound to the device of this zkt_init: the communicator of zkt_comm_init (zkt_comm_finalize), the last one-shot Bulletproofs context, the statement tables of the last four Groth16 keys, the io-point tables of the last two Pinocchio keys, the comb tables of the two BLS12-381 generators and the library's side stream.  Handles the caller still owns (zkt_*_bases, contexts, keys) must be freed BEFORE it.  zkt_init may then be called again, on the same or another device */
int zkt_version(void);
const char* zkt_strerror(int status);
size_t zkt_last_error_index(void);        /* thread-local; valid after a non-OK return */

/* a1–a3: PrimeFieldElem::{plus,minus,times,sq,cube,negate,inv,pow,pow_seq,repeat} prime_field_elem.rs:278-457, over the four prime fields
 * the reference instantiates: fq / fr = BLS12-381 base field and subgroup order (params.rs:8-16), sp / sn = secp256k1 base field and group
 * order (secp256k1/affine_point.rs:30-47).  Elements are 6 limbs (fq) or 4 limbs (fr, sp, sn).  A limb vector that is not below the field
 * order is reduced on load, exactly as PrimeFieldElem::new does (prime_field_elem.rs:263-272); outputs are always canonical.
 *   inv        ZKT_ERR_INV_ZERO (+ index) on a zero, like safe_inv / inv (:379-382, :434-436)
 *   pow        out[i] = a[i] ^ e_i with e_i = exp_limbs little-endian u64 limbs at exps + i*exp_limbs (exp_shared != 0: one exponent at
 *              exps for every element); exponent 0 gives 1 for every base (:311-328)
 *   pow_seq    out[i] = base^i, i < n (:346-361);  repeat  out[i] = base (:363-376) */
int zkt_fq_add_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int zkt_fq_sub_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int zkt_fq_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int zkt_fq_sqr_batch(const uint64_t* a, uint64_t* out, size_t n);
int zkt_fq_cube_batch(const uint64_t* a, uint64_t* out, size_t n);
int zkt_fq_neg_batch(const uint64_t* a, uint64_t* out, size_t n);
int zkt_fq_inv_batch(const uint64_t* a, uint64_t* out, size_t n);
int zkt_fq_pow_batch(const uint64_t* a, const uint64_t* exps, size_t exp_limbs, int exp_shared, uint64_t* out, size_t n);
int zkt_fq_pow_seq(const uint64_t* base, size_t n, uint64_t* out);
int zkt_fq_repeat(const uint64_t* base, size_t n, uint64_t* out);
int zkt_fr_add_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int zkt_fr_sub_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int zkt_fr_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int zkt_fr_sqr_batch(const uint64_t* a, uint64_t* out, size_t n);
int zkt_fr_cube_batch(const uint64_t* a, uint64_t* out, size_t n);
int zkt_fr_neg_batch(const uint64_t* a, uint64_t* out, size_t n);
int zkt_fr_inv_batch(const uint64_t* a, uint64_t* out, size_t n);
int zkt_fr_pow_batch(const uint64_t* a, const uint64_t* exps, size_t exp_limbs, int exp_shared, uint64_t* out, size_t n);
int zkt_fr_pow_seq(const uint64_t* base, size_t n, uint64_t* out);
int zkt_fr_repeat(const uint64_t* base, size_t n, uint64_t* out);
int zkt_sp_add_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int zkt_sp_sub_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int zkt_sp_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int zkt_sp_sqr_batch(const uint64_t* a, uint64_t* out, size_t n);
int zkt_sp_cube_batch(const uint64_t* a, uint64_t* out, size_t n);
int zkt_sp_neg_batch(const uint64_t* a, uint64_t* out, size_t n);
int zkt_sp_inv_batch(const uint64_t* a, uint64_t* out, size_t n);
int zkt_sp_pow_batch(const uint64_t* a, const uint64_t* exps, size_t exp_limbs, int exp_shared, uint64_t* out, size_t n);
int zkt_sp_pow_seq(const uint64_t* base, size_t n, uint64_t* out);
int zkt_sp_repeat(const uint64_t* base, size_t n, uint64_t* out);
int zkt_sn_add_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int zkt_sn_sub_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int zkt_sn_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int zkt_sn_sqr_batch(const uint64_t* a, uint64_t* out, size_t n);
int zkt_sn_cube_batch(const uint64_t* a, uint64_t* out, size_t n);
int zkt_sn_neg_batch(const uint64_t* a, uint64_t* out, size_t n);
int zkt_sn_inv_batch(const uint64_t* a, uint64_t* out, size_t n);
int zkt_sn_pow_batch(const uint64_t* a, const uint64_t* exps, size_t exp_limbs, int exp_shared, uint64_t* out, size_t n);
int zkt_sn_pow_seq(const uint64_t* base, size_t n, uint64_t* out);
int zkt_sn_repeat(const uint64_t* base, size_t n, uint64_t* out);

/* a18 (vector forms of the field, used by the Bulletproofs code): PrimeFieldElems::sum prime_field_elems.rs:35-41 (the fold acc + x from zero;
 * an EMPTY vector is the reference's assert -> ZKT_ERR_SHAPE) and PrimeFieldElems * PrimeFieldElem :152-175 (every element times ONE scalar k;
 * empty vector -> ZKT_ERR_SHAPE likewise).  The element-wise +, -, * of two vectors (:90-150) are the *_add/sub/mul_batch calls above. */
int zkt_fq_sum(const uint64_t* a, size_t n, uint64_t* out);
int zkt_fr_sum(const uint64_t* a, size_t n, uint64_t* out);
int zkt_sp_sum(const uint64_t* a, size_t n, uint64_t* out);
int zkt_sn_sum(const uint64_t* a, size_t n, uint64_t* out);
int zkt_fq_scale_batch(const uint64_t* a, const uint64_t* k, uint64_t* out, size_t n);
int zkt_fr_scale_batch(const uint64_t* a, const uint64_t* k, uint64_t* out, size_t n);
int zkt_sp_scale_batch(const uint64_t* a, const uint64_t* k, uint64_t* out, size_t n);
int zkt_sn_scale_batch(const uint64_t* a, const uint64_t* k, uint64_t* out, size_t n);

/* a4–a6: Fq2 fq2.rs:21-151, Fq6 fq6.rs:22-171, Fq12 fq12.rs:23-172 */
int zkt_fq2_add_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int zkt_fq2_sub_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int zkt_fq2_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int zkt_fq2_inv_batch(const uint64_t* a, uint64_t* out, size_t n);
int zkt_fq2_neg_batch(const uint64_t* a, uint64_t* out, size_t n);
int zkt_fq2_reduce_batch(const uint64_t* a, uint64_t* out, size_t n);   /* Fq2::reduce = x(1+u), fq2.rs:52-58 */
int zkt_fq6_add_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int zkt_fq6_sub_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int zkt_fq6_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int zkt_fq6_inv_batch(const uint64_t* a, uint64_t* out, size_t n);
int zkt_fq6_neg_batch(const uint64_t* a, uint64_t* out, size_t n);
int zkt_fq6_reduce_batch(const uint64_t* a, uint64_t* out, size_t n);   /* Fq6::reduce = x v, fq6.rs:54-62 */
int zkt_fq12_add_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int zkt_fq12_sub_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int zkt_fq12_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);   /* also GTPoint `*`, gt_point.rs:16-31 */
int zkt_fq12_inv_batch(const uint64_t* a, uint64_t* out, size_t n);
int zkt_fq12_neg_batch(const uint64_t* a, uint64_t* out, size_t n);
/* Fq12::pow fq12.rs:42-57: every element to the same exponent (little-endian u32 limbs) */
int zkt_fq12_pow_batch(const uint64_t* a, const uint32_t* exp_limbs, size_t exp_nlimbs, uint64_t* out, size_t n);

/* a7, a16: impl_affine_add! macros.rs:34-163; Neg g1_point.rs:177-195 */
int zkt_g1_add_batch(const zkt_g1_affine* a, const zkt_g1_affine* b, zkt_g1_affine* out, size_t n);
int zkt_g1_neg_batch(const zkt_g1_affine* a, zkt_g1_affine* out, size_t n);
int zkt_g2_add_batch(const zkt_g2_affine* a, const zkt_g2_affine* b, zkt_g2_affine* out, size_t n);
int zkt_g2_neg_batch(const zkt_g2_affine* a, zkt_g2_affine* out, size_t n);
int zkt_secp_add_batch(const zkt_secp_affine* a, const zkt_secp_affine* b, zkt_secp_affine* out, size_t n);
/* a16: generators G1Point::g() g1_point.rs:38-59, G2Point::g() g2_point.rs:36-58, secp256k1 AffinePoint::g() affine_point.rs:40-60 (host constants) */
void zkt_g1_generator(zkt_g1_affine* out);
void zkt_g2_generator(zkt_g2_affine* out);
void zkt_secp_generator(zkt_secp_affine* out);
/* a16: RationalPoint::is_rational_point g1_point.rs:97-113, g2_point.rs:70-82, secp256k1/affine_point.rs:92-104 — out[i] = 1 iff point i is
 * rational and satisfies y^2 = x^3 + b (false at infinity).  The constructors G1Point::new / G2Point::new do not check this, and neither do
 * the entry points of this header: group law and pairing follow the reference's formulas on whatever coordinates they are given. */
int zkt_g1_is_on_curve_batch(const zkt_g1_affine* points, uint32_t* out, size_t n);
int zkt_g2_is_on_curve_batch(const zkt_g2_affine* points, uint32_t* out, size_t n);
int zkt_secp_is_on_curve_batch(const zkt_secp_affine* points, uint32_t* out, size_t n);
/* out[i] = 1 iff order * P_i == infinity (order = r for G1/G2, n for secp256k1): membership of the prime-order subgroup every point the
 * reference builds lies in (g * k; get_random_point g1_point.rs:83-88).  No reference counterpart; see the pairing entry points below for why it matters. */
int zkt_g1_in_subgroup_batch(const zkt_g1_affine* points, uint32_t* out, size_t n);
int zkt_g2_in_subgroup_batch(const zkt_g2_affine* points, uint32_t* out, size_t n);
int zkt_secp_in_subgroup_batch(const zkt_secp_affine* points, uint32_t* out, size_t n);
/* a8: impl_scalar_mul_point! macros.rs:1-32 — out[i] = scalars[i] * points[i] */
int zkt_g1_mul_batch(const zkt_g1_affine* points, const uint64_t* scalars, int scalar_limbs, zkt_g1_affine* out, size_t n);
int zkt_g2_mul_batch(const zkt_g2_affine* points, const uint64_t* scalars, int scalar_limbs, zkt_g2_affine* out, size_t n);
int zkt_secp_mul_batch(const zkt_secp_affine* points, const uint64_t* scalars, int scalar_limbs, zkt_secp_affine* out, size_t n);
/* a18 (vector forms of the group): AffinePoints::sum secp256k1/affine_points.rs:25-31 — the fold from AffinePoint::zero(), so n = 0 gives the point
 * at infinity — and AffinePoints * PrimeFieldElem :105-122, every point times ONE scalar k (scalar_limbs u64 limbs, used as-is like a8).  The
 * element-wise point-vector + point-vector (:84-103) and point-vector * scalar-vector (:124-144) are *_add_batch / *_mul_batch above; the same calls
 * exist for G1 and G2. */
int zkt_g1_sum(const zkt_g1_affine* points, size_t n, zkt_g1_affine* out);
int zkt_g2_sum(const zkt_g2_affine* points, size_t n, zkt_g2_affine* out);
int zkt_secp_sum(const zkt_secp_affine* points, size_t n, zkt_secp_affine* out);
int zkt_g1_scale_batch(const zkt_g1_affine* points, const uint64_t* k, int scalar_limbs, zkt_g1_affine* out, size_t n);
int zkt_g2_scale_batch(const zkt_g2_affine* points, const uint64_t* k, int scalar_limbs, zkt_g2_affine* out, size_t n);
int zkt_secp_scale_batch(const zkt_secp_affine* points, const uint64_t* k, int scalar_limbs, zkt_secp_affine* out, size_t n);
/* a9: Polynomial::eval_with_g1_hidings polynomial.rs:271-281 — out = sum_i scalars[i]*bases[i];
 * scalars are 4 limbs (256 bits) each, used as-is */
int zkt_g1_msm(const zkt_g1_affine* bases, const uint64_t* scalars, size_t n, zkt_g1_affine* out);

/* f-1 and the secp256k1 vector form: Polynomial::eval_with_g2_hidings polynomial.rs:283-293;
 * (AffinePoints * PrimeFieldElems).sum() secp256k1/affine_points.rs:25-31,123-144.  4-limb scalars, used as-is.
 * Same Pippenger pipeline as G1, instantiated over Fq2 / the secp256k1 field. */
int zkt_g2_msm(const zkt_g2_affine* bases, const uint64_t* scalars, size_t n, zkt_g2_affine* out);
int zkt_secp_msm(const zkt_secp_affine* bases, const uint64_t* scalars, size_t n, zkt_secp_affine* out);

/* a10–a13: Pairing::tate pairing.rs:86-100 — out[i] = Fq12 of tate(g1[i], g2[i]).
 * Domain: bit-identical to the reference for every pair of coordinates.  For P in G1 and Q in G2 (every point the reference constructs) the value
 * comes from a 127-step Miller loop, tate = eta^(2x^2-1) for the twisted-ate value eta (csrc/pairing.h); both memberships and both curve equations
 * are tested exactly, per element.  Q on the twist outside G2 takes the 255-step loop over r - 1; P outside G1 or a point off its curve is
 * recomputed on the reference's own chain, including its panics: ZKT_ERR_INFINITY (+ index) when an argument or a multiple of P met by that
 * chain is the point at infinity (rational_function.rs:36,59). */
int zkt_tate_batch(const zkt_g1_affine* g1, const zkt_g2_affine* g2, uint64_t* out_fq12, size_t n);
/* a12, a14: raw Miller values and the Weil pairing, bit-exact (the reference uses them in its tests only):
 * Pairing::calc_g1_g2 pairing.rs:54, calc_g2_g1 pairing.rs:55, weil = calc_g1_g2(P,Q) * calc_g2_g1(Q,P)^-1 pairing.rs:75-84 */
int zkt_miller_g1g2_batch(const zkt_g1_affine* g1, const zkt_g2_affine* g2, uint64_t* out_fq12, size_t n);
int zkt_miller_g2g1_batch(const zkt_g2_affine* g2, const zkt_g1_affine* g1, uint64_t* out_fq12, size_t n);
int zkt_weil_batch(const zkt_g1_affine* g1, const zkt_g2_affine* g2, uint64_t* out_fq12, size_t n);
/* Diagnostic (no reference counterpart): `count` independent runs of the self-test program of the lazily reduced Fq arithmetic
 * (csrc/fq_program.h: a pseudo-random straight-line program of field operations over four registers, seeds seed0 + i), one per lane.
 * in/out: count x 4 canonical Fq; violations[i] = limb/size invariant violations seen by run i.  The test-suite replays the program
 * in python integers (tests/test_hostcheck.py for the host build of the same header, tests/test_gpu_parity.py for the device). */
int zkt_selftest_fq_program(uint64_t seed0, int steps, const uint64_t* in, uint64_t* out, int32_t* violations, size_t count);
/* Diagnostic (no reference counterpart): Fq12 operations computed on the lane-distributed form of the small-batch pairing (csrc/zkt_dpairing.hip),
 * so that its building blocks can be compared with zkt_fq12_*_batch: op 0 a*b, 1 a^2, 2 a^q, 3 a^(q^2), 4 a^(q^6) (conjugate), 5 1/a,
 * 6 a^2 by the Granger-Scott formulas (a in the cyclotomic subgroup). */
int zkt_debug_dfq12_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
/* a15: GTPoint == gt_point.rs:33-39 (all 12 coefficients); returns 1/0, or <0 = -status */
int zkt_gt_eq(const uint64_t* a_fq12, const uint64_t* b_fq12);

/* a17: Groth16 over the kernels above — CRS (crs.rs:17-43), CRS::new crs.rs:49-146, Prover::prove prover.rs:96-147,
 * Verifier::verify verifier.rs:30-54.  The reference samples alpha,beta,gamma,delta,x (crs.rs:59-63) and r,s
 * (prover.rs:100-101) from OS entropy; here they are arguments (4-limb Fr, non-zero).  QAP polynomials ui/vi/wi are
 * (m+1) x n dense Fr coefficient arrays, low degree first (Prover.ui/vi/wi prover.rs:44-46); wires = a_0..a_m
 * (wires.rs:12-39), statement = a_0..a_l; h = quotient polynomial coefficients (prover.rs:64-71), h_len <= n. */
typedef struct {
  size_t n, l, m;                       /* constraints, last statement wire, last wire (prover.rs:36-38) */
  zkt_g1_affine *g1_alpha, *g1_beta, *g1_delta, *g1_xi /*n*/, *g1_uvw_stmt /*l+1*/, *g1_uvw_wit /*m-l*/, *g1_xt_by_delta /*n*/;
  zkt_g2_affine *g2_beta, *g2_gamma, *g2_delta, *g2_xi /*n*/;
  uint64_t* gt_alpha_beta;              /* 72 limbs */
} zkt_groth16_crs;
int zkt_groth16_setup(zkt_groth16_crs* crs, const uint64_t* ui, const uint64_t* vi, const uint64_t* wi,
                      const uint64_t* alpha, const uint64_t* beta, const uint64_t* gamma, const uint64_t* delta, const uint64_t* x);
int zkt_groth16_prove(const zkt_groth16_crs* crs, const uint64_t* ui, const uint64_t* vi, const uint64_t* wires,
                      const uint64_t* h, size_t h_len, const uint64_t* r, const uint64_t* s,
                      zkt_g1_affine* A, zkt_g2_affine* B, zkt_g1_affine* C);
/* f-2: many proofs against one CRS, one proof per lane; the three pairings of a proof share one Miller squaring chain and
 * one final exponentiation (the decision is a bool, so this is parity-safe).  stmt_wires: n_proofs x n_stmt.  ok[i] = 1/0. */
int zkt_groth16_verify_batch(const zkt_groth16_crs* crs, const zkt_g1_affine* A, const zkt_g2_affine* B, const zkt_g1_affine* C,
                             const uint64_t* stmt_wires, size_t n_stmt, size_t n_proofs, uint32_t* ok);
/* Optional, once per verifying key: build the per-key tables of the verification fast path now and wait for them (statement points' fixed-base tables, line tables of
 * gamma and delta, the ate counterpart of alpha_beta: ~5 ms).  Without it the library builds them at the FIRST small-batch verification against a key, beside that call
 * (which is served by kernels that need nothing of the key: ~10.5 ms instead of ~5); every later call on the key takes ~5 ms.  Keys are cached by their bytes (last
 * four).  Decisions do not depend on whether this was called. */
int zkt_groth16_vk_prepare(const zkt_groth16_crs* crs, size_t n_stmt);
/* returns 1 accept, 0 reject, negative = -status (a pairing argument at infinity panics in the reference) */
int zkt_groth16_verify(const zkt_groth16_crs* crs, const zkt_g1_affine* A, const zkt_g2_affine* B, const zkt_g1_affine* C,
                       const uint64_t* stmt_wires, size_t n_stmt);

/* f-4: equalities of pairing products as the reference's callers test them (lhs == rhs on GTPoints: signature.rs:34-39,
 * pinocchio/verifier.rs:43-84).  For each of n elements: prod_{j<k} tate(+-g1[i*k+j], g2[i*k+j]) == 1, k <= 4; negate[j] != 0
 * negates slot j's G1 point (e(-P,Q) = e(P,Q)^-1), so e(P1,Q1) == e(P2,Q2) e(P3,Q3) is k = 3, negate = {0,1,1}.  The k Miller
 * loops share one squaring chain and one final exponentiation (the decision is a bool, so this is parity-safe).  ok[i] = 1/0;
 * ZKT_ERR_INFINITY (+index) if an argument is the point at infinity (the reference's tate() panics).
 * Shared by every verification entry point (this one, zkt_groth16_verify*, zkt_bls_verify_batch, zkt_pinocchio_verify): the rewriting of
 * lhs == rhs as a product == 1 uses e(-P,Q) = e(P,Q)^-1, which holds for P of order r on the curve.  An element with a G1 argument outside the
 * order-r subgroup, or with a point off its curve, is detected and evaluated the reference's way instead: every pairing of both sides through the
 * reference's own Miller chain and the exact final exponentiation, slots with negate = 0 on the left, the others (un-negated) on the right, Fq12
 * equality (verifier.rs:36-53, signature.rs:34-39) — the reference's accept / reject, or ZKT_ERR_INFINITY (+index) where its tate() panics. */
int zkt_pairing_product_check_batch(const zkt_g1_affine* g1, const zkt_g2_affine* g2, const uint8_t* negate, size_t k, size_t n, uint32_t* ok);
/* on != 0: such elements are REJECTED (ok = 0) without the reference's evaluation — the behaviour of earlier versions of this library, for hosts that
 * treat a point outside its group as an attack and do not want to spend ~0.1 s of one lane on it.  Process-wide; also ZKT_VERIFY_FAIL_CLOSED=1. */
void zkt_verify_set_fail_closed(int on);
/* f-4: BLS signatures, Signer signature.rs:8-40.  Messages are n byte strings, concatenated, offsets[n+1].
 * G2Point::hash_to_g2point g2_point.rs:84-88 (the bytes as a big-endian integer, reduced mod r, times the G2 generator);
 * sign = hash * sk (signature.rs:28-31, computed as generator * (h * sk mod r): the same group element; sk = 4-limb PrivateKey.value, private_key.rs:10-27; gen_public_key is
 * zkt_bls_public_keys_batch, or zkt_g1_mul_batch of the generator); verify = tate(g1, sig) == tate(pk, hash) (signature.rs:34-39), one signature per lane. */
int zkt_bls_hash_to_g2_batch(const uint8_t* msgs, const uint64_t* offsets, size_t n, zkt_g2_affine* out);
/* Signer::gen_public_key signature.rs:24-27 for n private keys (4 limbs each): pks[i] = G1 generator * sks[i], through the generator's comb table */
int zkt_bls_public_keys_batch(const uint64_t* sks, size_t n, zkt_g1_affine* pks);
int zkt_bls_sign_batch(const uint8_t* msgs, const uint64_t* offsets, const uint64_t* sks, size_t n, zkt_g2_affine* sigs);
int zkt_bls_verify_batch(const uint8_t* msgs, const uint64_t* offsets, const zkt_g2_affine* sigs, const zkt_g1_affine* pks, size_t n, uint32_t* ok);

/* f-4: Pinocchio (protocol 2 of eprint 2013/279) — CRS (EvaluationKeys crs.rs:12-22, VerificationKeys crs.rs:24-39), CRS::new crs.rs:49-161,
 * Prover::prove pinocchio/prover.rs:98-170, Verifier::verify pinocchio/verifier.rs:31-85, Proof proof.rs:6-17.  vi/wi/yi are the
 * (n_io + n_mid) x n dense Fr coefficient arrays of Prover.vi/wi/yi (prover.rs:43-45), low degree first; wires 0..n_io-1 are
 * Witness::io() (witness.rs:20-23, the constant one included), the rest Witness::mid() (witness.rs:25-27); max_degree as
 * prover.rs:68-78.  rnd = r_v, r_w, alpha_v, alpha_w, alpha_y, beta, gamma, s (crs.rs:58-64,82), 4 limbs each, non-zero;
 * delta_v, delta_y = prover.rs:104-105; h = coefficients of p / t (prover.rs:143-146), h_len <= max_degree. */
typedef struct {
  size_t n, n_io, n_mid, max_degree;
  zkt_g1_affine *vk_mid /*n_mid*/, *g1_wk_mid; zkt_g2_affine* g2_wk_mid; zkt_g1_affine *yk_mid, *alpha_vk_mid, *alpha_wk_mid, *alpha_yk_mid;
  zkt_g2_affine* si /*max_degree*/; zkt_g1_affine* beta_vwy_k_mid;
  zkt_g1_affine* one_g1; zkt_g2_affine *one_g2, *alpha_v; zkt_g1_affine* alpha_w; zkt_g2_affine *alpha_y, *gamma, *beta_gamma; zkt_g1_affine* t;
  zkt_g1_affine* vk_io /*n_io*/; zkt_g2_affine* wk_io; zkt_g1_affine *yk_io, *alpha_v_t, *alpha_y_t, *beta_t;
} zkt_pinocchio_crs;
typedef struct {
  zkt_g1_affine *v_mid_s, *g1_w_mid_s; zkt_g2_affine* g2_w_mid_s; zkt_g1_affine* y_mid_s; zkt_g2_affine* h_s;
  zkt_g1_affine *alpha_v_mid_s, *alpha_w_mid_s, *alpha_y_mid_s, *beta_vwy_mid_s;
} zkt_pinocchio_proof;
int zkt_pinocchio_setup(zkt_pinocchio_crs* crs, const uint64_t* vi, const uint64_t* wi, const uint64_t* yi, const uint64_t* rnd);
int zkt_pinocchio_prove(const zkt_pinocchio_crs* crs, const uint64_t* wires, const uint64_t* h, size_t h_len,
                        const uint64_t* delta_v, const uint64_t* delta_y, zkt_pinocchio_proof* proof);
/* The prover with the evaluation key resident (the reference keeps its EvaluationKeys in the CRS for every prove call, pinocchio/prover.rs:98-103): the ten base
 * sets of prover.rs:133-156 stay in HBM with their window-multiple tables, a proof uploads the wire values once and runs its ten multi-scalar multiplications
 * through the pipelined resident interface.  Same arguments and the same nine proof points as zkt_pinocchio_prove; a handle serves one proof at a time. */
typedef struct zkt_pinocchio_pk zkt_pinocchio_pk;
int zkt_pinocchio_pk_create(const zkt_pinocchio_crs* crs, zkt_pinocchio_pk** out);
int zkt_pinocchio_prove_resident(zkt_pinocchio_pk* pk, const uint64_t* wires, const uint64_t* h, size_t h_len,
                                 const uint64_t* delta_v, const uint64_t* delta_y, zkt_pinocchio_proof* proof);
void zkt_pinocchio_pk_free(zkt_pinocchio_pk* pk);
/* 1 accept, 0 reject, negative = -status (a pairing argument at infinity panics in the reference); the five equalities are
 * decided in the reference's order (a rejection by an earlier one wins over a panic of a later one).  Up to 12 io wires: fixed-base
 * tables of the key's io points are built on first sight of a key (~30 ms) and kept for the last two keys, a verification then takes
 * ~6.5 ms; calls are serialised inside the library. */
int zkt_pinocchio_verify(const zkt_pinocchio_crs* crs, const zkt_pinocchio_proof* proof, const uint64_t* io_wires);

/* a18: Bulletproofs::inner_product_argument bulletproofs.rs:19-55 over secp256k1; n a power of two; a, b are 4-limb
 * residues mod the group order; xs = one challenge per level (the reference draws them at bulletproofs.rs:42).
 * out_trace (optional): per level {L, R, P'}.  Returns 1/0 like the reference's bool, negative = -status.
 * Without a trace only the bool is observable and the argument collapses to ONE multi-scalar multiplication over [gg | hh | u] (L_j, R_j are never formed). */
int zkt_bp_inner_product_argument(size_t n, const zkt_secp_affine* gg, const zkt_secp_affine* hh, const zkt_secp_affine* u,
                                  const zkt_secp_affine* P, const uint64_t* a, const uint64_t* b, const uint64_t* xs,
                                  zkt_secp_affine* out_trace);
/* The same argument with the generators resident: gg, hh, u are fixed for a deployment (the reference rebuilds nothing either — its callers
 * pass the same AffinePoints every time, bulletproofs.rs:139), so their window-multiple table and the work buffers are built once.
 * A context serves one call at a time; results are identical to zkt_bp_inner_product_argument.  gg, hh, u, P, a, b may be host or device
 * pointers here (xs: host). */
typedef struct zkt_bp_ipa_ctx zkt_bp_ipa_ctx;
int zkt_bp_ipa_ctx_create(size_t n, const zkt_secp_affine* gg, const zkt_secp_affine* hh, const zkt_secp_affine* u, zkt_bp_ipa_ctx** out);
void zkt_bp_ipa_ctx_free(zkt_bp_ipa_ctx* ctx);
int zkt_bp_inner_product_argument_ctx(zkt_bp_ipa_ctx* ctx, const zkt_secp_affine* P, const uint64_t* a, const uint64_t* b, const uint64_t* xs,
                                      zkt_secp_affine* out_trace);

/* a18: Bulletproofs::range_proof bulletproofs.rs:58-147 (n = bit length, a power of two; aL = the value's bits).  rnd =
 * alpha, rho, y, z, tau1, tau2, x, sL[n], sR[n] (the values the reference draws at :76,:79-81,:84-85,:97-98,:102);
 * u = the random point of :137 and xs the inner-product challenges (only with use_ipa).  out_pts (optional) = A,S,T1,T2,P.
 * The context built for gg, hh, u (window-multiple table, work buffers: ~20 ms at 65,536 generators) is kept after the call and reused by the next
 * one-shot call that brings the same generators byte for byte (host pointers; ZKT_BP_CTX_CACHE=0 disables); zkt_bp_inner_product_argument likewise. */
int zkt_bp_range_proof(size_t n, const zkt_secp_affine* V, const uint64_t* aL, const uint64_t* gamma, const zkt_secp_affine* g,
                       const zkt_secp_affine* h, const zkt_secp_affine* gg, const zkt_secp_affine* hh, int use_ipa,
                       const uint64_t* rnd, const zkt_secp_affine* u, const uint64_t* xs, zkt_secp_affine* out_pts);

/* The same range proof over a context's resident generators (gg, hh, u of zkt_bp_ipa_ctx_create; n = the context's size): no table build and no
 * generator upload per proof.  Results are identical to zkt_bp_range_proof. */
int zkt_bp_range_proof_ctx(zkt_bp_ipa_ctx* ctx, const zkt_secp_affine* V, const uint64_t* aL, const uint64_t* gamma, const zkt_secp_affine* g,
                           const zkt_secp_affine* h, int use_ipa, const uint64_t* rnd, const uint64_t* xs, zkt_secp_affine* out_pts);

/* ---- device-resident entry points (inputs/outputs already in HBM) -------------------- */
/* Bases kept on the device in kernel layout (internal limb form, x and y, 112 B per G1 point) — the analogue of
 * a CRS that is uploaded once (crs.rs:85-135) and reused by every prove call. */
typedef struct zkt_g1_bases zkt_g1_bases;
int zkt_g1_bases_upload(const zkt_g1_affine* host_bases, size_t n, zkt_g1_bases** out);
int zkt_g1_bases_from_device(const zkt_g1_affine* dev_bases, size_t n, void* stream, zkt_g1_bases** out);
size_t zkt_g1_bases_len(const zkt_g1_bases* b);
void zkt_g1_bases_free(zkt_g1_bases* b);
/* MSM over resident bases and DEVICE scalars (4 limbs each).  Writes the affine sum to
 * host `out` (blocking) — and, if `dev_partial_jac` is non-NULL, the un-normalised
 * Jacobian partial sum (ZKT_G1_PARTIAL_WORDS u32 words: X,Y,Z in the engine's internal
 * limb form, opaque to the caller) to that DEVICE buffer for a
 * multi-GPU combine (see zkt_g1_jac_sum_dev). */
int zkt_g1_msm_dev(const zkt_g1_bases* bases, const uint64_t* dev_scalars, size_t n, void* stream,
                   zkt_g1_affine* out, uint32_t* dev_partial_jac);
/* Pipelined form: up to ZKT_MSM_SLOTS (8) MSMs over the same bases in flight, each on its own internal
 * stream and workspace, so the digit sort of one, the bucket accumulation of the next and the
 * latency-bound reduction tail of a third overlap on the chip.  submit() orders the slot behind
 * everything already enqueued on `stream` (where the scalars are produced) and returns at once;
 * collect() blocks until that slot's affine result (and optional Jacobian partial) is available. */
#define ZKT_MSM_SLOTS 8
/* u32 words of one opaque Jacobian partial sum (3 internal coordinates) per group */
#define ZKT_G1_PARTIAL_WORDS 42
#define ZKT_G2_PARTIAL_WORDS 84
#define ZKT_SECP_PARTIAL_WORDS 24
int zkt_g1_msm_submit(zkt_g1_bases* bases, const uint64_t* dev_scalars, size_t n, void* stream, int slot);
int zkt_g1_msm_collect(zkt_g1_bases* bases, int slot, zkt_g1_affine* out, uint32_t* dev_partial_jac);
/* combine step of a sharded MSM: sum `count` Jacobian partials (ZKT_G1_PARTIAL_WORDS u32 words each, device)
 * and normalise to affine on the host */
int zkt_g1_jac_sum_dev(const uint32_t* dev_partials, size_t count, void* stream, zkt_g1_affine* out);
/* bytes of device workspace a zkt_g1_msm_dev of n terms allocates once and keeps */
size_t zkt_g1_msm_workspace_bytes(size_t n);

/* the same resident-bases / pipelined MSM interface for G2 and secp256k1 (Jacobian partial = ZKT_G2_PARTIAL_WORDS / ZKT_SECP_PARTIAL_WORDS u32 words) */
typedef struct zkt_g2_bases zkt_g2_bases;
typedef struct zkt_secp_bases zkt_secp_bases;
int zkt_g2_bases_upload(const zkt_g2_affine* host_bases, size_t n, zkt_g2_bases** out);
int zkt_g2_bases_from_device(const zkt_g2_affine* dev_bases, size_t n, void* stream, zkt_g2_bases** out);
size_t zkt_g2_bases_len(const zkt_g2_bases* b);
void zkt_g2_bases_free(zkt_g2_bases* b);
int zkt_g2_msm_dev(const zkt_g2_bases* bases, const uint64_t* dev_scalars, size_t n, void* stream, zkt_g2_affine* out, uint32_t* dev_partial_jac);
int zkt_g2_msm_submit(zkt_g2_bases* bases, const uint64_t* dev_scalars, size_t n, void* stream, int slot);
int zkt_g2_msm_collect(zkt_g2_bases* bases, int slot, zkt_g2_affine* out, uint32_t* dev_partial_jac);
int zkt_g2_jac_sum_dev(const uint32_t* dev_partials, size_t count, void* stream, zkt_g2_affine* out);
int zkt_secp_bases_upload(const zkt_secp_affine* host_bases, size_t n, zkt_secp_bases** out);
int zkt_secp_bases_from_device(const zkt_secp_affine* dev_bases, size_t n, void* stream, zkt_secp_bases** out);
size_t zkt_secp_bases_len(const zkt_secp_bases* b);
void zkt_secp_bases_free(zkt_secp_bases* b);
int zkt_secp_msm_dev(const zkt_secp_bases* bases, const uint64_t* dev_scalars, size_t n, void* stream, zkt_secp_affine* out, uint32_t* dev_partial_jac);
int zkt_secp_msm_submit(zkt_secp_bases* bases, const uint64_t* dev_scalars, size_t n, void* stream, int slot);
int zkt_secp_msm_collect(zkt_secp_bases* bases, int slot, zkt_secp_affine* out, uint32_t* dev_partial_jac);
int zkt_secp_jac_sum_dev(const uint32_t* dev_partials, size_t count, void* stream, zkt_secp_affine* out);

/* f-3: Groth16 at scale on the reference's evaluation domain {1..n}.  The reference keeps (m+1) dense interpolated
 * polynomials per matrix (QAP::build, qap/qap.rs:137-226; Prover.ui/vi/wi prover.rs:44-46) and the prover works in
 * coefficient form (prover.rs:64-71,103-131) — O(m n) group operations, infeasible at 2^20 constraints.  These entry
 * points take the R1CS itself, one sparse row per constraint (R1CS.constraints, r1cs.rs; Constraint{a,b,c},
 * constraint.rs:5-9; SparseVec), and produce the SAME proof points: setup derives device-resident Lagrange-basis
 * bases from the trapdoor, prove is 3 sparse mat-vecs + Fr NTTs + three MSMs whose outputs are A, B, C.
 * rowptr: n+1 offsets into col/val; col: wire index 0..m; val: 4-limb canonical Fr. */
typedef struct { const uint64_t* rowptr; const uint32_t* col; const uint64_t* val; } zkt_sparse_rows;
typedef struct zkt_groth16_pk zkt_groth16_pk;
/* CRS::new (crs.rs:49-146) with injected trapdoors.  Fills the verifying part of `vk` (g1_alpha, g1_beta, g1_delta,
 * g1_uvw_stmt[l+1], g2_beta, g2_gamma, g2_delta, gt_alpha_beta — and g1_uvw_wit[m-l] if that pointer is non-NULL;
 * g1_xi, g1_xt_by_delta, g2_xi are not produced) and returns the proving key, resident on the device.
 * ZKT_ERR_INV_ZERO if x lies in {1..2n-1} (the reference would still build a CRS for x in n+1..2n-1; a uniform x never does). */
int zkt_groth16_setup_r1cs(size_t n, size_t l, size_t m, const zkt_sparse_rows* A, const zkt_sparse_rows* B, const zkt_sparse_rows* C,
                           const uint64_t* alpha, const uint64_t* beta, const uint64_t* gamma, const uint64_t* delta, const uint64_t* x,
                           zkt_groth16_crs* vk, zkt_groth16_pk** out);
/* Prover::prove (prover.rs:96-147), r and s injected; wires = a_0..a_m, 4-limb canonical Fr on the host. */
int zkt_groth16_prove_r1cs(zkt_groth16_pk* pk, const uint64_t* wires, const uint64_t* r, const uint64_t* s,
                           zkt_g1_affine* A, zkt_g2_affine* B, zkt_g1_affine* C);
/* the same with the wires already in HBM (device pointer) */
int zkt_groth16_prove_r1cs_dev(zkt_groth16_pk* pk, const uint64_t* dev_wires, const uint64_t* r, const uint64_t* s,
                               zkt_g1_affine* A, zkt_g2_affine* B, zkt_g1_affine* C);
/* Pipelined form: two proofs in flight on one (unsharded) key, the Fr stage of the next proof under the MSMs of the current one.
 * slot is 0 or 1; collect a slot before submitting it again. */
int zkt_groth16_prove_r1cs_submit(zkt_groth16_pk* pk, int slot, const uint64_t* dev_wires, const uint64_t* r, const uint64_t* s);
int zkt_groth16_prove_r1cs_collect(zkt_groth16_pk* pk, int slot, zkt_g1_affine* A, zkt_g2_affine* B, zkt_g1_affine* C);
/* Multi-GPU form (BASELINE config 4): rank `shard` of `nshards` keeps a contiguous index range of each resident base set.  A proof is
 * then zkt_groth16_prove_r1cs_partials on every rank (dev_partials: ZKT_GROTH16_PARTIAL_WORDS u32 = the Jacobian partials of A, B, C),
 * an all_gather of those, and zkt_g1_jac_sum_dev / zkt_g2_jac_sum_dev / zkt_g1_jac_sum_dev over the gathered A, B, C columns
 * (each made contiguous: count x PARTIAL_WORDS). */
#define ZKT_GROTH16_PARTIAL_WORDS (2 * ZKT_G1_PARTIAL_WORDS + ZKT_G2_PARTIAL_WORDS)
int zkt_groth16_setup_r1cs_sharded(size_t n, size_t l, size_t m, const zkt_sparse_rows* A, const zkt_sparse_rows* B, const zkt_sparse_rows* C,
                                   const uint64_t* alpha, const uint64_t* beta, const uint64_t* gamma, const uint64_t* delta, const uint64_t* x,
                                   size_t shard, size_t nshards, zkt_groth16_crs* vk, zkt_groth16_pk** out);
int zkt_groth16_prove_r1cs_partials(zkt_groth16_pk* pk, const uint64_t* dev_wires, const uint64_t* r, const uint64_t* s, uint32_t* dev_partials);
void zkt_groth16_pk_free(zkt_groth16_pk* pk);

/* ---- multi-GPU: one process per GPU -------------------------------------------------------------------------------
 * The index range of an MSM (or of the three resident base sets of a Groth16 key) is partitioned over the ranks; every rank computes the
 * Jacobian partial sum of its shard; the only exchange is ONE all-gather of the fixed-size partials (168 B G1, 336 B G2, 672 B per proof)
 * and a local combine on every rank — an elliptic-curve sum is not an RCCL reduction op.  Every rank returns the same affine result.
 * Transport: RCCL over xGMI (zkt_comm_unique_id on rank 0, the id shipped by the host like ncclGetUniqueId's, zkt_comm_init on every
 * rank after zkt_init), or a host callback for hosts that bring their own exchange (MPI, gloo, tests): fn all-gathers bytes_per_rank
 * bytes from every rank's `send` into `recv` (rank-major) and returns 0.  The callback runs inside a collective entry point: it may call
 * zkt_comm_rank / zkt_comm_world (lock-free), but no other collective entry point.  zkt_comm_init with world == 1 needs no RCCL library (the
 * exchange is then a device-to-device copy; with the library present a one-rank communicator runs the same calls as an 8-rank one). */
#define ZKT_COMM_ID_BYTES 128
typedef int (*zkt_allgather_fn)(void* ctx, const void* send, void* recv, size_t bytes_per_rank);
int zkt_comm_unique_id(uint8_t id[ZKT_COMM_ID_BYTES]);
int zkt_comm_init(int rank, int world, const uint8_t id[ZKT_COMM_ID_BYTES]);     /* world == 1: id may be NULL */
int zkt_comm_init_callback(int rank, int world, zkt_allgather_fn fn, void* ctx);
void zkt_comm_finalize(void);
int zkt_comm_rank(void);                                                        /* -1 before zkt_comm_init */
int zkt_comm_world(void);                                                       /* 0 before zkt_comm_init */
/* contiguous, balanced index range [lo, hi) of `rank` (the first n % world ranks hold one extra term) */
void zkt_comm_shard_range(size_t n, int rank, int world, size_t* lo, size_t* hi);
/* Polynomial::eval_with_g1_hidings (polynomial.rs:271-281) with the terms partitioned over the ranks: `bases` holds THIS rank's shard,
 * dev_scalars its n_local scalars.  Blocking; collective (every rank must call it). */
int zkt_g1_msm_sharded(zkt_g1_bases* bases, const uint64_t* dev_scalars, size_t n_local, void* stream, zkt_g1_affine* out);
int zkt_g2_msm_sharded(zkt_g2_bases* bases, const uint64_t* dev_scalars, size_t n_local, void* stream, zkt_g2_affine* out);
int zkt_secp_msm_sharded(zkt_secp_bases* bases, const uint64_t* dev_scalars, size_t n_local, void* stream, zkt_secp_affine* out);
/* pipelined form: zkt_*_msm_submit on every rank, then this instead of zkt_*_msm_collect (collective, same slot order on every rank) */
int zkt_g1_msm_sharded_collect(zkt_g1_bases* bases, int slot, zkt_g1_affine* out);
int zkt_g2_msm_sharded_collect(zkt_g2_bases* bases, int slot, zkt_g2_affine* out);
int zkt_secp_msm_sharded_collect(zkt_secp_bases* bases, int slot, zkt_secp_affine* out);
/* Prover::prove (prover.rs:96-147) for ONE proof sharded over the ranks (BASELINE config 4): pk from zkt_groth16_setup_r1cs_sharded(…,
 * zkt_comm_rank(), zkt_comm_world(), …).  The mat-vecs are replicated; of the quotient every rank evaluates its own range only (no exchange); collective. */
int zkt_groth16_prove_r1cs_sharded(zkt_groth16_pk* pk, const uint64_t* dev_wires, const uint64_t* r, const uint64_t* s,
                                   zkt_g1_affine* A, zkt_g2_affine* B, zkt_g1_affine* C);

int zkt_g1_mul_batch_dev(const zkt_g1_affine* dev_points, const uint64_t* dev_scalars, int scalar_limbs,
                         zkt_g1_affine* dev_out, size_t n, void* stream);
int zkt_g2_mul_batch_dev(const zkt_g2_affine* dev_points, const uint64_t* dev_scalars, int scalar_limbs,
                         zkt_g2_affine* dev_out, size_t n, void* stream);
/* BLOCKING, unlike the other *_dev entry points: the pairings run on the library's one staging stream, ordered behind `stream` by an event, and the call
 * returns when the result is complete (work queued afterwards on any stream sees it).  Calls are serialised against each other and against the
 * other staged entry points; the library spends no per-caller-stream scratch on them ("Device-resident use"). */
int zkt_tate_batch_dev(const zkt_g1_affine* dev_g1, const zkt_g2_affine* dev_g2, uint64_t* dev_out_fq12,
                       size_t n, void* stream);
int zkt_fq_mul_batch_dev(const uint64_t* dev_a, const uint64_t* dev_b, uint64_t* dev_out, size_t n, void* stream);

/* timing hook for bench.py: device time (ms, HIP events on `stream`) of the dominant kernel
 * of the last zkt_g1_msm_dev / zkt_tate_batch_dev call on this thread, and its name */
float zkt_last_kernel_ms(void);
const char* zkt_last_kernel_name(void);

#ifdef __cplusplus
}
#endif
#endif /* ZKT_H */
