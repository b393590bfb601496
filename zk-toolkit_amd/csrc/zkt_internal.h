// Internal launch interface between the C ABI (zkt_api.cpp) and the HIP kernel files.
// All pointers are DEVICE pointers in the include/zkt.h layouts, viewed as u32 words.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace zkt {

enum FieldId { F_FQ = 0, F_FR = 1, F_SP = 2, F_SN = 3 };
enum FpOp { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2, OP_SQR = 3, OP_NEG = 4, OP_INV = 5, OP_CUBE = 6 };
enum TowerOp { T_ADD = 0, T_SUB = 1, T_MUL = 2, T_INV = 3, T_NEG = 4, T_REDUCE = 5 };
enum GroupId { G_G1 = 0, G_G2 = 1, G_SECP = 2 };

// error word: index of the first offending element, or ~0ull
static constexpr unsigned long long NO_ERR = ~0ull;

hipError_t launch_fp_op(int field, int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n,
                        unsigned long long* err, hipStream_t s);
// pow (prime_field_elem.rs:311-328): out[i] = a[i]^e_i, exponents of e_words u32 words each (shared: one exponent for all elements)
hipError_t launch_fp_pow(int field, const uint32_t* a, const uint32_t* e, int e_words, bool shared, uint32_t* out, size_t n, hipStream_t s);
// pow_seq (prime_field_elem.rs:346-361): out[i] = base^i for i < n;  repeat (:363-376): out[i] = base
hipError_t launch_fp_pow_seq(int field, const uint32_t* base, uint32_t* out, size_t n, bool repeat, hipStream_t s);
// PrimeFieldElems::sum (prime_field_elems.rs:35-41) and PrimeFieldElems * PrimeFieldElem (:152-175); parts: fp_sum_scratch_elems() elements of scratch
size_t fp_sum_scratch_elems();
hipError_t launch_fp_sum(int field, const uint32_t* a, size_t n, uint32_t* out, uint32_t* parts, hipStream_t s);
hipError_t launch_fp_scale(int field, const uint32_t* a, const uint32_t* k, uint32_t* out, size_t n, hipStream_t s);
hipError_t launch_selftest_fq_program(unsigned long long seed0, int steps, const uint32_t* in4, uint32_t* out4, int* bad, size_t count, hipStream_t s);
hipError_t launch_tower_op(int deg, int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n,
                           unsigned long long* err, hipStream_t s);
hipError_t launch_fq12_pow(const uint32_t* a, const uint32_t* exp_dev, int exp_nlimbs, uint32_t* out, size_t n, hipStream_t s);
hipError_t launch_group_add(int grp, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n, hipStream_t s);
hipError_t launch_group_neg(int grp, const uint32_t* a, uint32_t* out, size_t n, hipStream_t s);
hipError_t launch_group_mul(int grp, const uint32_t* pts, const uint32_t* scalars, int scalar_words, uint32_t* out, size_t n, hipStream_t s, bool fixed_base = false, bool fixed_scalar = false);
// pred 0: on the curve (is_rational_point); pred 1: order * P == infinity (order: little-endian u32 words on the device)
// out[k * n + i] = point i of array k lies on E and in G1 (or is the point at infinity); K <= 4 arrays, `stride` in u32 words
struct G1Fits { const uint32_t* pts[4]; uint32_t stride[4]; };
hipError_t launch_g1_fits(const G1Fits& f, int K, uint32_t* out, size_t n, hipStream_t s);

hipError_t launch_group_pred(int grp, int pred, const uint32_t* pts, const uint32_t* order, int order_words, uint32_t* out, size_t n, hipStream_t s);
hipError_t launch_group_sum_inplace(int grp, uint32_t* pts, size_t n, hipStream_t s);
// several independent batched scalar multiplications in one launch (strides in u32 words, 0 = broadcast one point / one scalar)
struct MulSeg { const uint32_t* pts; const uint32_t* k; uint32_t* out; uint32_t count, pt_stride, k_stride; };
struct MulSegs { MulSeg s[16]; int n; };
hipError_t launch_group_mul_segs(int grp, const MulSegs& segs, int scalar_words, hipStream_t s);
// A short straight-line program over scalar-field elements (canonical 8-word residues in device memory), run by ONE lane in ONE launch: the scalar
// algebra between the vector kernels of a protocol is a chain of dependent one-element operations, and a launch per operation costs ~50 us each.
struct ScalarOp { int op; const uint32_t* a; const uint32_t* b; uint32_t* out; };      // op: OP_ADD / OP_SUB / OP_MUL / OP_NEG / OP_INV (b unused for the last two)
struct ScalarOps { ScalarOp o[48]; int n; };
hipError_t launch_scalar_ops(int field, const ScalarOps& ops, unsigned long long* err, hipStream_t s);
// out[j] = sum of up to 4 affine points in[j][*] (complete addition, one lane per output, one normalisation each): the tail of a protocol adds a handful
// of single points in a fixed pattern, and a launch per addition (each with its own inversion) is ~150 us
struct PointSums { const uint32_t* in[12][4]; int cnt[12]; uint32_t* out[12]; int n; };
hipError_t launch_point_sums(int grp, const PointSums& p, hipStream_t s);
// Fixed-base products: table[w] = 16^w P (64 affine points, one launch of one wave per point, ~4 ms once) and then k P = sum_w digit_w(k) table[w]
// by ONE WAVE per product — lane w takes the w-th 4-bit digit of the 256-bit scalar, a tree adds the 64 partial products (~0.2 ms instead of the
// ~4-5 ms of a 255-step double-and-add in one lane).  The Bulletproofs range proof multiplies the same g, h, u a dozen times per proof.
struct FixedMul { const uint32_t* table; const uint32_t* k; uint32_t* out; };
struct FixedMuls { FixedMul m[16]; int n; };
struct FixedTables { const uint32_t* point[12]; uint32_t* table[12]; int n; };     // up to twelve points per launch, one wave each, side by side
hipError_t launch_fixed_tables(int grp, const FixedTables& t, hipStream_t s);
// out[i] = k[i] * G for the group's standard generator (G_G1 / G_G2; gen_abi = that generator on the device, read on first use only): a comb table of
// 960 multiples built once per process, at most 64 mixed additions per product instead of a 255-step double-and-add.  k: 8 words per scalar.
hipError_t launch_generator_mul(int grp, const uint32_t* gen_abi, const uint32_t* k, uint32_t* out, size_t n, hipStream_t s);
hipError_t launch_fixed_muls(int grp, const FixedMuls& f, hipStream_t s);
// the same product for a whole batch: out[(j * n + i)] = k[(i * n_pts + j)] * P_j from tables[j] (64 points each), one wave per product, grid = n * n_pts
hipError_t launch_fixed_muls_batch(int grp, const uint32_t* tables, const uint32_t* k, uint32_t* out, size_t n, int n_pts, hipStream_t s);
hipError_t launch_miller_exact(int which, const uint32_t* g1, const uint32_t* g2, uint32_t* out, size_t n, unsigned long long* err, hipStream_t s);
hipError_t launch_groth16_verify(const uint32_t* A, const uint32_t* B, const uint32_t* C, const uint32_t* uvw_stmt, const uint32_t* stmt, int n_stmt,
                                 const uint32_t* gamma, const uint32_t* delta, const uint32_t* alpha_beta, uint32_t* ok, size_t n,
                                 unsigned long long* err, hipStream_t s, const uint32_t* ate_key = nullptr);
// what a verifying key contributes to the 63-step loop (line tables of gamma and delta, the ate counterpart of alpha_beta, verdicts on its points); layout at k_ate_key_prep
static constexpr size_t ATE_KEY_WORDS = 2 * (size_t)68 * 84 + 144 + 1;
hipError_t launch_ate_key_prep(const uint32_t* alpha, const uint32_t* beta, const uint32_t* gamma, const uint32_t* delta, const uint32_t* uvw_stmt, int n_stmt,
                               uint32_t* key, hipStream_t s);
// 8-bit window tables of a key's statement points and the statement sums of a large batch from them (zkt_group.hip)
size_t stmt_wide_table_words(int n_pts);
hipError_t launch_stmt_wide_tables(const uint32_t* tab16, int n_pts, uint32_t* tables, hipStream_t s);
hipError_t launch_stmt_sums_wide(const uint32_t* tables, const uint32_t* stmt, int n_stmt, uint32_t* out, size_t n, hipStream_t s);
// elements with a G1 argument outside the order-r subgroup (or a point off its curve): 0 = both sides evaluated the reference's way (default), 1 = rejected
void verify_set_fail_closed(int on);
int verify_fail_closed();
size_t dproduct_limit();      // elements x pairs up to which the verification entry points use the lane-distributed kernels
// stmt_tables (optional): fixed-base tables of the n_stmt statement points (launch_fixed_tables), which replace the statement's 255-step scalar multiplications
hipError_t launch_groth16_verify_small(const uint32_t* A, const uint32_t* B, const uint32_t* C, const uint32_t* uvw_stmt, const uint32_t* stmt_tables, const uint32_t* stmt, int n_stmt,
                                       const uint32_t* gamma, const uint32_t* delta, const uint32_t* alpha_beta, uint32_t* tmp, uint32_t* S, uint32_t* ok, size_t n,
                                       unsigned long long* err, hipStream_t s, const uint32_t* ate_target = nullptr);
// prod_k tate(+-P_k, Q_k) == 1 per element, K <= 4 pairs sharing one Miller squaring chain and one final exponentiation.
// Slot k reads its G1 point at g1[k] + i*s1[k] words (stride 0 = the same point for every element), likewise G2; neg[k] negates P_k.
struct PairArgs { const uint32_t* g1[4]; const uint32_t* g2[4]; uint32_t s1[4], s2[4]; uint32_t neg[4]; };
// p_trusted: bit k set = slot k's G1 point is a constant of the library known to lie in G1 (no membership test spent on it)
hipError_t launch_pairing_product_check(const PairArgs& a, int K, uint32_t* ok, size_t n, unsigned long long* err, hipStream_t s, uint32_t p_trusted = 0);
// lane-distributed pairing (zkt_dpairing.hip): diagnostic Fq12 ops on the distributed form
hipError_t launch_dfq12_op(int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n, hipStream_t s);
// one pairing per 12 lanes; elements whose P is outside G1 get out[i*144 + mark_word] = mark (see zkt_tate.hip)
hipError_t launch_dtate(const uint32_t* g1, const uint32_t* g2, uint32_t* out, size_t n, unsigned long long* err, uint32_t mark_word, uint32_t mark, bool short_loop, hipStream_t s);
// prod_k tate(+-P_k, Q_k) == target (NULL: == 1) with the K Miller loops in K lane groups of one wave (small batches)
// kcount (optional, device): element i multiplies only its first kcount[i] <= K pairs; its unused slots must hold a copy of its pair 0
hipError_t launch_dproduct(const PairArgs& a, int K, const uint32_t* target, uint32_t* ok, size_t n, unsigned long long* err, bool short_loop, hipStream_t s, const uint8_t* kcount = nullptr);
// the same decision on the 63-step loop (target: the ate counterpart of a key's alpha_beta, or NULL for == 1); ok = 2 where a Q is outside G2 (left to the kernels behind)
hipError_t launch_dproduct_ate(const PairArgs& a, int K, const uint32_t* target, uint32_t* ok, size_t n, unsigned long long* err, hipStream_t s, const uint8_t* kcount = nullptr);
hipError_t launch_key_ab(const uint32_t* alpha, const uint32_t* beta, uint32_t* out, uint32_t* flag, uint32_t bit, uint32_t* gt, hipStream_t s);      // one launch, two lane groups' worth of blocks: a(beta, alpha)^(3h) into out (*flag |= bit when beta is in G2) and tate(alpha, beta) into gt
hipError_t launch_ate_guards(const PairArgs& a, int K, uint32_t* flags, size_t n, hipStream_t s, uint32_t p_skip = 0);      // flags[i] = 1: every P of element i on E and in G1, every Q on E'
// Small batches of products with DIFFERENT pair counts in one launch (K = the largest): the 127-step kernels with their guards beside them, as
// launch_pairing_product_check does for n*K <= the small-batch limit, but WITHOUT the 255-step / exact re-evaluation: ok[i] = 2 means "element i does not fit
// the short loop, evaluate it another way".  n * K must be within the small-batch limit.
hipError_t launch_pairing_product_check_counts(const PairArgs& a, int K, const uint8_t* kcount, uint32_t* ok, size_t n, unsigned long long* err, hipStream_t s);
// The 127-step loop of the small-batch kernels runs on trust; its preconditions (points on their curves, Q in G2) are checked by a one-lane-per-element
// kernel on a side stream meanwhile.  guard_fork makes `side` wait for everything queued on `s`; guard_join makes `s` wait for the side stream.
hipError_t guard_fork(hipStream_t s, hipStream_t* side);
hipError_t guard_join(hipStream_t s, hipStream_t side);
hipError_t launch_short_loop_guards(const PairArgs& a, int K, uint32_t* flags, size_t n, hipStream_t s);      // flags[i] = 1: every pair of element i fits the 127-step loop
// zkt_shutdown: release what is bound to the device of the current zkt_init (generator comb tables; the guard side stream and its events), so that
// a later zkt_init on another device starts clean
void group_release_device_state();
void pairing_release_device_state();
hipError_t launch_tate(const uint32_t* g1, const uint32_t* g2, uint32_t* out, size_t n, unsigned long long* err, hipStream_t s);

// ---- MSM (zkt_msm.hip), generic over the group (G_G1, G_G2, G_SECP) ------------------------------------------
struct MsmPlan {
  size_t n;            // terms
  int grp;             // GroupId
  int c;               // window bits
  int nwin;            // windows
  size_t nbuckets;     // resident form: 2^(c-1), shared by all windows (window multiples are precomputed); direct form: nwin * 2^(c-1)
  size_t ws_bytes;     // workspace bytes per in-flight MSM
  int direct;          // 1 = table-free one-shot form: `table` is the n bases themselves, every window has its own 2^(c-1) buckets
  size_t half;         // buckets per window, 2^(c-1)
  uint32_t chunk;      // most entries one accumulate task (lane) adds: buckets with more are cut into equal pieces (8..128, pick_chunk)
  int aff_rounds;      // G2, large resident MSMs: pair-tree rounds in affine coordinates ahead of the XYZZ accumulate (zkt_msm_affine.hip); 0 = none
  size_t aff_off;      // byte offset of their buffers inside the workspace
};
// buffers of the affine rounds (zkt_msm_affine.hip): layer r >= 1 = points + infinity bytes at (offsets[b] >> r) + b; cntR / offR describe the last layer
struct MsmAffineWs { uint32_t *cntR, *offR, *pref; uint32_t* pts[5]; uint8_t* inf[5]; };
static constexpr int MSM_AFFINE_MAX_ROUNDS = 4;
size_t msm_affine_ws_bytes(size_t entries, size_t nbuckets, int rounds, int coord_words);
MsmAffineWs msm_affine_carve(void* base, size_t entries, size_t nbuckets, int rounds, int coord_words);
hipError_t launch_msm_affine_final_layer(const uint32_t* offsets, size_t nbuckets, int rounds, const MsmAffineWs& w, hipStream_t s);
hipError_t launch_msm_affine_rounds_g2(const uint32_t* table, const uint32_t* entries, const uint32_t* offsets, size_t nbuckets, size_t entries_bound, int rounds,
                                       const MsmAffineWs& w, hipStream_t s);
MsmPlan msm_plan(size_t n, int grp);
// table-free form for one-shot calls (zkt_*_msm with host pointers): no window-multiple table to build — nwin bucket sets, the per-window
// sums reduced side by side (grid.y = window) and joined by nwin-1 runs of c doublings
MsmPlan msm_plan_direct(size_t n, int grp);
// table: nwin*n affine points (x,y raw Montgomery coordinates), inf: nwin*n bytes.  scalars: n x 8 u32.
// Three stages so the API layer can run them on three streams (sort | accumulate | reduce) and overlap
// consecutive MSMs; the result is a Jacobian partial (3 coordinates) and optionally the affine ABI point.
hipError_t launch_msm_to_kernel_layout(int grp, const uint32_t* abi_pts, uint32_t* table, uint8_t* base_inf, size_t n, hipStream_t s);
// tmp: (nwin - 1) * n * 2 coordinates of scratch (same layout as the table rows 1..nwin-1), only needed during the call
hipError_t launch_msm_precompute(int grp, uint32_t* table, uint8_t* inf, size_t n, int c, int nwin, uint32_t* tmp, hipStream_t s);
hipError_t launch_msm_sort(const MsmPlan& plan, const uint8_t* base_inf, const uint32_t* scalars, void* workspace, hipStream_t s);
hipError_t launch_msm_accumulate(const MsmPlan& plan, const uint32_t* table, void* workspace, hipStream_t s);
hipError_t launch_msm_reduce(const MsmPlan& plan, void* workspace, uint32_t* dev_result_jac, uint32_t* dev_out_abi, hipStream_t s);
// out = a + b on two Jacobian partials (3 coordinates each)
hipError_t launch_msm_jac_add(int grp, const uint32_t* a, const uint32_t* b, uint32_t* out, hipStream_t s);
// sum of `count` Jacobian partials, `stride_words` u32 apart (3 coordinates each), normalised to one affine ABI point
hipError_t launch_msm_jac_sum_to_affine(int grp, const uint32_t* jac_partials, size_t count, size_t stride_words, uint32_t* out_abi_pt, hipStream_t s);

}  // namespace zkt

// G2 bucket accumulation with two lanes per task (zkt_msm_g2pair.hip, a translation unit with its own namespace): global scope
hipError_t zkt_launch_accumulate_g2_pair(const uint32_t* table, const uint32_t* entries, const uint32_t* offsets, const void* order, const uint32_t* task_off,
                                         size_t nbuckets, uint32_t* sums, uint32_t* partial, size_t max_tasks, hipStream_t s);
// DIRECT form behind the affine rounds: bucket b = the points pts[off[b] .. off[b] + cnt[b]) of the last layer (inf[slot] != 0: skip), no entry list, no signs
hipError_t zkt_launch_accumulate_g2_pair_direct(const uint32_t* pts, const uint8_t* inf, const uint32_t* off, const uint32_t* cnt, const void* order, const uint32_t* task_off,
                                                size_t nbuckets, uint32_t* sums, uint32_t* partial, size_t max_tasks, hipStream_t s);
