// Batched prime-field and tower kernels: one element per lane.
// Rows a1–a6 of SURVEY §8: PrimeFieldElem ops (prime_field_elem.rs:278-457), Fq2/Fq6/Fq12
// ops (fq2.rs, fq6.rs, fq12.rs).  Inputs/outputs are canonical residues (include/zkt.h).
#include "abi.h"
#include "fq_program.h"
#include "zkt_internal.h"

namespace zkt {

static constexpr int TPB = 256;
static inline unsigned nblocks(size_t n, int tpb = TPB) { return (unsigned)((n + tpb - 1) / tpb); }

template <class C, int OP>
__global__ void __launch_bounds__(TPB) k_fp_op(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                               uint32_t* __restrict__ out, size_t n, unsigned long long* err) {
  size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  if constexpr (C::W == 28) {                                          // lazy-limb field: enter and leave the Montgomery domain
    constexpr int A = C::ABI_N;
    Fp<C> x = ld_fp<C>(a + i * A), r;
    if (OP == OP_ADD) r = fp_add(x, ld_fp<C>(b + i * A));
    else if (OP == OP_SUB) r = fp_sub(x, ld_fp<C>(b + i * A));
    else if (OP == OP_NEG) r = fp_neg(x);
    else if (OP == OP_MUL) r = fp_mul(x, ld_fp<C>(b + i * A));
    else if (OP == OP_SQR) r = fp_sqr(x);
    else if (OP == OP_CUBE) r = fp_mul(fp_sqr(x), x);
    else if (fp_is_zero(x)) { atomicMin(err, (unsigned long long)i); r = x; }
    else r = fp_inv(x);
    st_fp<C>(out + i * A, r);
  } else {
  // canonical 32-bit-limb fields: any 256-bit input is first reduced mod the order, as PrimeFieldElem::new does (prime_field_elem.rs:263-272)
  Fp<C> x = fp_canon32(ld_raw<C>(a + i * C::N)), r, r2;
  for (int j = 0; j < C::N; ++j) r2.v[j] = C::r2(j);
  if (OP == OP_ADD) r = fp_add(x, fp_canon32(ld_raw<C>(b + i * C::N)));          // canonical in, canonical out
  else if (OP == OP_SUB) r = fp_sub(x, fp_canon32(ld_raw<C>(b + i * C::N)));
  else if (OP == OP_NEG) r = fp_neg(x);
  else if (OP == OP_MUL) r = fp_mul(fp_mul(x, fp_canon32(ld_raw<C>(b + i * C::N))), r2);     // (a b R^-1) R^2 R^-1 = a b
  else if (OP == OP_SQR) r = fp_mul(fp_mul(x, x), r2);
  else if (OP == OP_CUBE) { Fp<C> xm = fp_mul(x, r2); r = fp_mul(fp_mul(xm, xm), x); }        // (xR)(xR)/R = x^2 R;  x^2 R * x / R = x^3
  else {                                                             // safe_inv: Err on zero (prime_field_elem.rs:379-382)
    if (fp_is_zero(x)) { atomicMin(err, (unsigned long long)i); r = x; }
    else {
      Fp<C> one = fp_zero<C>(); one.v[0] = 1;
      r = fp_mul(fp_inv(fp_mul(x, r2)), one);
    }
  }
  st_raw<C>(out + i * C::N, r);
  }
}

template <class C>
static hipError_t launch_fp_c(int op, const uint32_t* a, const uint32_t* b, uint32_t* o, size_t n, unsigned long long* err, hipStream_t s) {
  if (n == 0) return hipSuccess;
  dim3 g(nblocks(n)), t(TPB);
  switch (op) {
    case OP_ADD: hipLaunchKernelGGL((k_fp_op<C, OP_ADD>), g, t, 0, s, a, b, o, n, err); break;
    case OP_SUB: hipLaunchKernelGGL((k_fp_op<C, OP_SUB>), g, t, 0, s, a, b, o, n, err); break;
    case OP_MUL: hipLaunchKernelGGL((k_fp_op<C, OP_MUL>), g, t, 0, s, a, b, o, n, err); break;
    case OP_SQR: hipLaunchKernelGGL((k_fp_op<C, OP_SQR>), g, t, 0, s, a, b, o, n, err); break;
    case OP_NEG: hipLaunchKernelGGL((k_fp_op<C, OP_NEG>), g, t, 0, s, a, b, o, n, err); break;
    case OP_INV: hipLaunchKernelGGL((k_fp_op<C, OP_INV>), g, t, 0, s, a, b, o, n, err); break;
    case OP_CUBE: hipLaunchKernelGGL((k_fp_op<C, OP_CUBE>), g, t, 0, s, a, b, o, n, err); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_fp_op(int field, int op, const uint32_t* a, const uint32_t* b, uint32_t* o, size_t n, unsigned long long* err, hipStream_t s) {
  switch (field) {
    case F_FQ: return launch_fp_c<FqC>(op, a, b, o, n, err, s);
    case F_FR: return launch_fp_c<FrC>(op, a, b, o, n, err, s);
    case F_SP: return launch_fp_c<SpC>(op, a, b, o, n, err, s);
    case F_SN: return launch_fp_c<SnC>(op, a, b, o, n, err, s);
  }
  return hipErrorInvalidValue;
}

// pow / pow_seq / repeat (row a3).  One element per lane; the exponent is scanned from its top set bit, so lanes of a wave run
// as long as the widest exponent among them.
template <class C>
__global__ void __launch_bounds__(TPB) k_fp_pow(const uint32_t* __restrict__ a, const uint32_t* __restrict__ e, int e_words, int shared,
                                                uint32_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  constexpr int A = C::ABI_N;
  st_fp<C>(out + i * A, fp_pow(ld_fp<C>(a + i * A), e + (shared ? 0 : i * (size_t)e_words), e_words));
}
template <class C>
__global__ void __launch_bounds__(TPB) k_fp_pow_seq(const uint32_t* __restrict__ base, uint32_t* __restrict__ out, size_t n, int repeat) {
  size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  constexpr int A = C::ABI_N;
  const Fp<C> b = ld_fp<C>(base);
  const uint32_t e[2] = {(uint32_t)i, (uint32_t)((unsigned long long)i >> 32)};
  st_fp<C>(out + i * A, repeat ? b : fp_pow(b, e, 2));              // base^i: the reference's running product reaches the same residue
}
hipError_t launch_fp_pow(int field, const uint32_t* a, const uint32_t* e, int e_words, bool shared, uint32_t* o, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  dim3 g(nblocks(n)), t(TPB);
  switch (field) {
    case F_FQ: hipLaunchKernelGGL(k_fp_pow<FqC>, g, t, 0, s, a, e, e_words, shared ? 1 : 0, o, n); break;
    case F_FR: hipLaunchKernelGGL(k_fp_pow<FrC>, g, t, 0, s, a, e, e_words, shared ? 1 : 0, o, n); break;
    case F_SP: hipLaunchKernelGGL(k_fp_pow<SpC>, g, t, 0, s, a, e, e_words, shared ? 1 : 0, o, n); break;
    case F_SN: hipLaunchKernelGGL(k_fp_pow<SnC>, g, t, 0, s, a, e, e_words, shared ? 1 : 0, o, n); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_fp_pow_seq(int field, const uint32_t* base, uint32_t* o, size_t n, bool repeat, hipStream_t s) {
  if (n == 0) return hipSuccess;
  dim3 g(nblocks(n)), t(TPB);
  switch (field) {
    case F_FQ: hipLaunchKernelGGL(k_fp_pow_seq<FqC>, g, t, 0, s, base, o, n, repeat ? 1 : 0); break;
    case F_FR: hipLaunchKernelGGL(k_fp_pow_seq<FrC>, g, t, 0, s, base, o, n, repeat ? 1 : 0); break;
    case F_SP: hipLaunchKernelGGL(k_fp_pow_seq<SpC>, g, t, 0, s, base, o, n, repeat ? 1 : 0); break;
    case F_SN: hipLaunchKernelGGL(k_fp_pow_seq<SnC>, g, t, 0, s, base, o, n, repeat ? 1 : 0); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// PrimeFieldElems * PrimeFieldElem (prime_field_elems.rs:152-175): out[i] = a[i] * k, one scalar for the whole vector
template <class C>
__global__ void __launch_bounds__(TPB) k_fp_scale(const uint32_t* __restrict__ a, const uint32_t* __restrict__ k, uint32_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  constexpr int A = C::ABI_N;
  st_fp<C>(out + i * A, fp_mul(ld_fp<C>(a + i * A), ld_fp<C>(k)));
}
// PrimeFieldElems::sum (prime_field_elems.rs:35-41): the fold acc + x from zero.  Grid-stride partial sums per lane, an LDS tree per block, one
// canonical partial per block; the second launch (one block) sums the partials.  Addition is exact in any order, so the residue is the reference's.
static constexpr int SUM_MAX_BLOCKS = 256;
template <class C>
__global__ void __launch_bounds__(TPB) k_fp_sum(const uint32_t* __restrict__ a, size_t n, uint32_t* __restrict__ out) {
  constexpr int A = C::ABI_N;
  __shared__ uint32_t lds[TPB / 2 * C::N];
  Fp<C> acc = fp_zero<C>();
  for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * TPB) acc = fp_add(acc, ld_fp<C>(a + i * A));
  const int t = threadIdx.x;
  for (int d = TPB / 2; d >= 1; d >>= 1) {
    if (t >= d && t < 2 * d) st_raw<C>(lds + (t - d) * C::N, acc);
    __syncthreads();
    if (t < d) acc = fp_add(acc, ld_raw<C>(lds + t * C::N));
    __syncthreads();
  }
  if (t == 0) st_fp<C>(out + (size_t)blockIdx.x * A, acc);
}
template <class C> static hipError_t sum_c(const uint32_t* a, size_t n, uint32_t* o, uint32_t* parts, hipStream_t s) {
  const unsigned nb = (unsigned)(nblocks(n) < (unsigned)SUM_MAX_BLOCKS ? nblocks(n) : SUM_MAX_BLOCKS);
  if (nb <= 1) { hipLaunchKernelGGL(k_fp_sum<C>, dim3(1), dim3(TPB), 0, s, a, n, o); return hipGetLastError(); }
  hipLaunchKernelGGL(k_fp_sum<C>, dim3(nb), dim3(TPB), 0, s, a, n, parts);
  hipLaunchKernelGGL(k_fp_sum<C>, dim3(1), dim3(TPB), 0, s, (const uint32_t*)parts, (size_t)nb, o);
  return hipGetLastError();
}
size_t fp_sum_scratch_elems() { return SUM_MAX_BLOCKS; }
hipError_t launch_fp_sum(int field, const uint32_t* a, size_t n, uint32_t* o, uint32_t* parts, hipStream_t s) {
  switch (field) {
    case F_FQ: return sum_c<FqC>(a, n, o, parts, s);
    case F_FR: return sum_c<FrC>(a, n, o, parts, s);
    case F_SP: return sum_c<SpC>(a, n, o, parts, s);
    case F_SN: return sum_c<SnC>(a, n, o, parts, s);
  }
  return hipErrorInvalidValue;
}
hipError_t launch_fp_scale(int field, const uint32_t* a, const uint32_t* k, uint32_t* o, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  dim3 g(nblocks(n)), t(TPB);
  switch (field) {
    case F_FQ: hipLaunchKernelGGL(k_fp_scale<FqC>, g, t, 0, s, a, k, o, n); break;
    case F_FR: hipLaunchKernelGGL(k_fp_scale<FrC>, g, t, 0, s, a, k, o, n); break;
    case F_SP: hipLaunchKernelGGL(k_fp_scale<SpC>, g, t, 0, s, a, k, o, n); break;
    case F_SN: hipLaunchKernelGGL(k_fp_scale<SnC>, g, t, 0, s, a, k, o, n); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// diagnostic: `count` independent runs of the fp.h self-test program (fq_program.h), one per lane, seeds seed0 + lane
__global__ void __launch_bounds__(64) k_selftest_fq_program(unsigned long long seed0, int steps, const uint32_t* __restrict__ in4, uint32_t* __restrict__ out4,
                                                            int* __restrict__ bad, size_t count) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= count) return;
  bad[i] = fq_program(seed0 + i, steps, in4 + i * 4 * FqC::ABI_N, out4 + i * 4 * FqC::ABI_N);
}
hipError_t launch_selftest_fq_program(unsigned long long seed0, int steps, const uint32_t* in4, uint32_t* out4, int* bad, size_t count, hipStream_t s) {
  if (count == 0) return hipSuccess;
  hipLaunchKernelGGL(k_selftest_fq_program, dim3(nblocks(count, 64)), dim3(64), 0, s, seed0, steps, in4, out4, bad, count);
  return hipGetLastError();
}

// ---- tower ---------------------------------------------------------------------
template <int DEG> struct TowerT;
template <> struct TowerT<2> { typedef Fq2 T; static constexpr int W = 24;
  __device__ static T ld(const uint32_t* p) { return ld_fq2(p); } __device__ static void st(uint32_t* p, const T& v) { st_fq2(p, v); } };
template <> struct TowerT<6> { typedef Fq6 T; static constexpr int W = 72;
  __device__ static T ld(const uint32_t* p) { return ld_fq6(p); } __device__ static void st(uint32_t* p, const T& v) { st_fq6(p, v); } };
template <> struct TowerT<12> { typedef Fq12 T; static constexpr int W = 144;
  __device__ static T ld(const uint32_t* p) { return ld_fq12(p); } __device__ static void st(uint32_t* p, const T& v) { st_fq12(p, v); } };

__device__ inline Fq2 t_op(int op, const Fq2& x, const Fq2& y, bool& zero_inv) {
  switch (op) {
    case T_ADD: return fq2_add(x, y); case T_SUB: return fq2_sub(x, y); case T_MUL: return fq2_mul(x, y);
    case T_NEG: return fq2_neg(x); case T_REDUCE: return fq2_mul_xi(x);
    default: if (fq2_is_zero(x)) { zero_inv = true; return x; } return fq2_inv(x);
  }
}
__device__ inline Fq6 t_op(int op, const Fq6& x, const Fq6& y, bool& zero_inv) {
  switch (op) {
    case T_ADD: return fq6_add(x, y); case T_SUB: return fq6_sub(x, y); case T_MUL: return fq6_mul(x, y);
    case T_NEG: return fq6_neg(x); case T_REDUCE: return fq6_mul_v(x);
    default: if (fq6_is_zero(x)) { zero_inv = true; return x; } return fq6_inv(x);
  }
}
__device__ inline Fq12 t_op(int op, const Fq12& x, const Fq12& y, bool& zero_inv) {
  switch (op) {
    case T_ADD: return fq12_add(x, y); case T_SUB: return fq12_sub(x, y); case T_MUL: return fq12_mul(x, y);
    case T_NEG: return fq12_neg(x);
    default: if (fq6_is_zero(x.c0) && fq6_is_zero(x.c1)) { zero_inv = true; return x; } return fq12_inv(x);
  }
}

template <int DEG>
__global__ void __launch_bounds__(64) k_tower_op(int op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                 uint32_t* __restrict__ out, size_t n, unsigned long long* err) {
  typedef TowerT<DEG> TT;
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  typename TT::T x = TT::ld(a + i * TT::W), y = x;
  if (b) y = TT::ld(b + i * TT::W);
  bool zero_inv = false;
  typename TT::T r = t_op(op, x, y, zero_inv);
  if (zero_inv) atomicMin(err, (unsigned long long)i);
  TT::st(out + i * TT::W, r);
}
hipError_t launch_tower_op(int deg, int op, const uint32_t* a, const uint32_t* b, uint32_t* o, size_t n, unsigned long long* err, hipStream_t s) {
  if (n == 0) return hipSuccess;
  dim3 g(nblocks(n, 64)), t(64);
  if (deg == 2) hipLaunchKernelGGL(k_tower_op<2>, g, t, 0, s, op, a, b, o, n, err);
  else if (deg == 6) hipLaunchKernelGGL(k_tower_op<6>, g, t, 0, s, op, a, b, o, n, err);
  else if (deg == 12) { if (op == T_REDUCE) return hipErrorInvalidValue; hipLaunchKernelGGL(k_tower_op<12>, g, t, 0, s, op, a, b, o, n, err); }
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

__global__ void __launch_bounds__(64) k_fq12_pow(const uint32_t* __restrict__ a, const uint32_t* __restrict__ e, int nl,
                                                 uint32_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  Fq12 x = ld_fq12(a + i * 144);
  st_fq12(out + i * 144, fq12_pow(x, e, nl));
}
hipError_t launch_fq12_pow(const uint32_t* a, const uint32_t* e, int nl, uint32_t* o, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_fq12_pow, dim3(nblocks(n, 64)), dim3(64), 0, s, a, e, nl, o, n);
  return hipGetLastError();
}

// one lane runs a short program of dependent scalar operations (zkt_internal.h: ScalarOps); canonical residues in and out, as k_fp_op
template <class C>
__global__ void __launch_bounds__(64) k_scalar_ops(ScalarOps ops, unsigned long long* err) {
  static_assert(C::W == 32, "canonical 32-bit-limb fields");
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Fp<C> r2; for (int j = 0; j < C::N; ++j) r2.v[j] = C::r2(j);
  for (int k = 0; k < ops.n; ++k) {
    const ScalarOp o = ops.o[k];
    Fp<C> x = fp_canon32(ld_raw<C>(o.a)), r;
    switch (o.op) {
      case OP_ADD: r = fp_add(x, fp_canon32(ld_raw<C>(o.b))); break;
      case OP_SUB: r = fp_sub(x, fp_canon32(ld_raw<C>(o.b))); break;
      case OP_MUL: r = fp_mul(fp_mul(x, fp_canon32(ld_raw<C>(o.b))), r2); break;
      case OP_NEG: r = fp_neg(x); break;
      default:                                                             // OP_INV: safe_inv, Err on zero (prime_field_elem.rs:379-382)
        if (fp_is_zero(x)) { atomicMin(err, 0ull); r = x; }
        else { Fp<C> one = fp_zero<C>(); one.v[0] = 1; r = fp_mul(fp_inv(fp_mul(x, r2)), one); }
    }
    st_raw<C>(o.out, r);
    __threadfence();                                                       // the next operation may read what this one wrote
  }
}
hipError_t launch_scalar_ops(int field, const ScalarOps& ops, unsigned long long* err, hipStream_t s) {
  if (ops.n < 0 || ops.n > 48) return hipErrorInvalidValue;
  if (ops.n == 0) return hipSuccess;
  switch (field) {
    case F_FR: hipLaunchKernelGGL(k_scalar_ops<FrC>, dim3(1), dim3(64), 0, s, ops, err); break;
    case F_SN: hipLaunchKernelGGL(k_scalar_ops<SnC>, dim3(1), dim3(64), 0, s, ops, err); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace zkt
