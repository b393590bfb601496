// VALU issue-rate roof of gfx950 for the instruction mix of the field kernels, with an f32 control row.
//   hipcc --offload-arch=gfx950 -O3 -o valu_roof tools/ubench/valu_roof.hip && ./valu_roof > profiles/r02_valu_ubench.txt
// Question settled here (round-1 review, weak #2): MI355X_MICROARCH.md describes SIMD-32 units (a wave64 v_fma_f32 issues in 2
// cycles, 32 lanes/clk/SIMD); the integer multiply-add the field arithmetic is made of was measured at ~4-5 cycles per wave64
// instruction.  Does this harness SEE 32 lanes/clk for f32, i.e. is the integer rate really half of it, or is the harness blind?
// Every row reports lanes/clk/SIMD against the IN-KERNEL clock (s_memtime / s_memrealtime, MI355X_MICROARCH.md "DVFS give-back" (6)),
// not against the 2.4 GHz nameplate, plus the absolute chip-wide rate in T lane-ops/s.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

enum { FMA_F32, PK_FMA_F32, ADD_U32, AND_B32, LSHR_B64, MUL_LO, MAD64, MIX_FIELD, FMA_F64, ADD_F64, LSHL_ADD_U64, MIX_DPF, NOPS };
static const char* NAMES[NOPS] = {"v_fma_f32 (control)", "v_pk_fma_f32 (2 fma/lane)", "v_add_u32", "v_and_b32", "v_lshrrev_b64", "v_mul_lo_u32", "v_mad_u64_u32",
                                  "field mix: 4 mad64 + lshr64 + and + mul_lo + add", "v_fma_f64", "v_add_f64", "v_lshl_add_u64",
                                  "fp64 limb product: fma64 + add64 + fma64 + 2 lshl_add_u64"};
// Round 3 (review item 9, go / no-go): the Montgomery product on 8 x 52-bit limbs held as doubles.  One limb product a_i b_j < 2^104 is
//   hi = fma_rz(a, b, 2^104)      mantissa field = a b >> 52
//   lo = fma_rz(a, b, (2^104 + 2^52) - hi)      mantissa field = a b mod 2^52
// and both raw bit patterns are added into 64-bit column sums (the exponent patterns are subtracted once per column): 2 v_fma_f64 + 1 v_add_f64 + 2 64-bit integer adds
// per limb product, 64 + 72 limb products per Fq product (8 x 8 for a b, 8 rounds of (1 + 8) for the reduction) = 680 instructions + ~70 of carry propagation and
// integer <-> double moves, against 392 v_mad_u64_u32 + ~110 others for the 14 x 28-bit form the library uses.  The last row is that instruction mix.
// instructions issued per inner step (8 independent chains) for each op
__host__ __device__ constexpr int instr_per_step(int op) { return op == MIX_FIELD ? 8 * 8 : op == MIX_DPF ? 8 * 5 : 8; }

template <int OP> __global__ void __launch_bounds__(64) rate(uint64_t* out, unsigned long long* clk, int iters, uint32_t seed) {
  uint64_t acc[8]; float f[8]; uint64_t p[8];      // p: two packed f32
  double d[8], e[8]; const double dx = 4503599627370495.0 - threadIdx.x, dy = 4503599627370001.0 + seed, c1 = 0x1p104, c2 = 0x1p104 + 0x1p52;
  uint32_t x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9e3779b9u;
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc[i] = x + i; f[i] = 1.0f + i; p[i] = 0x3f8000003f800000ull + i; d[i] = 1.0 + i; e[i] = 2.0 + i; }
  float fx = 1.0000001f, fy = 0.9999999f; const uint64_t px = 0x3f8000013f7fffffull;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == FMA_F32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(fx), "v"(fy));
        if (OP == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[i]) : "v"(px));
        if (OP == ADD_U32) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(y)); acc[i] = lo; }
        if (OP == AND_B32) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_and_b32 %0, %0, %1" : "+v"(lo) : "v"(y)); acc[i] = lo; }
        if (OP == LSHR_B64) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[i]));
        if (OP == MUL_LO) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(y)); acc[i] = lo; }
        if (OP == MAD64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "vcc");
        if (OP == FMA_F64) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(dx), "v"(dy));
        if (OP == ADD_F64) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[i]) : "v"(dx));
        if (OP == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(p[i]));
        if (OP == MIX_DPF) {         // one limb product of the fp64 form: hi, the offset, lo, and the two raw patterns added into the column sums
          asm volatile("v_fma_f64 %0, %3, %4, %5\n\tv_add_f64 %1, %6, -%0\n\tv_fma_f64 %1, %3, %4, %1\n\tv_lshl_add_u64 %2, %0, 0, %2\n\tv_lshl_add_u64 %2, %1, 0, %2"
                       : "=&v"(d[i]), "=&v"(e[i]), "+v"(acc[i]) : "v"(dx), "v"(dy), "v"(c1), "v"(c2));
        }
        if (OP == MIX_FIELD) {       // one column of the 28-bit-limb Montgomery product: MADs into a 64-bit accumulator, then shift, mask, m = lo * inv, add
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0\n\tv_mad_u64_u32 %0, vcc, %1, %1, %0\n\tv_mad_u64_u32 %0, vcc, %2, %2, %0\n\t"
                       "v_lshrrev_b64 %0, 28, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "vcc");
          uint32_t lo = (uint32_t)acc[i];
          asm volatile("v_and_b32 %0, 0xfffffff, %0\n\tv_mul_lo_u32 %0, %0, %2\n\tv_add_u32 %0, %0, %1" : "+v"(lo) : "v"(x), "v"(y));
          acc[i] += lo & 1;
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i] + (uint64_t)f[i] + p[i] + (uint64_t)__double_as_longlong(d[i]) + (uint64_t)__double_as_longlong(e[i]);
  out[(size_t)blockIdx.x * 64 + threadIdx.x] = s + x;
  if (threadIdx.x == 0) { clk[2 * (size_t)blockIdx.x] = t1 - t0; clk[2 * (size_t)blockIdx.x + 1] = r1 - r0; }
}

template <int OP> void run(int waves_per_simd, int iters) {
  const int blocks = 256 * 4 * waves_per_simd;
  uint64_t* out; unsigned long long* clk;
  CHK(hipMalloc(&out, (size_t)blocks * 64 * 8)); CHK(hipMalloc(&clk, (size_t)blocks * 16));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(64), 0, 0, out, clk, 10, 1u); CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0)); hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(64), 0, 0, out, clk, iters, 2u); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h((size_t)blocks * 2); CHK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> ghz, cyc;
  for (int b = 0; b < blocks; ++b) if (h[2 * b + 1]) { ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1); cyc.push_back((double)h[2 * b]); }
  std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
  const double clock = ghz[ghz.size() / 2], wave_cycles = cyc[cyc.size() / 2];
  const double instr_per_wave = (double)iters * 4 * instr_per_step(OP);
  const double lane_ops = (double)blocks * 64 * instr_per_wave * (OP == PK_FMA_F32 ? 2 : 1);
  // lanes/clk/SIMD from the SAME measurement as the T lane-ops/s column (event time, 1024 SIMDs, the in-kernel clock): round 2 printed a figure
  // derived from the median per-wave cycle count instead, which at 8 waves per SIMD — where the waves of a SIMD do not all run for the whole
  // launch — read 44 lanes/clk beside a rate that is 28 (evidence hygiene).  The per-wave figure stays as its own column.
  const double rate = lane_ops / (ms * 1e-3);
  const double lanes_per_clk_simd = rate / (1024.0 * clock * 1e9);
  printf("%-52s waves/SIMD=%d  %7.3f ms  clock %.2f GHz  %5.1f lanes/clk/SIMD  cycles/wave-instr/SIMD (median wave) %.2f  %6.1f T lane-ops/s\n", NAMES[OP], waves_per_simd, ms, clock,
         lanes_per_clk_simd, wave_cycles / (instr_per_wave * waves_per_simd), rate / 1e12);
  CHK(hipFree(out)); CHK(hipFree(clk));
}

int main() {
  hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
  printf("# tools/ubench/valu_roof.hip on %s, CUs=%d, nameplate %d kHz.  Independent chains (8 per wave), all CUs busy; lanes/clk/SIMD against the in-kernel clock.\n",
         p.name, p.multiProcessorCount, p.clockRate);
  for (int w : {1, 2, 4, 8}) {
    run<FMA_F32>(w, 4000); run<PK_FMA_F32>(w, 4000); run<ADD_U32>(w, 4000); run<AND_B32>(w, 4000); run<LSHR_B64>(w, 4000); run<MUL_LO>(w, 4000); run<MAD64>(w, 4000);
    run<MIX_FIELD>(w, 1000);
    if (w <= 4) { run<FMA_F64>(w, 4000); run<ADD_F64>(w, 4000); run<LSHL_ADD_U64>(w, 4000); run<MIX_DPF>(w, 1500); }
  }
  return 0;
}
