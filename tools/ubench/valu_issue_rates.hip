// micro-benchmarks: VALU integer/f64 issue rates on gfx950 + Montgomery mul variants
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "mm.hip"

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template<int OP> __global__ void rate(uint64_t* out, int iters, uint32_t seed) {
  uint64_t acc[8]; uint32_t x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9e3779b9u;
  double d[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc[i] = x + i; d[i] = 1.0 + i; }
  double dx = 1.0000001, dy = 0.9999999;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "vcc");
        if (OP == 1) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(y)); acc[i] = lo; }
        if (OP == 2) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(y)); acc[i] = lo; }
        if (OP == 3) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(lo) : "v"(y) : "vcc"); acc[i] = lo; }
        if (OP == 4) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i+1)&7]));
        if (OP == 5) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(dx), "v"(dy));
        if (OP == 6) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(lo) : "v"(y)); acc[i] = lo; }
        if (OP == 7) { uint32_t lo = (uint32_t)acc[i], hi = (uint32_t)(acc[i]>>32); asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(y) : "vcc"); acc[i] = lo | ((uint64_t)hi<<32); }
        if (OP == 8) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(lo) : "v"(y), "v"(x)); acc[i] = lo; }
        if (OP == 9) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[i]), "+v"(x) : "v"(y) : "vcc"); }
      }
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i] + (uint64_t)d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + x;
}

template<int OP> double run_rate(const char* name, int blocks, int threads, int iters) {
  uint64_t* out; CHK(hipMalloc(&out, (size_t)blocks * threads * 8));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  rate<OP><<<blocks, threads>>>(out, 10, 1); CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0)); rate<OP><<<blocks, threads>>>(out, iters, 2); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  double ops = (double)blocks * threads * iters * 32.0;
  double per_cu_clk = ops / (ms * 1e-3) / 256.0 / 2.4e9;   // lane-ops per CU per clock at 2.4 GHz
  printf("%-28s blocks=%d thr=%d  %.3f ms  %.2f Gop/s  %.1f lane-ops/clk/CU (@2.4GHz)\n", name, blocks, threads, ms, ops / ms * 1e-6, per_cu_clk);
  CHK(hipFree(out)); return ms;
}

template<int V> void run_mm(const char* name, int blocks, int threads, int iters) {
  size_t n = (size_t)blocks * threads;
  std::vector<uint32_t> h((n + 1) * 12);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u) & ((i % 12 == 11) ? 0x0fffffffu : 0xffffffffu);
  uint32_t *in, *out; CHK(hipMalloc(&in, h.size() * 4)); CHK(hipMalloc(&out, n * 48));
  CHK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  k<V><<<blocks, threads>>>(out, in, 4); CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0)); k<V><<<blocks, threads>>>(out, in, iters); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<uint32_t> r(12); CHK(hipMemcpy(r.data(), out + 12 * 77, 48, hipMemcpyDeviceToHost));
  printf("%-28s blocks=%d thr=%d  %.3f ms  %.2f G Fq-mul/s   sample=%08x%08x\n", name, blocks, threads, ms, (double)n * iters / ms * 1e-6, r[11], r[0]);
  CHK(hipFree(in)); CHK(hipFree(out));
}

int main() {
  hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
  printf("device: %s  CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  int it = 2000;
  for (int thr : {256, 512}) {
    int blocks = 256 * (2048 / thr);
    run_rate<0>("v_mad_u64_u32", blocks, thr, it);
    run_rate<9>("v_mad_u64_u32+addc", blocks, thr, it);
    run_rate<1>("v_mul_lo_u32", blocks, thr, it);
    run_rate<2>("v_mul_hi_u32", blocks, thr, it);
    run_rate<3>("v_add_co_u32", blocks, thr, it);
    run_rate<7>("v_add_co+addc", blocks, thr, it);
    run_rate<8>("v_add3_u32", blocks, thr, it);
    run_rate<4>("v_lshl_add_u64", blocks, thr, it);
    run_rate<5>("v_fma_f64", blocks, thr, it);
    run_rate<6>("v_mad_u32_u24", blocks, thr, it);
  }
  for (int blocks : {256 * 4, 256 * 8}) {
    run_mm<0>("mont mul A (compiler CIOS)", blocks, 256, 2000);
    run_mm<1>("mont mul C (asm FIPS)", blocks, 256, 2000);
  }
  return 0;
}
