// latency / issue microbench for v_mad_u64_u32 chains at 1..4 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int CH> __global__ void __launch_bounds__(64) k(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint64_t acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = threadIdx.x + c;
  uint32_t x = a + threadIdx.x, y = b;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int c = 0; c < CH; ++c) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(x), "v"(y) : "vcc");
    }
  }
  uint64_t s = 0; for (int c = 0; c < CH; ++c) s += acc[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int CH> void run(uint64_t* d, int wps) {
  int blocks = 256 * 4 * wps, iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(64), 0, 0, d, 3u, 5u, 10);
  hipEventRecord(e0); hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(64), 0, 0, d, 3u, 5u, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr_per_wave = (double)iters * 16 * CH;
  double clk = ms * 1e-3 * 2.4e9;
  printf("chains=%d waves/SIMD=%d  %.3f ms  clk/instr/wave=%.2f  SIMD issue interval=%.2f clk\n", CH, wps, ms, clk / instr_per_wave, clk / (instr_per_wave * wps));
}
int main() { uint64_t* d; hipMalloc(&d, 8 * 64 * 256 * 4 * 8);
  for (int w = 1; w <= 4; ++w) { run<1>(d, w); run<2>(d, w); run<4>(d, w); }
  return 0; }
