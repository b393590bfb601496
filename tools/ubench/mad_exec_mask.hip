// Does a v_mad_u64_u32 cost less when only part of the wave is active?  Latency-bound single-lane kernels (window joins, table builds, the trees of the
// reduce stage) keep a handful of lanes busy: if the SIMD skipped the 16-lane passes whose lanes are all masked off, packing the active lanes into the
// first 16 would be a free speed-up.  One wave per SIMD, a dependent chain per lane, active lanes = threadIdx.x < ACTIVE.
// Build: hipcc --offload-arch=gfx950 -O2 -o build/mad_exec_mask tools/ubench/mad_exec_mask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(64) k(uint64_t* out, uint32_t a, uint32_t b, int iters, int active, int stride) {
  // the rest of the kernel runs with EXEC = `active` lanes: the first ones (stride 1) or every stride-th lane
  if (threadIdx.x % stride != 0 || (int)(threadIdx.x / stride) >= active) return;
  uint64_t acc0 = threadIdx.x, acc1 = threadIdx.x + 1;
  uint32_t x = a + threadIdx.x, y = b;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc0) : "v"(x), "v"(y) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc1) : "v"(x), "v"(y) : "vcc");
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc0 + acc1;
}
int main() {
  uint64_t* d; hipMalloc(&d, 8 * 64 * 1024);
  const int blocks = 1024, iters = 2000;                // one wave per SIMD
  const int cases[][2] = {{1, 1}, {8, 1}, {16, 1}, {64, 1}, {8, 1}, {12, 1}, {15, 1}, {16, 1}, {1, 1}, {8, 8}, {16, 4}, {4, 16}, {2, 32}, {32, 2}};
  for (auto& cs : cases) { const int a = cs[0], st = cs[1];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, 3u, 5u, 10, a, st);
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, 3u, 5u, iters, a, st); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("active lanes %2d stride %2d: %.3f ms  %.2f clk per v_mad_u64_u32 per wave (2.4 GHz)\n", a, st, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 32));
  }
  return 0;
}
