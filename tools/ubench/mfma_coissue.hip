// Go / no-go for "the constant-operand half of the Montgomery product (m * p) as an i8 MFMA beside the VALU stream" (round-1 review, item 9).
// Measures only what cannot be derived on paper: what k v_mfma_i32_32x32x32_i8 per 392 v_mad_u64_u32 (one field multiply's worth) cost the MAD stream
// of the SAME wave.  16 MFMAs cover m * p for the 64 elements of a wave (49 x 48 byte convolution: 2 K-steps x 4 column tiles x 2 row halves).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_coissue tools/ubench/mfma_coissue.hip && ./mfma_coissue
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int NMFMA> __global__ void __launch_bounds__(64) k(uint64_t* out, int iters, uint32_t seed) {
  uint64_t acc[8]; uint32_t x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9e3779b9u;
  for (int i = 0; i < 8; ++i) acc[i] = x + i;
  v4i a = {(int)x, (int)y, (int)(x ^ y), 7}, b = {3, (int)y, 5, (int)x};
  v16i c0 = {0}, c1 = {0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int blk = 0; blk < 16; ++blk) {          // 16 x (24 or 25 MADs) ~ 392 MADs, one optional MFMA per block
#pragma unroll
      for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "vcc");
      if (blk < NMFMA) { if (blk & 1) c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0); else c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0); }
    }
  }
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  for (int i = 0; i < 16; ++i) s += (uint32_t)(c0[i] + c1[i]);
  out[(size_t)blockIdx.x * 64 + threadIdx.x] = s;
}
template <int NMFMA> void run(int wps) {
  const int blocks = 256 * 4 * wps, iters = 400;
  uint64_t* out; hipMalloc(&out, (size_t)blocks * 64 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NMFMA>, dim3(blocks), dim3(64), 0, 0, out, 10, 1u); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(k<NMFMA>, dim3(blocks), dim3(64), 0, 0, out, iters, 2u); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mads = (double)blocks * 64 * iters * 16 * 24;
  printf("waves/SIMD=%d  MFMA per 384 MADs=%2d  %.3f ms  %.1f T MAD lane-ops/s\n", wps, NMFMA, ms, mads / (ms * 1e-3) / 1e12);
  hipFree(out);
}
int main() {
  printf("# tools/ubench/mfma_coissue.hip: v_mad_u64_u32 stream with k x v_mfma_i32_32x32x32_i8 issued by the same wave\n");
  for (int w : {1, 2, 4}) { run<0>(w); run<8>(w); run<16>(w); }
  return 0;
}
