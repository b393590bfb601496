#include <hip/hip_runtime.h>
#include <stdint.h>
#define N 12
__device__ constexpr uint32_t P[12] = {0xffffaaab,0xb9feffff,0xb153ffff,0x1eabfffe,0xf6b0f624,0x6730d2a0,0xf38512bf,0x64774b84,0x434bacd7,0x4b1ba7b6,0x397fe69a,0x1a0111ea};
#define INV 0xfffcfffdu
__device__ __forceinline__ void mac(uint64_t& lo, uint32_t& hi, uint32_t a, uint32_t b) {
  unsigned long long c; lo = __builtin_addcll(lo, (unsigned long long)a * b, 0, &c); hi += (uint32_t)c;
}
__device__ __forceinline__ void mul_e(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  uint64_t lo = 0; uint32_t hi = 0; uint32_t m[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) mac(lo, hi, a[i], b[k-i]);
#pragma unroll
    for (int j = 0; j < k; ++j) mac(lo, hi, m[j], P[k-j]);
    m[k] = (uint32_t)lo * INV;
    mac(lo, hi, m[k], P[0]);
    lo = (lo >> 32) | ((uint64_t)hi << 32); hi = 0;
  }
#pragma unroll
  for (int k = N; k < 2*N; ++k) {
#pragma unroll
    for (int i = k-N+1; i < N; ++i) mac(lo, hi, a[i], b[k-i]);
#pragma unroll
    for (int j = k-N+1; j < N; ++j) mac(lo, hi, m[j], P[k-j]);
    r[k-N] = (uint32_t)lo;
    lo = (lo >> 32) | ((uint64_t)hi << 32); hi = 0;
  }
  uint32_t s[N]; unsigned bw = 0;
#pragma unroll
  for (int j = 0; j < N; ++j) s[j] = __builtin_subc(r[j], P[j], bw, &bw);
#pragma unroll
  for (int j = 0; j < N; ++j) r[j] = bw ? r[j] : s[j];
}
__global__ void k2(uint32_t* out, const uint32_t* in, int iters) {
  int tid = blockIdx.x*blockDim.x + threadIdx.x;
  uint32_t a[N], b[N];
  for (int j = 0; j < N; ++j) { a[j] = in[tid*N+j]; b[j] = in[(tid+1)*N+j]; }
  for (int it = 0; it < iters; ++it) {
    uint32_t r[N];
    mul_e(r,a,b);
    for (int j = 0; j < N; ++j) { a[j] = b[j]; b[j] = r[j]; }
  }
  for (int j = 0; j < N; ++j) out[tid*N+j] = b[j];
}
