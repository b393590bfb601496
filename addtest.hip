#include <hip/hip_runtime.h>
#include <stdint.h>
#define N 12
__device__ constexpr uint32_t P[12] = {0xffffaaab,0xb9feffff,0xb153ffff,0x1eabfffe,0xf6b0f624,0x6730d2a0,0xf38512bf,0x64774b84,0x434bacd7,0x4b1ba7b6,0x397fe69a,0x1a0111ea};
__device__ __forceinline__ void add_mod(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  uint32_t t[N]; uint64_t c = 0;
#pragma unroll
  for (int j = 0; j < N; ++j) { c += (uint64_t)a[j] + b[j]; t[j] = (uint32_t)c; c >>= 32; }
  uint32_t s[N]; uint64_t bw = 0;
#pragma unroll
  for (int j = 0; j < N; ++j) { uint64_t d = (uint64_t)t[j] - P[j] - bw; s[j] = (uint32_t)d; bw = (d>>63)&1; }
  bool ge = !bw;   // a+b < 2^384 always since p < 2^381
#pragma unroll
  for (int j = 0; j < N; ++j) r[j] = ge ? s[j] : t[j];
}
__device__ __forceinline__ void sub_mod(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  uint32_t t[N]; uint64_t bw = 0;
#pragma unroll
  for (int j = 0; j < N; ++j) { uint64_t d = (uint64_t)a[j] - b[j] - bw; t[j] = (uint32_t)d; bw = (d>>63)&1; }
  uint32_t mask = (uint32_t)0 - (uint32_t)bw; uint64_t c = 0;
#pragma unroll
  for (int j = 0; j < N; ++j) { c += (uint64_t)t[j] + (P[j] & mask); r[j] = (uint32_t)c; c >>= 32; }
}
__global__ void kadd(uint32_t* out, const uint32_t* in) {
  int tid = blockIdx.x*blockDim.x + threadIdx.x;
  uint32_t a[N], b[N], r[N];
  for (int j = 0; j < N; ++j) { a[j] = in[tid*N+j]; b[j] = in[(tid+1)*N+j]; }
  add_mod(r,a,b);
  for (int j = 0; j < N; ++j) out[tid*N+j] = r[j];
}
__global__ void ksub(uint32_t* out, const uint32_t* in) {
  int tid = blockIdx.x*blockDim.x + threadIdx.x;
  uint32_t a[N], b[N], r[N];
  for (int j = 0; j < N; ++j) { a[j] = in[tid*N+j]; b[j] = in[(tid+1)*N+j]; }
  sub_mod(r,a,b);
  for (int j = 0; j < N; ++j) out[tid*N+j] = r[j];
}
