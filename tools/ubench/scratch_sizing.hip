// How much device memory does one hardware queue take when a kernel with a private (scratch) frame of F bytes per lane runs on it, and does the grid size
// matter?  Prints free memory (hipMemGetInfo) after single-wave and full-chip launches of kernels with 1, 4 and 16 KB frames on fresh streams.
// Build: hipcc --offload-arch=gfx950 -O2 -o build/scratch_sizing tools/ubench/scratch_sizing.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int WORDS>
__global__ void k_frame(unsigned* out, int rounds) {
  volatile unsigned frame[WORDS];                       // volatile: stays in private memory
  for (int i = 0; i < WORDS; ++i) frame[i] = i * 2654435761u + threadIdx.x;
  unsigned acc = 0;
  for (int r = 0; r < rounds; ++r) for (int i = 0; i < WORDS; ++i) acc += frame[(i * 7 + r) % WORDS];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
static double free_gib() { size_t f = 0, t = 0; hipMemGetInfo(&f, &t); return f / 1073741824.0; }
template <int WORDS> static void probe(const char* name, unsigned* out, int blocks) {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const double before = free_gib();
  hipLaunchKernelGGL(k_frame<WORDS>, dim3(blocks), dim3(64), 0, s, out, 1);
  hipStreamSynchronize(s);
  const double after = free_gib();
  printf("%-28s grid %5d waves: free %8.3f -> %8.3f GiB  (-%.3f)\n", name, blocks, before, after, before - after);
  // the stream is left alive on purpose: a destroyed stream may hand its queue (and scratch) back
}
int main() {
  unsigned* out; hipMalloc(&out, 8192 * 64 * 4);
  printf("start: free %.3f GiB\n", free_gib());
  probe<256>("1 KB frame, new stream", out, 1);
  probe<256>("1 KB frame, new stream", out, 8192);
  probe<1024>("4 KB frame, new stream", out, 1);
  probe<1024>("4 KB frame, new stream", out, 8192);
  probe<4096>("16 KB frame, new stream", out, 1);
  probe<4096>("16 KB frame, new stream", out, 1);
  probe<1024>("4 KB frame, new stream", out, 1);
  printf("end: free %.3f GiB\n", free_gib());
  return 0;
}
