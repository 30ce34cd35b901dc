// Micro-benchmark (r02): how fast can the chip START waves?  Kernels that do (almost) nothing, launched as N workgroups of
// 64 / 256 threads with 0 / 8 KB of LDS and few / many registers.  ns per wave = the floor a one-wave-per-item kernel cannot beat.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
template <int LDS, int REGS>
__global__ void k(uint32_t *out, uint32_t n) {
  __shared__ uint32_t lds[LDS ? LDS / 4 : 1];
  uint32_t v = threadIdx.x;
  if (LDS) { lds[threadIdx.x & 63] = v; v += lds[(threadIdx.x + 1) & 63]; }
  if (REGS > 32) { // keep many registers alive
    uint32_t r[REGS ? REGS : 1];
#pragma unroll
    for (int i = 0; i < REGS; i++) r[i] = v * (i + 1);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < REGS; i++) asm volatile("" : "+v"(r[i]));
#pragma unroll
    for (int i = 0; i < REGS; i++) v ^= r[i];
  }
  if (v == 0xFFFFFFFFu && blockIdx.x == n) out[0] = v;
}
template <int LDS, int REGS> void run(const char *tag, uint32_t *out, uint32_t wgs, int threads) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float best = 1e9;
  for (int rep = 0; rep < 4; rep++) {
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL((k<LDS, REGS>), dim3(wgs), dim3(threads), 0, 0, out, wgs + 1);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    if (rep && ms < best) best = ms;
  }
  const double waves = (double)wgs * (threads / 64);
  printf("%-34s %8u workgroups x %3d threads  %.3f ms  %.2f ns per wave  %.2f ns per workgroup\n", tag, wgs, threads, best, best * 1e6 / waves, best * 1e6 / wgs);
}
int main() {
  uint32_t *out; (void)hipMalloc(&out, 64);
  const uint32_t W = 3000000;
  run<0, 0>("empty, 64 threads", out, W, 64);
  run<0, 0>("empty, 256 threads", out, W / 4, 256);
  run<0, 0>("empty, 1024 threads", out, W / 16, 1024);
  run<8192, 0>("8 KB LDS, 64 threads", out, W, 64);
  run<8192, 0>("8 KB LDS (per WG), 256 threads", out, W / 4, 256);
  run<0, 96>("96 VGPRs, 64 threads", out, W, 64);
  run<8192, 96>("8 KB LDS + 96 VGPRs, 64 threads", out, W, 64);
  run<8192, 96>("8 KB LDS + 96 VGPRs, 256 threads", out, W / 4, 256);
  return 0;
}
