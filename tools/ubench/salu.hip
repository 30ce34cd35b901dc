// Micro-benchmark: scalar ALU issue rate on gfx950, alone and interleaved with VALU work.
// Question: is the scalar unit a co-limiter (one SALU per SIMD issue slot) for a kernel with ~equal
// SALU and VALU instruction counts?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed) {
  uint32_t sa = seed, sb = seed * 3 + 1, sc = seed ^ 0x55, sd = seed + 7, se = seed + 11, sf = seed ^ 3, sg = seed + 5, sh = seed ^ 9;
  uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x55, d = b + 7, e = a + 11, f = b ^ 3, g = c + 5, h = d ^ 9;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (OP == 0 || OP == 2) { // 8 SALU
        asm volatile("s_add_u32 %0, %0, %1\n s_add_u32 %2, %2, %3\n s_add_u32 %4, %4, %5\n s_add_u32 %6, %6, %7\n"
                     "s_xor_b32 %1, %1, %0\n s_xor_b32 %3, %3, %2\n s_xor_b32 %5, %5, %4\n s_xor_b32 %7, %7, %6"
                     : "+s"(sa), "+s"(sb), "+s"(sc), "+s"(sd), "+s"(se), "+s"(sf), "+s"(sg), "+s"(sh) : : "scc");
      }
      if (OP == 1 || OP == 2) { a += b; c += d; e += f; g += h; b ^= a; d ^= c; f ^= e; h ^= g; } // 8 VALU
    }
  }
  if ((a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ sa ^ sb ^ sc ^ sd ^ se ^ sf ^ sg ^ sh) == 0x12345678u) out[0] = a;
}
template <int OP> void run(const char *tag, uint32_t *out, int wps) {
  const int iters = 2000, blocks = 256 * wps; // wps waves per SIMD
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
  (void)hipEventRecord(a, 0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 2u);
  (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  double groups_per_simd = (double)wps * iters * 8; // one group = 8 SALU and/or 8 VALU
  double cycles = ms * 1e-3 * 2.4e9;
  printf("%-34s waves/SIMD %d: %.3f ms -> %.2f cycles per group of 8 per SIMD\n", tag, wps, ms, cycles / groups_per_simd);
}
int main() {
  uint32_t *out; (void)hipMalloc(&out, 64);
  for (int w : {1, 5}) {
    if (w == 1) { run<0>("8 SALU", out, 1); run<1>("8 VALU", out, 1); run<2>("8 SALU + 8 VALU interleaved", out, 1); }
    else { run<0>("8 SALU", out, 5); run<1>("8 VALU", out, 5); run<2>("8 SALU + 8 VALU interleaved", out, 5); }
  }
  return 0;
}
