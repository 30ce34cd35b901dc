// Micro-benchmark: integer VALU issue rate on gfx950 (wave64).  How many cycles does one v_add_u32 /
// v_and_b32 / v_lshrrev_b32 / v_alignbyte / v_lshrrev_b64 / v_mul_lo_u32 occupy a SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed) {
  uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x55, d = b + 7;
  uint32_t e = a + 11, f = b ^ 3, g = c + 5, h = d ^ 9;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (OP == 0) { a += b; c += d; e += f; g += h; b += a; d += c; f += e; h += g; }
      if (OP == 1) { a = (a >> 1) & 0x7F7F7F7Fu; c = (c >> 3) & 0x7F7F7F7Fu; e = (e >> 1) & b; g = (g >> 2) & d; a += 0x80808080u; c ^= f; e ^= h; g ^= b; }
      if (OP == 2) { a = __builtin_amdgcn_alignbyte(a, b, c); c = __builtin_amdgcn_alignbyte(c, d, e); e = __builtin_amdgcn_alignbyte(e, f, g); g = __builtin_amdgcn_alignbyte(g, h, a); b ^= a; d ^= c; f ^= e; h ^= g; }
      if (OP == 3) { uint64_t x = ((uint64_t)a << 32 | b) >> (c & 31); uint64_t y = ((uint64_t)e << 32 | f) >> (g & 31); a = (uint32_t)x; b = (uint32_t)(x >> 8); e = (uint32_t)y; f = (uint32_t)(y >> 8); c += d; g += h; }
      if (OP == 4) { a *= b; c *= d; e *= f; g *= h; b += 1; d += 1; f += 1; h += 1; }
    }
  }
  if ((a ^ b ^ c ^ d ^ e ^ f ^ g ^ h) == 0x12345678u) out[0] = a;
}
template <int OP> void run(const char *tag, uint32_t *out, int valu_per_unroll) {
  const int iters = 2000, blocks = 256 * 8; // 8 waves per SIMD
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
  (void)hipEventRecord(a, 0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 2u);
  (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  // per SIMD: 8 waves x iters x 8 x valu_per_unroll wave-instructions
  double instr_per_simd = 8.0 * iters * 8 * valu_per_unroll;
  double cycles = ms * 1e-3 * 2.4e9;
  printf("%-28s %.3f ms  -> %.2f cycles per wave64 VALU instruction per SIMD (at 2.4 GHz)\n", tag, ms, cycles / instr_per_simd);
}
int main() {
  uint32_t *out; (void)hipMalloc(&out, 64);
  run<0>("v_add_u32", out, 8);
  run<1>("shift+and SWAR mix", out, 11);
  run<2>("v_alignbyte + xor", out, 8);
  run<3>("64-bit shift funnel", out, 10);
  run<4>("v_mul_lo_u32 + add", out, 8);
  return 0;
}
