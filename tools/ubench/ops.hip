// Micro-benchmark (r02): issue cost and semantics of the byte-wise multimedia ops the MC stage wants on gfx950:
// v_lerp_u8 (byte average), v_perm_b32 (byte select), v_alignbyte (how many bits of the shift count are honoured),
// v_min3/v_max3, v_pk_* 16-bit.  One line per op: cycles per wave64 instruction per SIMD at 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed) {
  uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x55, d = b + 7;
  uint32_t e = a + 11, f = b ^ 3, g = c + 5, h = d ^ 9;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (OP == 0) { a += b; c += d; e += f; g += h; b += a; d += c; f += e; h += g; }
      if (OP == 1) { a = __builtin_amdgcn_lerp(a, b, c); c = __builtin_amdgcn_lerp(c, d, e); e = __builtin_amdgcn_lerp(e, f, g); g = __builtin_amdgcn_lerp(g, h, a);
                     b = __builtin_amdgcn_lerp(b, a, d); d = __builtin_amdgcn_lerp(d, c, f); f = __builtin_amdgcn_lerp(f, e, h); h = __builtin_amdgcn_lerp(h, g, b); }
      if (OP == 2) { a = __builtin_amdgcn_perm(a, b, c); c = __builtin_amdgcn_perm(c, d, e); e = __builtin_amdgcn_perm(e, f, g); g = __builtin_amdgcn_perm(g, h, a);
                     b = __builtin_amdgcn_perm(b, a, d); d = __builtin_amdgcn_perm(d, c, f); f = __builtin_amdgcn_perm(f, e, h); h = __builtin_amdgcn_perm(h, g, b); }
      if (OP == 3) { auto med = [](uint32_t x, uint32_t lo, uint32_t hi) { int v = (int)x; v = v < (int)lo ? (int)lo : v; v = v > (int)hi ? (int)hi : v; return (uint32_t)v; };
                     a = med(a, 0, 255); c = med(c, 0, 255); e = med(e, 0, 255); g = med(g, 0, 255);
                     a += b; c += d; e += f; g += h; }
      if (OP == 4) { a &= b; c &= d; e &= f; g &= h; b ^= a + 1; d ^= c + 1; f ^= e + 1; h ^= g + 1; } // and, add, xor: 12 VALU... counted as 12
      if (OP == 5) { typedef short s2 __attribute__((ext_vector_type(2)));
                     s2 x = __builtin_bit_cast(s2, a), y = __builtin_bit_cast(s2, b), z = __builtin_bit_cast(s2, c), w = __builtin_bit_cast(s2, d);
                     x += y; z += w; y += x; w += z; x = __builtin_elementwise_max(x, z); y = __builtin_elementwise_min(y, w); z += x; w += y;
                     a = __builtin_bit_cast(uint32_t, x); b = __builtin_bit_cast(uint32_t, y); c = __builtin_bit_cast(uint32_t, z); d = __builtin_bit_cast(uint32_t, w); }
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
  double instr_per_simd = 8.0 * iters * 8 * valu_per_unroll;
  double cycles = ms * 1e-3 * 2.4e9;
  printf("%-28s %.3f ms  -> %.2f cycles per wave64 VALU instruction per SIMD (at 2.4 GHz)\n", tag, ms, cycles / instr_per_simd);
}
// semantics: lerp over all byte pairs and both rounding bits; alignbyte / perm selectors
__global__ void sem(uint32_t *out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; // 0..65535: a = t & 255, b = t >> 8
  const uint32_t a = t & 255, b = t >> 8;
  const uint32_t A = a * 0x01010101u, B = b * 0x01010101u;
  uint32_t bad = 0;
  const uint32_t l0 = __builtin_amdgcn_lerp(A, B, 0u), l1 = __builtin_amdgcn_lerp(A, B, 0x01010101u), l2 = __builtin_amdgcn_lerp(A, B, 0x00010001u);
  if (l0 != ((a + b) >> 1) * 0x01010101u) bad |= 1;
  if (l1 != ((a + b + 1) >> 1) * 0x01010101u) bad |= 2;
  if (l2 != (((a + b + 1) >> 1) * 0x00010001u | ((a + b) >> 1) * 0x01000100u)) bad |= 4;
  // the MC identity: (a>>1)+(b>>1) == lerp(a & 0xFE, b & 0xFE, 0)
  if (__builtin_amdgcn_lerp(A & 0xFEFEFEFEu, B & 0xFEFEFEFEu, 0u) != ((a >> 1) + (b >> 1)) * 0x01010101u) bad |= 8;
  // alignbyte with shift counts 0..7: which bits count?
  const uint32_t hi = 0x77665544u, lo = 0x33221100u;
  uint32_t ab = 0;
  for (uint32_t s = 0; s < 8; s++) {
    const uint32_t r = __builtin_amdgcn_alignbyte(hi, lo, s);
    const uint64_t full = ((uint64_t)hi << 32 | lo) >> (8 * s);
    if (r == (uint32_t)full) ab |= 1u << s;
  }
  // perm: selector bytes 0..7 pick from {S0:S1} (S1 = bytes 0..3)
  const uint32_t p = __builtin_amdgcn_perm(hi, lo, 0x04030201u);
  if (p != 0x44332211u) bad |= 16;
  if (t == 0) { out[1] = ab; }
  if (bad) atomicOr(&out[0], bad);
}
int main() {
  uint32_t *out; (void)hipMalloc(&out, 64); (void)hipMemset(out, 0, 64);
  hipLaunchKernelGGL(sem, dim3(256), dim3(256), 0, 0, out);
  uint32_t h[2]; (void)hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
  printf("semantics: bad=0x%x (0 = lerp/perm as assumed)  alignbyte shift counts that behave as a 64-bit funnel: mask 0x%02x\n", h[0], h[1]);
  run<0>("v_add_u32", out, 8);
  run<1>("v_lerp_u8", out, 8);
  run<2>("v_perm_b32", out, 8);
  run<3>("v_med3_i32 + add", out, 8);
  run<4>("and/add/xor", out, 12);
  run<5>("v_pk_add/max/min_i16", out, 8);
  return 0;
}
