// Micro-benchmark (r02): what does the lane -> (macroblock, row, 16-byte half) assignment of the window fetches cost?
// One wave per octet of macroblocks, 640x480 frames at pitch 1024 (the kernels' slot layout), pseudo-random motion vectors
// of +-8 pixels per macroblock; every variant brings 16 luma rows x 32 B and 2 x 8 chroma rows x 16 B per macroblock into
// LDS (global_load_lds_dwordx4), waits, and stores the octet as whole rows.  No arithmetic.
//   V0  lane = j*8 + g, j = row*2 + half      (r02 kernel: the two halves of a row are 8 lanes apart)
//   V1  lane = r*16 + g*2 + half              (the two halves of a row are adjacent lanes)
//   V2  as V1, plus descriptor-like 32-byte load per lane first (dependent address), as the kernel has
//   V3  V1 with one 32-byte-aligned window (pos & ~31): how much do misaligned 32-byte windows cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
enum { S = 1024, W = 640, H = 480, MBW = 40, MBH = 30, OPR = 5, OPC = 150 };
static const size_t YSZ = (size_t)S * H, SLOT = YSZ * 3 / 2, CLIP = SLOT * 2;
typedef const void __attribute__((address_space(1))) *gptr_t;
typedef void __attribute__((address_space(3))) *lptr_t;
#define DMA16(src, dst) __builtin_amdgcn_global_load_lds((gptr_t)(src), (lptr_t)(dst), 16, 0, 0)
__device__ __forceinline__ int mvhash(uint32_t mb) { uint32_t h = mb * 2654435761u; h ^= h >> 15; return (int)(h & 15) - 8 + (((int)((h >> 8) & 15) - 8) * S); }
template <int V>
__global__ __launch_bounds__(64) void fetch(uint8_t *planes, const uint4 *desc, int n_clips, uint32_t per_xcd) {
  __shared__ __attribute__((aligned(16))) uint8_t L[8192];
  const uint32_t oi = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (oi >= (uint32_t)OPC * n_clips) return;
  const uint32_t clip = oi / OPC, rem = oi % OPC, mby = rem / OPR, ox = rem % OPR;
  uint8_t *base = planes + (size_t)clip * CLIP;
  const int lane = threadIdx.x;
  const int off0 = mby * 16 * S + ox * 128;
  int g, r, h;
  if (V == 0) { g = lane & 7; r = lane >> 4; h = (lane >> 3) & 1; }
  else { g = (lane >> 1) & 7; r = lane >> 4; h = lane & 1; }
  const uint32_t mb = (clip * MBH + mby) * MBW + ox * 8 + g;
  int mv = mvhash(mb);
  if (V == 2) { const uint4 d = desc[mb * 2], d2 = desc[mb * 2 + 1]; mv += (int)((d.x + d2.y) & 0); }
  // keep windows inside the picture
  int pos = off0 + g * 16 + mv;
  if (mby == 0 || mby == MBH - 1 || (ox == 0 && g == 0) || (ox == OPR - 1 && g == 7)) pos = off0 + g * 16;
  const int al = V == 3 ? ~31 : ~3;
  const uint8_t *sy = base + SLOT + ((pos + r * S) & al) + h * 16;
#pragma unroll
  for (int t = 0; t < 4; t++) DMA16(sy + 4 * t * S, L + t * 1024);
  // chroma: round = plane, 8 rows x 8 MBs, one chunk each: lane = row*8 + g (V0) or row*8 + g (same)
  const int cg = lane & 7, cr = lane >> 3;
  const uint32_t cmb = (clip * MBH + mby) * MBW + ox * 8 + cg;
  int cpos = (off0 >> 1) + cg * 8 + (mvhash(cmb) >> 1);
  if (mby == 0 || mby == MBH - 1 || (ox == 0 && cg == 0) || (ox == OPR - 1 && cg == 7)) cpos = (off0 >> 1) + cg * 8;
  const uint8_t *sc = base + SLOT + YSZ + ((cpos + cr * S) & ~3);
  DMA16(sc, L + 4096);
  DMA16(sc + S / 2, L + 5120);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const int i = lane + 64 * it, gq = i & 7, row16 = i >> 3;
    *(uint4 *)(base + off0 + row16 * S + gq * 16) = *(const uint4 *)(L + row16 * 256 + gq * 16);
    const int row = row16 & 7;
    *(uint2 *)(base + YSZ + (off0 >> 1) + it * (S >> 1) + row * S + gq * 8) = *(const uint2 *)(L + 4096 + it * 1024 + row * 128 + gq * 8);
  }
}
template <int V> float run(uint8_t *p, const uint4 *desc, int n_clips) {
  const uint32_t n = OPC * n_clips, grid = (n + 7) / 8 * 8;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float best = 1e9;
  for (int rep = 0; rep < 5; rep++) {
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(fetch<V>, dim3(grid), dim3(64), 0, 0, p, desc, n_clips, grid / 8);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    if (rep && ms < best) best = ms;
  }
  return best;
}
int main(int argc, char **argv) {
  const int n_clips = argc > 1 ? atoi(argv[1]) : 4096;
  uint8_t *p; if (hipMalloc(&p, CLIP * n_clips + 8192) != hipSuccess) return 1;
  (void)hipMemset(p, 1, CLIP * n_clips + 8192);
  p += 4096;
  uint4 *desc; (void)hipMalloc(&desc, (size_t)n_clips * MBW * MBH * 32 + 4096); (void)hipMemset(desc, 0, (size_t)n_clips * MBW * MBH * 32 + 4096);
  const double bytes = 3.0 * W * H * n_clips;
  const char *names[] = {"V0 halves 8 lanes apart", "V1 halves adjacent", "V2 V1 + descriptor load", "V3 V1, 32-B aligned windows"};
  float ms[4] = {run<0>(p, desc, n_clips), run<1>(p, desc, n_clips), run<2>(p, desc, n_clips), run<3>(p, desc, n_clips)};
  for (int v = 0; v < 4; v++) printf("%-30s %d clips  %.3f ms  %.2f TB/s algorithmic\n", names[v], n_clips, ms[v], bytes / ms[v] / 1e9);
  return 0;
}
