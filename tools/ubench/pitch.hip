// Micro-benchmark (r02): does the reference's plane pitch (Stride 1024 for a 640-wide picture, MD.cs:50-52) cost HBM
// bandwidth?  Copies N frames of W x H (Y + UV, the kernels' slot layout) from one ring slot to another with one wave per
// octet of macroblocks (16 rows x 128 B luma + 2 x 8 rows x 64 B chroma), for several pitches.  Bytes counted: 3*W*H per
// frame (1.5 read + 1.5 written), as bench.py does.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
__global__ __launch_bounds__(64) void oct_copy(const uint8_t *src, uint8_t *dst, int W, int H, int P, uint32_t opr, uint32_t opc, uint32_t n_oct, uint32_t per_xcd, size_t frame_bytes) {
  const uint32_t oi = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (oi >= n_oct) return;
  const uint32_t clip = oi / opc, rem = oi % opc, mby = rem / opr, ox = rem % opr;
  const int lane = threadIdx.x;
  const size_t base = (size_t)clip * frame_bytes, ysz = (size_t)P * H;
  const int nmb = (W / 16 - (int)ox * 8) < 8 ? (W / 16 - (int)ox * 8) : 8;
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const int i = lane + 64 * it, gq = i & 7, yrow = i >> 3;
    if (gq < nmb) {
      const size_t oy = base + (size_t)(mby * 16 + yrow) * P + ox * 128 + gq * 16;
      *(uint4 *)(dst + oy) = *(const uint4 *)(src + oy);
      const int row = yrow & 7;
      const size_t oc = base + ysz + (size_t)(mby * 8 + row) * P + it * (P >> 1) + ox * 64 + gq * 8;
      *(uint2 *)(dst + oc) = *(const uint2 *)(src + oc);
    }
  }
}
int main(int argc, char **argv) {
  const int W = argc > 1 ? atoi(argv[1]) : 640, H = argc > 2 ? atoi(argv[2]) : 480;
  const int clips = argc > 3 ? atoi(argv[3]) : 4096;
  const int pitches[] = {W, W + 64, W + 128, 768, 896, 1024, 1024 + 64, 1024 + 128, 2048};
  for (int P : pitches) {
    if (P < W || (P & 15)) continue;
    const size_t frame = (size_t)P * H * 3 / 2, total = frame * clips + 4096;
    uint8_t *a, *b;
    if (hipMalloc(&a, total) != hipSuccess || hipMalloc(&b, total) != hipSuccess) { printf("pitch %d: alloc failed\n", P); continue; }
    (void)hipMemset(a, 1, total); (void)hipMemset(b, 0, total);
    const uint32_t opr = (W / 16 + 7) / 8, opc = opr * (H / 16), n_oct = opc * clips, grid = (n_oct + 7) / 8 * 8;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
      (void)hipEventRecord(e0, 0);
      hipLaunchKernelGGL(oct_copy, dim3(grid), dim3(64), 0, 0, a, b, W, H, P, opr, opc, n_oct, grid / 8, frame);
      (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    const double bytes = 3.0 * W * H * clips;
    printf("%dx%d pitch %4d  %d frames  %.3f ms  %.2f TB/s algorithmic (%.1f GB)\n", W, H, P, clips, best, bytes / best / 1e9, bytes / 1e9);
    (void)hipFree(a); (void)hipFree(b);
  }
  return 0;
}
