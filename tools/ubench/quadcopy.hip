// Ceiling for the inter kernel's memory shape: 512 clips x 640x480 (pitch 1024, Y then UV per ring slot, 6 slots
// per clip), one wave per quad of 4 macroblocks; read ring slot 1, write ring slot 0.  No decode, no arithmetic.
//   direct : lane = (row, mb): one 16-byte luma load + one 8-byte chroma load -> the same stores
//   staged : the kernel's six global->LDS DMA rounds (17-row windows), one wait, LDS -> whole-row stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
enum { S = 1024, W = 640, H = 480, MBW = 40, MBH = 30, QPR = 10, QPC = 300 };
static const size_t YSZ = (size_t)S * H, SLOT = YSZ * 3 / 2, CLIP = SLOT * 6;
typedef const void __attribute__((address_space(1))) *gptr_t;
typedef void __attribute__((address_space(3))) *lptr_t;
__global__ __launch_bounds__(256) void direct(uint8_t *planes, int n_clips, uint32_t per_xcd) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t qi = ((blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)) * 4 + wave;
  if (qi >= (uint32_t)QPC * n_clips) return;
  const uint32_t clip = qi / QPC, rem = qi % QPC, mby = rem / QPR, qx = rem % QPR;
  uint8_t *base = planes + (size_t)clip * (SLOT * 6);
  const int off0 = mby * 16 * S + qx * 64, gq = lane & 3, yrow = lane >> 2, pl = lane >> 5, row = (lane >> 2) & 7;
  const uint4 y = *(const uint4 *)(base + SLOT + off0 + yrow * S + gq * 16);
  const uint2 c = *(const uint2 *)(base + SLOT + YSZ + (off0 >> 1) + pl * (S >> 1) + row * S + gq * 8);
  *(uint4 *)(base + off0 + yrow * S + gq * 16) = y;
  *(uint2 *)(base + YSZ + (off0 >> 1) + pl * (S >> 1) + row * S + gq * 8) = c;
}
__global__ __launch_bounds__(256) void staged(uint8_t *planes, int n_clips, uint32_t per_xcd, int mvx) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[4][7696];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t qi = ((blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)) * 4 + wave;
  if (qi >= (uint32_t)QPC * n_clips) return;
  const uint32_t clip = qi / QPC, rem = qi % QPC, mby = rem / QPR, qx = rem % QPR;
  uint8_t *base = planes + (size_t)clip * (SLOT * 6);
  uint8_t *L = lds[wave];
  const int g = lane >> 4, j = lane & 15;
  const int off0 = mby * 16 * S + qx * 64, off = off0 + g * 16;
  const uint8_t *yw = base + SLOT + ((off + mvx) & ~15), *cw = base + SLOT + YSZ + (((off >> 1) + (mvx >> 1)) & ~15);
  const uint8_t *p0 = yw + j * S;
  __builtin_amdgcn_global_load_lds((gptr_t)p0, (lptr_t)(L + 0), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((gptr_t)p0, (lptr_t)(L + 1024), 16, 16, 0);
  const uint8_t *p2 = cw + (j >> 1) * S + (j & 1) * 16;
  __builtin_amdgcn_global_load_lds((gptr_t)p2, (lptr_t)(L + 2048), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((gptr_t)(p2 + S / 2), (lptr_t)(L + 3072), 16, 0, 0);
  const int h = j >> 1;
  const uint8_t *p4 = (h == 0 ? yw + 16 * S : h == 1 ? cw + 8 * S : h == 2 ? cw + S / 2 + 8 * S : yw) + (j & 1) * 16;
  __builtin_amdgcn_global_load_lds((gptr_t)p4, (lptr_t)(L + 4096), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const int gq = lane & 3, yrow = lane >> 2, pl = lane >> 5, row = (lane >> 2) & 7;
  const uint4 y = *(const uint4 *)(L + gq * 256 + yrow * 16);
  const uint4 c = *(const uint4 *)(L + 2048 + pl * 1024 + gq * 256 + row * 32);
  *(uint4 *)(base + off0 + yrow * S + gq * 16) = y;
  *(uint2 *)(base + YSZ + (off0 >> 1) + pl * (S >> 1) + row * S + gq * 8) = uint2{c.x, c.y};
}
int main() {
  const int n_clips = 512;
  uint8_t *p; if (hipMalloc(&p, CLIP * n_clips + 8192) != hipSuccess) return 1;
  (void)hipMemset(p, 1, CLIP * n_clips + 8192);
  p += 4096;
  const uint32_t quads = QPC * n_clips, grid = ((quads + 3) / 4 + 7) / 8 * 8;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const double bytes = 3.0 * W * H * n_clips;
  for (int v = 0; v < 3; v++) {
    float best = 1e9;
    for (int rep = 0; rep < 6; rep++) {
      (void)hipEventRecord(a, 0);
      if (v == 0) hipLaunchKernelGGL(direct, dim3(grid), dim3(256), 0, 0, p, n_clips, grid / 8);
      else hipLaunchKernelGGL(staged, dim3(grid), dim3(256), 0, 0, p, n_clips, grid / 8, v == 1 ? 0 : 37);
      (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b);
      if (rep && ms < best) best = ms;
    }
    printf("%-22s %.4f ms  -> %.2f TB/s of algorithmic bytes (%.0f MB)\n", v == 0 ? "direct" : v == 1 ? "staged (aligned)" : "staged (mv +37 px)", best, bytes / best / 1e9, bytes / 1e6);
  }
  return 0;
}
