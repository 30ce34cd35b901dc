// Micro-benchmark (r03): is a macroblock-tiled PRIVATE plane layout worth building?  (task 1: measure first.)
// Same job as fetchpat.hip -- one wave per octet of macroblocks, 640x480 frames, pseudo-random motion vectors per macroblock, every
// window brought into LDS by global_load_lds_dwordx4, the octet stored back -- in three layouts:
//   LIN   the reference's linear planes at pitch 1024 (fetchpat V1: 16 rows x 32 B + 2 x 8 rows x 16 B per macroblock, no half-pel rows)
//   QUAD  luma: a macroblock = 256 contiguous bytes = four 8x8 quadrants TL TR BL BR of 64 B; a 16-byte chunk = two rows of one quadrant.
//         chroma: a macroblock = 128 contiguous bytes = 8 rows of [U 8 B | V 8 B]; a 16-byte chunk = one row of both planes.
//         Tile index (row >> 4) * (S >> 4) + (col >> 4) over the whole Stride x Height plane: a bijection of the reference's linear offsets.
//         Fetches EVERY row and column the window needs (half-pel neighbours included, by two phase bits of the hash): up to 9 row
//         pairs x 3 quadrant columns of luma, 9 rows x 2 columns of chroma; chunks a window does not touch are masked off.
//   ROWM  luma: a macroblock = 16 rows x 16 B row-major (a chunk = one row); chroma as QUAD.  17 rows x 2 chunks.
// Stores: LIN 16 x 128 B + 2 x 8 x 64 B rows at pitch 1024; tiled 2 KB + 1 KB contiguous.
// Second part: what an INTRA macroblock's store costs: 16 x 16 B + 2 x 8 x 8 B row pieces at pitch 1024 against 256 + 128 contiguous bytes,
// four macroblocks per wave at pseudo-random places.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
enum { S = 1024, LGS = 10, W = 640, H = 480, MBW = 40, MBH = 30, OPR = 5, OPC = 150 };
static const size_t YSZ = (size_t)S * H, SLOT = YSZ * 3 / 2, CLIP = SLOT * 2;
typedef const void __attribute__((address_space(1))) *gptr_t;
typedef void __attribute__((address_space(3))) *lptr_t;
#define DMA16(src, dst) __builtin_amdgcn_global_load_lds((gptr_t)(src), (lptr_t)(dst), 16, 0, 0)
__device__ __forceinline__ uint32_t hash(uint32_t mb) { uint32_t h = mb * 2654435761u; h ^= h >> 15; h *= 0x9E3779B1u; h ^= h >> 13; return h; }
// row / column parts of the tiled addresses (they add)
__device__ __forceinline__ uint32_t qy_row(int row) { return ((uint32_t)(row >> 4) << (LGS + 4)) | ((row & 8) << 4) | ((row & 7) << 3); }
__device__ __forceinline__ uint32_t qy_col(int col) { return ((uint32_t)(col >> 4) << 8) | ((col & 8) << 3) | (col & 7); }
__device__ __forceinline__ uint32_t ry_row(int row) { return ((uint32_t)(row >> 4) << (LGS + 4)) | ((row & 15) << 4); }
__device__ __forceinline__ uint32_t ry_col(int col) { return ((uint32_t)(col >> 4) << 8) | (col & 15); }
__device__ __forceinline__ uint32_t qc_row(int row) { return ((uint32_t)(row >> 3) << (LGS + 3)) | ((row & 7) << 4); }
__device__ __forceinline__ uint32_t qc_col(int x) { return ((uint32_t)(x >> 3) << 7) | (x & 7); }

// window position of a macroblock: its own place plus a vector of +-range pels, kept inside the picture
__device__ __forceinline__ void mbpos(uint32_t h, int range, int mby, int mbx, int &x, int &y) {
  const int dx = (int)(h % (2 * range + 1)) - range, dy = (int)((h >> 8) % (2 * range + 1)) - range;
  x = mbx * 16 + dx; y = mby * 16 + dy;
  x = x < 0 ? 0 : (x > W - 34 ? W - 34 : x);
  y = y < 0 ? 0 : (y > H - 18 ? H - 18 : y);
}
template <int V> // 0 LIN, 1 QUAD, 2 ROWM
__global__ __launch_bounds__(64) void fetch(uint8_t *planes, int n_clips, uint32_t per_xcd, int range) {
  __shared__ __attribute__((aligned(16))) uint8_t L[10240];
  const uint32_t oi = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (oi >= (uint32_t)OPC * n_clips) return;
  const uint32_t clip = oi / OPC, rem = oi % OPC, mby = rem / OPR, ox = rem % OPR;
  uint8_t *base = planes + (size_t)clip * CLIP;
  const uint8_t *ref = base + SLOT;
  const int lane = threadIdx.x;
  const int g = V == 0 ? (lane >> 1) & 7 : lane & 7;
  const uint32_t mb = (clip * MBH + mby) * MBW + ox * 8 + g;
  const uint32_t h = hash(mb);
  const int hp = (h >> 20) & 1, vp = (h >> 21) & 1;
  int x, y;
  mbpos(h, range, mby, ox * 8 + g, x, y);
  if (V == 0) {
    const int r = lane >> 4, hh = lane & 1;
    const int pos = y * S + x;
    const uint8_t *sy = ref + ((pos + r * S) & ~3) + hh * 16;
#pragma unroll
    for (int t = 0; t < 4; t++) DMA16(sy + 4 * t * S, L + t * 1024);
    const int cg = lane & 7, cr = lane >> 3;
    int cxx, cyy;
    mbpos(hash((clip * MBH + mby) * MBW + ox * 8 + cg), range, mby, ox * 8 + cg, cxx, cyy);
    const int cpos = (cyy >> 1) * S + (cxx >> 1);
    const uint8_t *sc = ref + YSZ + ((cpos + cr * S) & ~3);
    DMA16(sc, L + 4096);
    DMA16(sc + S / 2, L + 5120);
  } else {
    const int j = lane >> 3;
    // luma
    if (V == 1) {
      const int ncol = ((x & 7) + 16 + hp + 7) >> 3, npair = ((y & 1) + 16 + vp + 1) >> 1;
      const int slot = j & 3;
      const uint32_t lin = (uint32_t)((y & ~1) * S + (x & ~7) + 8 * slot); // a chunk never leaves its row: S is a multiple of 16
      const int row0 = (int)(lin >> LGS);
      const uint8_t *cb = ref + qy_col((int)(lin & (S - 1)));
#pragma unroll
      for (int t = 0; t < 5; t++) {
        const int pair = 2 * t + (j >> 2);
        if (slot < ncol && pair < npair) DMA16(cb + qy_row(row0 + 2 * pair), L + t * 1024);
      }
    } else {
      const int nrow = 16 + vp, slot = j & 1, r4 = j >> 1;
      const uint32_t lin = (uint32_t)(y * S + (x & ~15) + 16 * slot);
      const int row0 = (int)(lin >> LGS);
      const uint8_t *cb = ref + ry_col((int)(lin & (S - 1)));
#pragma unroll
      for (int t = 0; t < 5; t++) {
        const int row = 4 * t + r4;
        if (row < nrow) DMA16(cb + ry_row(row0 + row), L + t * 1024);
      }
    }
    // chroma: 10 rows x 2 columns of [U|V] chunks
    {
      const int cx = x >> 1, cy = y >> 1, chp = (h >> 22) & 1, cvp = (h >> 23) & 1;
      const int ncol = ((cx & 7) + 8 + chp + 7) >> 3, nrow = 8 + cvp;
      const int slot = j & 1;
      const int col = (cx & ~7) + 8 * slot; // (no wrap handling in the benchmark: the windows stay inside the picture)
      const uint8_t *cb = ref + YSZ + qc_col(col);
#pragma unroll
      for (int t = 0; t < 3; t++) {
        const int row = 4 * t + (j >> 1);
        if (slot < ncol && row < nrow) DMA16(cb + qc_row(cy + row), L + 5120 + t * 1024);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (V == 0) {
    const int off0 = mby * 16 * S + ox * 128;
#pragma unroll
    for (int it = 0; it < 2; it++) {
      const int i = lane + 64 * it, gq = i & 7, row16 = i >> 3;
      *(uint4 *)(base + off0 + row16 * S + gq * 16) = *(const uint4 *)(L + row16 * 256 + gq * 16);
      const int row = row16 & 7;
      *(uint2 *)(base + YSZ + (off0 >> 1) + it * (S >> 1) + row * S + gq * 8) = *(const uint2 *)(L + 4096 + it * 1024 + row * 128 + gq * 8);
    }
  } else {
    uint8_t *oy = base + ((size_t)(mby * (S >> 4) + ox * 8) << 8), *oc = base + YSZ + ((size_t)(mby * (S >> 4) + ox * 8) << 7);
    *(uint4 *)(oy + lane * 16) = *(const uint4 *)(L + lane * 16);
    *(uint4 *)(oy + 1024 + lane * 16) = *(const uint4 *)(L + 1024 + lane * 16);
    *(uint4 *)(oc + lane * 16) = *(const uint4 *)(L + 5120 + lane * 16);
  }
}
template <int V> float run(uint8_t *p, int n_clips, int range) {
  const uint32_t n = OPC * n_clips, grid = (n + 7) / 8 * 8;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float best = 1e9;
  for (int rep = 0; rep < 5; rep++) {
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(fetch<V>, dim3(grid), dim3(64), 0, 0, p, n_clips, grid / 8, range);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    if (rep && ms < best) best = ms;
  }
  return best;
}

// ---- intra-like stores: four macroblocks per wave, 16 lanes each -----------------------------------------------------
template <int TILED>
__global__ __launch_bounds__(64) void mbstore(uint8_t *planes, int n_clips, uint32_t seed, int every) {
  const int lane = threadIdx.x, l = lane & 15;
  // macroblock n of the launch: one in `every` macroblocks of the clips in raster order (a P-frame's intra macroblocks), or all of them
  const uint32_t n = blockIdx.x * 4 + (lane >> 4);
  const uint32_t idx = n * every + (hash(n + seed) % every);
  const uint32_t clip = idx / (MBW * MBH), mb = idx % (MBW * MBH);
  if (clip >= (uint32_t)n_clips) return;
  const int mby = mb / MBW, mbx = mb % MBW;
  uint8_t *base = planes + (size_t)clip * CLIP;
  const uint4 v = uint4{n, n, n, n};
  if (TILED) {
    uint8_t *oy = base + ((size_t)(mby * (S >> 4) + mbx) << 8), *oc = base + YSZ + ((size_t)(mby * (S >> 4) + mbx) << 7);
    *(uint4 *)(oy + l * 16) = v;
    if (l < 8) *(uint4 *)(oc + l * 16) = v;
  } else {
    const int off = mby * 16 * S + mbx * 16;
    *(uint4 *)(base + off + l * S) = v;
    *(uint2 *)(base + YSZ + (off >> 1) + (l >> 3) * (S >> 1) + (l & 7) * S) = uint2{n, n};
  }
}
template <int TILED> float run_store(uint8_t *p, int n_clips, int every) {
  const uint32_t n_mb = (uint32_t)((size_t)n_clips * MBW * MBH / every), grid = n_mb / 4;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float best = 1e9;
  for (int rep = 0; rep < 4; rep++) {
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(mbstore<TILED>, dim3(grid), dim3(64), 0, 0, p, n_clips, (uint32_t)rep * 9973u, every);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    if (rep && ms < best) best = ms;
  }
  printf("%s  1 macroblock in %2d  %8u macroblocks  %.3f ms  %.2f macroblocks/ns\n", TILED ? "tiled 256+128 B contiguous " : "linear 16x16 B + 16x8 B rows", every, n_mb, best, n_mb / best / 1e6);
  return best;
}
int main(int argc, char **argv) {
  const int n_clips = argc > 1 ? atoi(argv[1]) : 4096;
  uint8_t *p; if (hipMalloc(&p, CLIP * n_clips + 8192) != hipSuccess) return 1;
  (void)hipMemset(p, 1, CLIP * n_clips + 8192);
  p += 4096;
  const double bytes = 3.0 * W * H * n_clips;
  const char *names[] = {"LIN  linear pitch 1024 (fetchpat V1)", "QUAD 8x8 quadrants, [U|V] rows", "ROWM 16x16 row-major, [U|V] rows"};
  for (int range : {0, 8, 16, 32}) {
    float ms[3] = {run<0>(p, n_clips, range), run<1>(p, n_clips, range), run<2>(p, n_clips, range)};
    for (int v = 0; v < 3; v++) printf("mv +-%2d  %-38s %d clips  %.3f ms  %.2f TB/s algorithmic\n", range, names[v], n_clips, ms[v], bytes / ms[v] / 1e9);
  }
  for (int every : {20, 1}) { run_store<0>(p, n_clips, every); run_store<1>(p, n_clips, every); }
  return 0;
}
