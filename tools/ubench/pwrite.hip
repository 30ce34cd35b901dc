// Micro-benchmark (r02): what does a PARTIAL-LINE store cost?  An intra macroblock is 16 rows x 16 B of luma (+ 2 x 8 rows x 8 B of
// chroma) in planes of pitch 1024: 32 stores into 32 different 128-byte lines whose other bytes belong to macroblocks written by
// another launch.  Every wave here writes `rows` row pieces of W bytes at pseudo-random places of an 8 GiB buffer (row r of a piece
// group at +r * 1024); W = 8..128.  Reports pieces per ns and GB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
template <int W, int TOUCH = 0> // bytes per row piece; lanes per piece = max(1, W / 16); TOUCH: load the own bytes first (brings the line into the L2)
__global__ __launch_bounds__(64) void pw(uint8_t *buf, size_t mask, uint32_t seed, int cluster) {
  constexpr int LPP = W >= 16 ? W / 16 : 1;           // lanes per piece
  const int lane = threadIdx.x, piece = lane / LPP, sub = lane % LPP; // 64 / LPP pieces per wave
  const uint32_t grp = blockIdx.x * (64 / LPP / 16 ? 64 / LPP / 16 : 1) + piece / 16; // a group = 16 rows
  uint32_t h = (grp + seed) * 2654435761u; h ^= h >> 13; h *= 0x9E3779B1u; h ^= h >> 16;
  size_t base = (((size_t)h << 10) & mask) + ((h >> 22) & 7) * 128; // somewhere, 128-byte aligned
  if (cluster) { // 2^cluster consecutive groups share a 2 MiB page (an intra kernel's consecutive macroblocks belong to one clip)
    uint32_t hp = ((grp >> cluster) + seed) * 2654435761u; hp ^= hp >> 13; hp *= 0x9E3779B1u; hp ^= hp >> 16;
    base = ((((size_t)hp << 21) & mask) | (base & ((1u << 21) - 1))) & mask;
  }
  uint8_t *p = buf + base + (size_t)(piece & 15) * 1024 + sub * 16;
  if (TOUCH) { const uint32_t t = *(const volatile uint32_t *)p; h += t & 1; }
  if (W >= 16) *(uint4 *)p = uint4{h, h, h, h};
  else if (W == 8) *(uint2 *)p = uint2{h, h};
  else *(uint32_t *)p = h;
}
static int g_cluster = 0;
template <int W, int TOUCH = 0> void run(uint8_t *buf, size_t mask, int waves) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float best = 1e9;
  for (int rep = 0; rep < 4; rep++) {
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL((pw<W, TOUCH>), dim3(waves), dim3(64), 0, 0, buf, mask, (uint32_t)rep * 77777u, g_cluster);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    if (rep && ms < best) best = ms;
  }
  constexpr int LPP = W >= 16 ? W / 16 : 1;
  const double pieces = (double)waves * (64 / LPP);
  printf("%s W=%3d B  %8.0f k pieces  %.3f ms  %.1f pieces/ns  %.0f GB/s\n", TOUCH ? "load+store" : "store     ", W, pieces / 1e3, best, pieces / best / 1e6, pieces * W / best / 1e6);
}
int main(int argc, char **argv) {
  const int waves = argc > 1 ? atoi(argv[1]) : 400000;
  const size_t bytes = (size_t)(argc > 2 ? atoi(argv[2]) : 8192) << 20; // footprint in MiB (a power of two): 8 GiB = nothing is reused; 64 MiB = inside the 256 MB Infinity Cache
  g_cluster = argc > 3 ? atoi(argv[3]) : 0;
  uint8_t *buf; if (hipMalloc(&buf, bytes + (1 << 20)) != hipSuccess) return 1;
  printf("footprint %zu MiB, %d waves, 2^%d consecutive groups per 2 MiB page\n", bytes >> 20, waves, g_cluster);
  (void)hipMemset(buf, 0, bytes);
  const size_t mask = bytes - (32 << 10) - 1;
  run<4>(buf, mask, waves); run<8>(buf, mask, waves); run<16>(buf, mask, waves); run<32>(buf, mask, waves); run<64>(buf, mask, waves); run<128>(buf, mask, waves);
  run<8, 1>(buf, mask, waves); run<16, 1>(buf, mask, waves); run<32, 1>(buf, mask, waves); run<64, 1>(buf, mask, waves);
  return 0;
}
