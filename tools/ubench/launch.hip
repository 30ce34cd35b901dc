// Micro-benchmark: how fast can gfx950 launch short-lived workgroups as a function of workgroup size and
// static LDS size?  Each wave loads one uint4 (a "descriptor") and exits.  614,400 waves per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int THREADS, int LDS_WORDS, bool BARRIER>
__global__ __launch_bounds__(THREADS) void k(const uint4 *desc, uint32_t *sink, int n_waves) {
  __shared__ uint32_t lds[LDS_WORDS > 0 ? LDS_WORDS : 1];
  const int wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * (THREADS / 64) + wave;
  if (gw >= n_waves) return;
  uint4 d = desc[gw];
  if (LDS_WORDS > 0) lds[threadIdx.x % LDS_WORDS] = d.x;
  if (BARRIER) __syncthreads();
  if (d.y == 0x12345678u) sink[0] = d.x + (LDS_WORDS > 0 ? lds[0] : 0);
}
template <int THREADS, int LDS_WORDS, bool BARRIER>
void run(const uint4 *desc, uint32_t *sink, int n_waves, const char *tag) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int grid = (n_waves + THREADS / 64 - 1) / (THREADS / 64);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<THREADS, LDS_WORDS, BARRIER>), dim3(grid), dim3(THREADS), 0, 0, desc, sink, n_waves);
  hipEventRecord(a, 0);
  for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k<THREADS, LDS_WORDS, BARRIER>), dim3(grid), dim3(THREADS), 0, 0, desc, sink, n_waves);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-34s threads=%4d lds=%6d B barrier=%d : %.4f ms per launch (%d WGs)\n", tag, THREADS, LDS_WORDS * 4, (int)BARRIER, ms / 20, grid);
}
int main() {
  const int n_waves = 614400;
  uint4 *desc; uint32_t *sink;
  hipMalloc(&desc, sizeof(uint4) * n_waves); hipMemset(desc, 0, sizeof(uint4) * n_waves); hipMalloc(&sink, 64);
  run<64, 0, false>(desc, sink, n_waves, "1 wave, no LDS");
  run<64, 864, false>(desc, sink, n_waves, "1 wave, 3.4 KB");
  run<256, 0, false>(desc, sink, n_waves, "4 waves, no LDS");
  run<256, 3456, false>(desc, sink, n_waves, "4 waves, 13.8 KB");
  run<256, 3456, true>(desc, sink, n_waves, "4 waves, 13.8 KB, barrier");
  run<256, 7776, false>(desc, sink, n_waves, "4 waves, 31 KB");
  run<512, 0, false>(desc, sink, n_waves, "8 waves, no LDS");
  run<512, 768, false>(desc, sink, n_waves, "8 waves, 3 KB");
  run<512, 3456, false>(desc, sink, n_waves, "8 waves, 13.8 KB");
  run<512, 7776, false>(desc, sink, n_waves, "8 waves, 31 KB");
  run<512, 7776, true>(desc, sink, n_waves, "8 waves, 31 KB, barrier");
  run<1024, 0, false>(desc, sink, n_waves, "16 waves, no LDS");
  return 0;
}
