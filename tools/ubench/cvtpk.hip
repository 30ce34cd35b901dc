// v_cvt_pk_u8_f32, alone and behind v_floor_f32, against "clamp to [0, 255], then (int) toward zero" (MD.cs:313-319) for every float
// bit pattern that is not a NaN (mobi_rgb.hip, put_u8).
// hipcc --offload-arch=gfx950 -O2 tools/ubench/cvtpk.hip -o tools/ubench/cvtpk.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ __forceinline__ uint32_t cvt_pk_u8(float x) {
  uint32_t d = 0;
  asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %2" : "=v"(d) : "v"(x), "v"(0u));
  return d & 0xFFu;
}
__global__ void check(unsigned long long *bad, uint32_t *first, int with_floor) {
  const uint32_t base = (blockIdx.x * 256u + threadIdx.x) * 256u;
  unsigned long long n = 0;
  for (uint32_t k = 0; k < 256; k++) {
    const float x = __uint_as_float(base + k);
    if (x != x) continue;
    float c = x < 0.f ? 0.f : x;
    c = c > 255.f ? 255.f : c;
    const uint32_t want = (uint32_t)(int)c;
    if (cvt_pk_u8(with_floor ? floorf(x) : x) != want) { if (!n) atomicMin(first, base + k); n++; }
  }
  if (n) atomicAdd(bad, n);
}
int main() {
  for (int with_floor = 0; with_floor < 2; with_floor++) {
    unsigned long long *bad, h = 0; uint32_t *first, hf = 0xFFFFFFFFu;
    (void)hipMalloc(&bad, 8); (void)hipMalloc(&first, 4); (void)hipMemset(bad, 0, 8); (void)hipMemcpy(first, &hf, 4, hipMemcpyHostToDevice);
    check<<<65536, 256>>>(bad, first, with_floor);
    (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
    float f; memcpy(&f, &hf, 4);
    printf("%s vs clamp + truncate: %llu of 2^32 patterns differ", with_floor ? "v_floor_f32 + v_cvt_pk_u8_f32" : "v_cvt_pk_u8_f32", h);
    if (h) printf(" (first: 0x%08x = %g)", hf, f);
    printf("\n");
  }
  return 0;
}
