// Calibration for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950: copy a known number of bytes with
// 4 B/lane and 16 B/lane accesses (MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 of a 16 B/lane stream;
// "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count").
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void copy4(const uint32_t *a, uint32_t *b, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) b[i] = a[i]; }
__global__ void copy16(const uint4 *a, uint4 *b, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) b[i] = a[i]; }
// 16 rows x 16 B fragments at a 1 KiB pitch per wave: the reconstruction kernels' own store/load shape
__global__ void copy_rows(const uint8_t *a, uint8_t *b, size_t n_mb, int mbw, int stride) {
  size_t mb = blockIdx.x * (size_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (mb >= n_mb) return;
  int lane = threadIdx.x & 63;
  size_t off = (mb / mbw) * (size_t)16 * stride + (mb % mbw) * 16 + (size_t)(lane >> 2) * stride + (lane & 3) * 4;
  *(uint32_t *)(b + off) = *(const uint32_t *)(a + off);
}
int main() {
  const size_t bytes = (size_t)512 << 20;
  uint8_t *a, *b; (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 0, bytes);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(copy4, dim3(bytes / 4 / 256), dim3(256), 0, 0, (const uint32_t *)a, (uint32_t *)b, bytes / 4);
    hipLaunchKernelGGL(copy16, dim3(bytes / 16 / 256), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, bytes / 16);
    // 1024-byte pitch, 640 used: 40 MBs per row, rows of MBs = bytes / (16*1024)
    const int stride = 1024, mbw = 40; const size_t n_mb = (bytes / (16 * (size_t)stride)) * mbw;
    hipLaunchKernelGGL(copy_rows, dim3((n_mb + 3) / 4), dim3(256), 0, 0, a, b, n_mb, mbw, stride);
  }
  (void)hipDeviceSynchronize();
  printf("copy4/copy16: %zu bytes read + %zu written per launch; copy_rows: %zu bytes each way\n", bytes, bytes, (bytes / (16 * (size_t)1024)) * 40 * 256);
  return 0;
}
