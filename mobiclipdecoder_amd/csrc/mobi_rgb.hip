// mobi_rgb.hip -- the Bitmap that MobiclipDecoder.DecodeFrame() returns (MD.cs:260-323), on the GPU.
//
// Per pixel: Y, plus U and V averaged from up to four chroma neighbours chosen by the pixel's parity (not on the
// last column / last row, MD.cs:269), then either the float BT.601-like matrix with the 16..255 range stretch
// (Moflex3DS, :297-305) or the integer Y+U-V / Y+V / Y-U-V form on truncated values (ModsDS, :306-311), clamp,
// truncate, pack as 0xAARRGGBB (:313-319).  HBM-bound: 1.5 bytes read, 4 written per pixel.
//
// Float semantics are the reference's: IEEE single, one rounding per C# operator in source order.  Hence the
// __f*_rn intrinsics throughout: hipcc contracts a*b+c into an FMA by default, which rounds once instead of twice.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mobi_kernels.h"
#include "mobi_tile.h"

namespace {
// x / 239f, correctly rounded, in three instructions instead of the ~10 of a generic IEEE division: q0 = x*r,
// q = fma(fma(-239, q0, x), r, q0) with r = RN(1/239).  Not a theorem for every divisor: it is CHECKED for this one,
// exhaustively over all 2^32 bit patterns (mobi_selftest_div239, tests/test_rgb.py): identical to __fdiv_rn for every
// float with 1e-30 <= |x| <= 1e30, and for x = 0 up to the sign of zero, which the clamp and the int cast discard.
// (The stretch's numerator is (c - 16) * 255 with |c| < 1000: zero, or at least 1e-4 in magnitude.)
__device__ __forceinline__ float div239(float x) {
  const float r = 1.0f / 239.0f;
  const float q0 = __fmul_rn(x, r);
  return __fmaf_rn(__fmaf_rn(-239.0f, q0, x), r, q0);
}
__device__ __forceinline__ uint32_t pack_argb(float R, float G, float B) {
  R = R < 0.f ? 0.f : R; R = R > 255.f ? 255.f : R; // :313-318
  G = G < 0.f ? 0.f : G; G = G > 255.f ? 255.f : G;
  B = B < 0.f ? 0.f : B; B = B > 255.f ? 255.f : B;
  return 0xFF000000u | ((uint32_t)(int)R << 16) | ((uint32_t)(int)G << 8) | (uint32_t)(int)B; // Color.FromArgb(r, g, b).ToArgb()
}
__device__ __forceinline__ uint32_t convert_px(int version, float Y2, float U, float V) {
  if (version == 2) { // Moflex3DS
    float R = __fadd_rn(Y2, __fmul_rn(1.420f, V));
    float G = __fsub_rn(__fsub_rn(Y2, __fmul_rn(0.344f, U)), __fmul_rn(0.714f, V));
    float B = __fadd_rn(Y2, __fmul_rn(1.772f, U));
    R = div239(__fmul_rn(__fsub_rn(R, 16.f), 255.f)); // (255f - 16f) is a constant
    G = div239(__fmul_rn(__fsub_rn(G, 16.f), 255.f));
    B = div239(__fmul_rn(__fsub_rn(B, 16.f), 255.f));
    return pack_argb(R, G, B);
  }
  const int y = (int)Y2, u = (int)U, v = (int)V; // ModsDS: casts truncate toward zero
  return pack_argb((float)(y + u - v), (float)(y + v), (float)(y - u - v));
}
} // namespace

// one lane = 4 horizontally adjacent pixels (one 16-byte store); one wave = one macroblock, lane = (row lane >> 2, pixels 4 * (lane & 3)):
// the planes are macroblock tiles (mobi_tile.h), so a wave reads its 256 luma bytes as four whole 64-byte pieces (r02's "one block =
// 1024 pixels of a row" would touch 8 bytes of every piece it reads)
extern "C" __global__ __launch_bounds__(64) void mobi_yuv_to_argb(const uint8_t *planes, uint64_t clip_bytes, uint32_t slot_bytes, int ring_base,
                                                                 int width, int height, int stride, int version, int clip0, uint32_t *out) {
  const int mbw = width >> 4, mby = blockIdx.x / mbw, mbx = blockIdx.x - mby * mbw, clip = blockIdx.z;
  const int lane = threadIdx.x, x0 = mbx * 16 + (lane & 3) * 4, y = mby * 16 + (lane >> 2);
  const uint8_t *Y = planes + (size_t)(clip0 + clip) * clip_bytes + (size_t)ring_base * slot_bytes;
  const uint8_t *UV = Y + (size_t)stride * height;
  const int S = stride, lgS = 31 - __builtin_clz((unsigned)stride);
  // every access names the reference's linear offset and is mapped
  const uint32_t yw = *(const uint32_t *)(Y + mobi_ty((uint32_t)(y * S + x0), lgS));
  const int c = (y >> 1) * S + (x0 >> 1);
  const bool lastrow = y == height - 1, odd = (y & 1) != 0, vert = odd && !lastrow;
  // chroma samples this lane may touch: columns c .. c+2 of this chroma row and, for odd luma rows, of the next one.
  // c is even: samples c, c+1 and c+2, c+3 are two 2-byte pieces, each inside one 8-sample tile row (c + 3 < Stride/2).
  float u[2][3], v[2][3];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const bool need = r == 0 || vert;
    uint32_t uw = 0, vw = 0;
    if (need) {
      const uint32_t t0 = mobi_tc((uint32_t)(c + r * S), lgS), t1 = mobi_tc((uint32_t)(c + 2 + r * S), lgS); // U half; V = + 8 in the tile row
      uw = (uint32_t) * (const uint16_t *)(UV + t0) | ((uint32_t) * (const uint16_t *)(UV + t1) << 16);
      vw = (uint32_t) * (const uint16_t *)(UV + t0 + 8) | ((uint32_t) * (const uint16_t *)(UV + t1 + 8) << 16);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
      u[r][k] = __fsub_rn((float)((uw >> (8 * k)) & 0xFF), 128.f);
      v[r][k] = __fsub_rn((float)((vw >> (8 * k)) & 0xFF), 128.f);
    }
  }
  uint32_t px[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int x = x0 + i, k = i >> 1;
    float U = u[0][k], V = v[0][k];
    if (x != width - 1 && !lastrow) { // MD.cs:269
      const bool h = (x & 1) != 0;
      if (h && !odd) { // case 1
        U = __fadd_rn(U, u[0][k + 1]); V = __fadd_rn(V, v[0][k + 1]);
        U = __fmul_rn(U, 0.5f); V = __fmul_rn(V, 0.5f); // == U / 2f exactly (power of two)
      } else if (!h && odd) { // case 2
        U = __fadd_rn(U, u[1][k]); V = __fadd_rn(V, v[1][k]);
        U = __fmul_rn(U, 0.5f); V = __fmul_rn(V, 0.5f);
      } else if (h && odd) { // case 3: +1, +Stride, +1+Stride in this order
        U = __fadd_rn(U, u[0][k + 1]); V = __fadd_rn(V, v[0][k + 1]);
        U = __fadd_rn(U, u[1][k]); V = __fadd_rn(V, v[1][k]);
        U = __fadd_rn(U, u[1][k + 1]); V = __fadd_rn(V, v[1][k + 1]);
        U = __fmul_rn(U, 0.25f); V = __fmul_rn(V, 0.25f);
      }
    }
    px[i] = convert_px(version, (float)((yw >> (8 * i)) & 0xFF), U, V);
  }
  *(uint4 *)(out + ((size_t)clip * height + y) * width + x0) = uint4{px[0], px[1], px[2], px[3]};
}

extern "C" int mobi_launch_argb(const MobiReconArgs *a, int version, int clip0, int n_clips, uint32_t *out_dev, hipStream_t s) {
  if (n_clips <= 0) return 0;
  const dim3 grid((unsigned)a->n_mbs, 1u, (unsigned)n_clips);
  hipLaunchKernelGGL(mobi_yuv_to_argb, grid, dim3(64), 0, s, (const uint8_t *)a->planes, (uint64_t)a->clip_bytes, a->slot_bytes, a->ring_base,
                     a->width, a->height, a->stride, version, clip0, out_dev);
  return (int)hipGetLastError();
}

// ---- self-test: div239 against the correctly rounded division for every float bit pattern ------------------
extern "C" __global__ void mobi_div239_check(unsigned long long *bad) {
  const uint32_t base = (blockIdx.x * 256u + threadIdx.x) * 256u;
  unsigned long long n = 0;
  for (uint32_t k = 0; k < 256; k++) {
    const float x = __uint_as_float(base + k);
    if (!(fabsf(x) >= 1e-30f && fabsf(x) <= 1e30f)) continue; // also drops NaN
    if (__float_as_uint(div239(x)) != __float_as_uint(__fdiv_rn(x, 239.0f))) n++;
  }
  if (n) atomicAdd(bad, n);
}
extern "C" long long mobi_launch_div239_check(hipStream_t s) {
  unsigned long long *bad = nullptr, h = 0;
  if (hipMalloc((void **)&bad, 8) != hipSuccess) return -1;
  (void)hipMemsetAsync(bad, 0, 8, s);
  hipLaunchKernelGGL(mobi_div239_check, dim3(65536), dim3(256), 0, s, bad);
  const bool ok = hipMemcpyAsync(&h, bad, 8, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
  (void)hipFree(bad);
  return ok ? (long long)h : -1;
}
