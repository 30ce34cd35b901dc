// mobi_rgb.hip -- the Bitmap that MobiclipDecoder.DecodeFrame() returns (MD.cs:260-323), on the GPU.
//
// Per pixel: Y, plus U and V averaged from up to four chroma neighbours chosen by the pixel's parity (not on the
// last column / last row, MD.cs:269), then either the float BT.601-like matrix with the 16..255 range stretch
// (Moflex3DS, :297-305) or the integer Y+U-V / Y+V / Y-U-V form on truncated values (ModsDS, :306-311), clamp,
// truncate, pack as 0xAARRGGBB (:313-319).  HBM-bound: 1.5 bytes read, 4 written per pixel.
//
// Float semantics are the reference's: IEEE single, one rounding per C# operator in source order.  Hence contraction is switched
// off where the arithmetic is (hipcc contracts a*b+c into an FMA by default, which rounds once instead of twice).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mobi_kernels.h"
#include "mobi_tile.h"

namespace {
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
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
// The same on two pixels at once: gfx950's packed single-precision instructions (v_pk_mul_f32, v_pk_add_f32, v_pk_fma_f32) round each
// half exactly as the scalar ones do, at twice the rate.  Contraction is OFF in these functions: a product and a sum are two roundings
// (the reference's), and only what is written as an fma is one.
#pragma clang fp contract(off)
__device__ __forceinline__ f32x2 div239_2(f32x2 x) {
  const f32x2 r = {1.0f / 239.0f, 1.0f / 239.0f}, m = {-239.0f, -239.0f};
  const f32x2 q0 = x * r;
  return __builtin_elementwise_fma(__builtin_elementwise_fma(m, q0, x), r, q0);
}
__device__ __forceinline__ f32x2 stretch2(f32x2 c) { // (c - 16f) * 255f / (255f - 16f), MD.cs:303-305
  const f32x2 k16 = {16.f, 16.f}, k255 = {255.f, 255.f};
  return div239_2((c - k16) * k255);
}
// clamp to [0, 255], truncate, place in byte `pos` of `old` (MD.cs:313-319).  v_cvt_pk_u8_f32 saturates to [0, 255] but rounds to
// nearest; behind a floor it has nothing left to round, and below zero floor and truncation differ only where both saturate to 0:
// identical to the comparison chain for every float that is not a NaN (tools/ubench/cvtpk.hip, all 2^32 patterns; alone it differs
// for 41.9 M of them, 0.5000001 first).  Two instructions per output byte instead of five.
template <int POS>
__device__ __forceinline__ uint32_t put_u8(float x, uint32_t old) {
  uint32_t d = old;
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_cvt_pk_u8_f32 %0, %1, %2, %0" : "+v"(d) : "v"(__builtin_floorf(x)), "n"(POS));
#endif
  return d;
}
__device__ __forceinline__ int clamp255(int x) { return x < 0 ? 0 : x > 255 ? 255 : x; }
// two pixels: luma bytes y0, y1 (as floats), chroma numerators in quarter samples (see the kernel)
__device__ __forceinline__ void convert2(int version, f32x2 Y2, int un0, int un1, int vn0, int vn1, uint32_t &p0, uint32_t &p1) {
  if (version == 2) { // Moflex3DS: the float matrix and the 16..255 stretch (MD.cs:297-305)
    const f32x2 q = {0.25f, 0.25f};
    const f32x2 U = f32x2{(float)un0, (float)un1} * q, V = f32x2{(float)vn0, (float)vn1} * q; // exact: small integers, a power of two
    const f32x2 kRV = {1.420f, 1.420f}, kGU = {0.344f, 0.344f}, kGV = {0.714f, 0.714f}, kBU = {1.772f, 1.772f};
    const f32x2 R = stretch2(Y2 + kRV * V);
    const f32x2 G = stretch2((Y2 - kGU * U) - kGV * V);
    const f32x2 B = stretch2(Y2 + kBU * U);
    p0 = put_u8<2>(R.x, put_u8<1>(G.x, put_u8<0>(B.x, 0xFF000000u))); // Color.FromArgb(r, g, b).ToArgb()
    p1 = put_u8<2>(R.y, put_u8<1>(G.y, put_u8<0>(B.y, 0xFF000000u)));
    return;
  }
  // ModsDS: (int) casts truncate toward zero; y + u - v, y + v, y - u - v on the truncated values (MD.cs:306-311)
  auto tz = [](int n) { return n >= 0 ? n >> 2 : -((-n) >> 2); };
  const int u0 = tz(un0), u1 = tz(un1), v0 = tz(vn0), v1 = tz(vn1), y0 = (int)Y2.x, y1 = (int)Y2.y;
  p0 = 0xFF000000u | ((uint32_t)clamp255(y0 + u0 - v0) << 16) | ((uint32_t)clamp255(y0 + v0) << 8) | (uint32_t)clamp255(y0 - u0 - v0);
  p1 = 0xFF000000u | ((uint32_t)clamp255(y1 + u1 - v1) << 16) | ((uint32_t)clamp255(y1 + v1) << 8) | (uint32_t)clamp255(y1 - u1 - v1);
}
__device__ __forceinline__ uint32_t lane_right(uint32_t v) { // the value of the lane to the right (lane + 1) inside a row of 16 lanes
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xF, 0xF, false); // row_shl:1
#else
  return v;
#endif
}
} // namespace

// One wave = two horizontally adjacent macroblocks (32 x 16 pixels); one lane = 4 pixels of two rows: lane & 7 picks the column group,
// lane >> 3 the row pair.  The planes are macroblock tiles (mobi_tile.h): a lane's two luma rows are 4 + 4 bytes of one 16-byte chunk, its
// chroma samples two bytes of one or two tile rows; a row of the wave's output is 128 contiguous bytes.
//
// Chroma (MD.cs:262-296): a pixel takes the sample under it, or -- not in the picture's last column or last row -- the mean of that
// sample and its right / lower / three neighbours, by the pixel's parity.  The samples are bytes minus 128 and the means divide by 2 or 4,
// so every intermediate float of the reference is an exact small multiple of 1/4: the numerators are added as integers (4a, 2(a + b),
// a + b + c + d, minus 512) and one exact multiplication by 0.25 gives the float the reference's additions and division give.
// r04: 29 vector instructions per pixel instead of 65 (two pixels per floating-point instruction, one conversion instruction per output
// byte instead of five, half the chroma loads per pixel) -- tools/exp_rgb.py.
extern "C" __global__ __launch_bounds__(64) void mobi_yuv_to_argb(const uint8_t *planes, uint64_t clip_bytes, uint32_t slot_bytes, int ring_base,
                                                                 int width, int height, int stride, int version, int clip0, uint32_t *out) {
  const int mbw = width >> 4, pairs = (mbw + 1) >> 1;
  const int mby = blockIdx.x / pairs, pr = blockIdx.x - mby * pairs, clip = blockIdx.z;
  const int lane = threadIdx.x, cg = lane & 7, rp = lane >> 3;
  const int mbx = 2 * pr + (cg >> 2), x0 = mbx * 16 + (cg & 3) * 4, y0 = mby * 16 + 2 * rp;
  const bool active = mbx < mbw; // (an odd number of macroblocks per row: the last wave of a row has one)
  const uint8_t *Y = planes + (size_t)(clip0 + clip) * clip_bytes + (size_t)ring_base * slot_bytes;
  const uint8_t *UV = Y + (size_t)stride * height;
  const int lgS = 31 - __builtin_clz((unsigned)stride);
  const uint32_t mbxa = active ? (uint32_t)mbx : 0u;
  // luma: rows y0, y0 + 1 are the two rows of one chunk
  const uint8_t *yp = Y + mobi_tile_y(mbxa, (uint32_t)mby, lgS) + (((rp >> 2) * 2 + ((cg & 3) >> 1)) << 6) + (((2 * rp) & 7) << 3) + ((cg & 1) << 2);
  const uint32_t yw0 = *(const uint32_t *)yp, yw1 = *(const uint32_t *)(yp + 8);
  // chroma rows y0 / 2 and, unless the lane's odd row is the picture's last, the one below (for rp == 7: in the tile below)
  const bool lastrow = y0 + 2 >= height;
  const bool lastcol = x0 + 4 >= width;
  const uint8_t *c0p = UV + mobi_tile_c(mbxa, (uint32_t)mby, lgS) + (rp << 4) + ((cg & 3) << 1);
  const uint8_t *c1p = lastrow ? c0p : rp < 7 ? c0p + 16 : UV + mobi_tile_c(mbxa, (uint32_t)mby + 1u, lgS) + ((cg & 3) << 1);
  const uint32_t u0w = *(const uint16_t *)c0p, v0w = *(const uint16_t *)(c0p + 8), u1w = *(const uint16_t *)c1p, v1w = *(const uint16_t *)(c1p + 8);
  // the sample to the right of the lane's two: the next lane's first one -- except for the wave's last column group, whose right
  // neighbour is the next pair's first macroblock (not looked at in the picture's last column)
  uint32_t ue0 = lane_right(u0w), ve0 = lane_right(v0w), ue1 = lane_right(u1w), ve1 = lane_right(v1w);
  if (cg == 7 && !lastcol) {
    const int d0 = (int)(mobi_tile_c((uint32_t)mbx + 1u, 0, lgS)) - (int)(mobi_tile_c((uint32_t)mbx, 0, lgS)) - 6; // first sample of the next tile's row, seen from this lane's pair
    ue0 = c0p[d0]; ve0 = c0p[d0 + 8]; ue1 = c1p[d0]; ve1 = c1p[d0 + 8];
  }
  // numerators in quarter samples, minus 4 * 128: even row: 4a, 2(a + b), 4b, 2(b + e); odd row: 2(a + a'), a + b + a' + b', 2(b + b'), b + e + b' + e'
  auto numerators = [&](uint32_t w0, uint32_t e0w, uint32_t w1, uint32_t e1w, int (&ev)[4], int (&od)[4]) {
    const int a0 = (int)(w0 & 0xFF), b0 = (int)(w0 >> 8) & 0xFF, e0 = (int)(e0w & 0xFF), a1 = (int)(w1 & 0xFF), b1 = (int)(w1 >> 8) & 0xFF, e1 = (int)(e1w & 0xFF);
    const int pa = 4 * a0 - 512, pb = 4 * b0 - 512; // the plain sample
    ev[0] = pa; ev[1] = 2 * (a0 + b0) - 512; ev[2] = pb; ev[3] = lastcol ? pb : 2 * (b0 + e0) - 512;
    od[0] = 2 * (a0 + a1) - 512; od[1] = a0 + b0 + a1 + b1 - 512; od[2] = 2 * (b0 + b1) - 512; od[3] = lastcol ? pb : b0 + e0 + b1 + e1 - 512;
    if (lastrow) { od[0] = pa; od[1] = pa; od[2] = pb; od[3] = pb; } // the lane's odd row is the picture's last: no mean of any kind (MD.cs:269)
  };
  int ue[4], uo[4], ve[4], vo[4];
  numerators(u0w, ue0, u1w, ue1, ue, uo);
  numerators(v0w, ve0, v1w, ve1, ve, vo);
  uint32_t pe[4], po[4];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const f32x2 ye = {(float)((yw0 >> (16 * k)) & 0xFF), (float)((yw0 >> (16 * k + 8)) & 0xFF)};
    const f32x2 yo = {(float)((yw1 >> (16 * k)) & 0xFF), (float)((yw1 >> (16 * k + 8)) & 0xFF)};
    convert2(version, ye, ue[2 * k], ue[2 * k + 1], ve[2 * k], ve[2 * k + 1], pe[2 * k], pe[2 * k + 1]);
    convert2(version, yo, uo[2 * k], uo[2 * k + 1], vo[2 * k], vo[2 * k + 1], po[2 * k], po[2 * k + 1]);
  }
  if (active) {
    uint32_t *o = out + ((size_t)clip * height + y0) * width + x0;
    // written once, read by nobody on this GPU: past the caches (0.222 -> 0.215 ms per 512 clips of 640x480, A/B on one box)
    __builtin_nontemporal_store(u32x4{pe[0], pe[1], pe[2], pe[3]}, (u32x4 *)o);
    __builtin_nontemporal_store(u32x4{po[0], po[1], po[2], po[3]}, (u32x4 *)(o + width));
  }
}

extern "C" int mobi_launch_argb(const MobiReconArgs *a, int version, int clip0, int n_clips, uint32_t *out_dev, hipStream_t s) {
  if (n_clips <= 0) return 0;
  const int mbw = a->mbw, pairs = (mbw + 1) / 2;
  const dim3 grid((unsigned)(pairs * (a->n_mbs / mbw)), 1u, (unsigned)n_clips);
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
