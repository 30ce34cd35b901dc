// mobi_kernels.hip -- gfx950 (MI355X / CDNA4) reconstruction kernels for the command lists of mobi_cmd.h.
//
//   mobi_recon_inter8  : the inter macroblocks of a frame step, one wavefront per OCTET of eight horizontally adjacent
//                        macroblocks: half-pel truncating motion compensation from the reference planes (CopyBlock,
//                        MD.cs:418-456), dequant + 8x8/4x4 integer inverse transforms + clamp-add of the residual
//                        (MD.cs:3424-3429, :3435-3798), the octet's tiles stored as 2 KB + 1 KB runs.
//   mobi_recon_intra   : intra macroblocks (I-frames and codes 6/7 inside P-frames), four per wavefront (16 lanes each): halo load
//   mobi_recon_intra_cl  with raster-order availability masking, predictors (MD.cs:1883-2774, :3017-3327) and
//                        residuals in decode order inside LDS; all dependency levels of a step in one launch, ordered by
//                        per-macroblock completion tags.
//   mobi_untile        : a frame out of the private tiled planes (mobi_tile.h) into the reference's row-major arrays.
//
// The planes are stored as macroblock tiles (mobi_tile.h, r03): every position in the command list is still the reference's linear
// byte offset; the kernels map it.  8-bit pel work is memory traffic and integer issue: no MFMA.  LDS use is per wave (no workgroup
// barriers): waves never share LDS data, so a wavefront-scope fence (a pure compiler barrier -- LDS executes a wave's instructions
// in order) is all that separates producer and consumer lanes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "mobi_cmd.h"
#include "mobi_kernels.h"
#include "mobi_recon_math.h"
#include "mobi_tile.h"

namespace {

enum { TP = 32 };                     // intra tile pitch: interior col c at byte 4+c, halo col -1 at byte 3

__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

struct Geo { // stride is 256/512/1024 (MD.cs:50-52): divide/modulo by shifts
  int width, height, stride, mbw, lg;
  __device__ __forceinline__ int owner_luma(int a) const {
    if (a < 0) return -1;
    const int row = a >> lg, col = a & (stride - 1);
    if (col >= width || row >= height) return -1;
    return (row >> 4) * mbw + (col >> 4);
  }
  __device__ __forceinline__ int owner_chroma(int a) const {
    if (a < 0) return -1;
    const int row = a >> lg, col = a & (stride - 1);
    const int x = col >= (stride >> 1) ? col - (stride >> 1) : col;
    if (x >= (width >> 1) || row >= (height >> 1)) return -1;
    return (row >> 3) * mbw + (x >> 3);
  }
};

// ---- reference fetch of the SLOW path (macroblocks with deeper partition trees; windows that wrap around the end of a plane row):
// a lane needs 5 consecutive bytes (4 pixels + the half-pel neighbour) of a row, and the same of the row below, at an arbitrary LINEAR
// byte offset `o` of a reference plane.  They are fetched as four aligned dwords, each at its own tiled address (mobi_tile.h: a dword
// never leaves a quadrant row, and the linear offset -> tile map handles wrap-around and padding by construction), and cut out with
// v_alignbyte.  All loads of a macroblock are issued before the first one is consumed: no control flow sits between them.
typedef uint2 __attribute__((aligned(4))) uint2_a4;
typedef uint4 __attribute__((aligned(4))) uint4_a4;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct Win { uint2 r0, r1; uint32_t sh; }; // row, row below, byte shift 0..3
// The four dwords of a window (o4, o4 + 4, o4 + S, o4 + S + 4 in linear terms) in tiled terms: the next dword is + 4 inside a
// quadrant row, else the next quadrant column's first (+ 60, or + 188 into the next tile); the row below is + 8 inside a quadrant,
// + 72 into the quadrant below, or the tile below's first row.  Column and row parts add.  That holds while o4 + 4 stays in its
// plane row; `general` (wave-uniform: some lane's window touches the last 8 bytes of a row) maps every dword on its own.
// (o may be as low as -3: the pixels of a lane that lie under its second .. fourth cell start that many bytes into the window, and
// only they are used.  The dword below the plane is never needed: the plane's first one is fetched instead.)
__device__ __forceinline__ Win fetch_win_y(const uint8_t *base, uint32_t slot, int o, int S, int lgS, bool general) {
  Win w;
  const uint32_t o4 = (uint32_t)o & ~3u;
  w.sh = (uint32_t)o & 3;
  if (general) {
    w.r0.x = *(const uint32_t *)(base + (slot + mobi_ty(o < 0 ? 0u : o4, lgS)));
    w.r0.y = *(const uint32_t *)(base + (slot + mobi_ty(o4 + 4, lgS)));
    w.r1.x = *(const uint32_t *)(base + (slot + mobi_ty(o4 + (uint32_t)S, lgS)));
    w.r1.y = *(const uint32_t *)(base + (slot + mobi_ty(o4 + (uint32_t)S + 4, lgS)));
  } else {
    const uint32_t oc = o < 0 ? 0u : o4;
    const uint32_t t = slot + mobi_ty(oc, lgS), row = oc >> lgS;
    const uint32_t dc = (oc & 4u) ? ((oc & 8u) ? 188u : 60u) : 4u;
    const uint32_t dr = (row & 7u) != 7u ? 8u : (row & 8u) ? ((16u << lgS) - 184u) : 72u;
    const uint32_t t1 = o < 0 ? slot + mobi_ty(0u, lgS) : t + dc; // (o < 0: o4 + 4 = 0)
    w.r0.x = *(const uint32_t *)(base + t);
    w.r0.y = *(const uint32_t *)(base + t1);
    w.r1.x = *(const uint32_t *)(base + (t + dr));
    w.r1.y = *(const uint32_t *)(base + (t1 + dr));
  }
  return w;
}
// chroma: the next dword is + 4 inside a tile row's plane half, else the next tile's (+ 124); the row below + 16, or the tile below's first row
__device__ __forceinline__ Win fetch_win_c(const uint8_t *base, uint32_t slot, int o, int S, int lgS, bool general) {
  Win w;
  const uint32_t o4 = (uint32_t)o & ~3u;
  w.sh = (uint32_t)o & 3;
  if (general) {
    w.r0.x = *(const uint32_t *)(base + (slot + mobi_tc(o < 0 ? 0u : o4, lgS)));
    w.r0.y = *(const uint32_t *)(base + (slot + mobi_tc(o4 + 4, lgS)));
    w.r1.x = *(const uint32_t *)(base + (slot + mobi_tc(o4 + (uint32_t)S, lgS)));
    w.r1.y = *(const uint32_t *)(base + (slot + mobi_tc(o4 + (uint32_t)S + 4, lgS)));
  } else {
    const uint32_t oc = o < 0 ? 0u : o4;
    const uint32_t t = slot + mobi_tc(oc, lgS), row = oc >> lgS;
    const uint32_t dc = (oc & 4u) ? 124u : 4u;
    const uint32_t dr = (row & 7u) != 7u ? 16u : (8u << lgS) - 112u;
    const uint32_t t1 = o < 0 ? slot + mobi_tc(0u, lgS) : t + dc;
    w.r0.x = *(const uint32_t *)(base + t);
    w.r0.y = *(const uint32_t *)(base + t1);
    w.r1.x = *(const uint32_t *)(base + (t + dr));
    w.r1.y = *(const uint32_t *)(base + (t1 + dr));
  }
  return w;
}
__device__ __forceinline__ uint32_t cut(uint2 r, uint32_t sh) { return __builtin_amdgcn_alignbyte(r.y, r.x, sh); }
__device__ __forceinline__ uint32_t cut1(uint2 r, uint32_t sh) { return sh == 3 ? r.y : __builtin_amdgcn_alignbyte(r.y, r.x, sh + 1); }
// CopyBlock arithmetic on four packed pixels, phase known per lane: no branches (MD.cs:424-452)
__device__ __forceinline__ uint32_t mc4_select(const Win &w, int phase) {
  const uint32_t M = 0x7F7F7F7Fu;
  const uint32_t a = cut(w.r0, w.sh), b = cut1(w.r0, w.sh), c = cut(w.r1, w.sh), d = cut1(w.r1, w.sh);
  const uint32_t ha = (a >> 1) & M, hb = (b >> 1) & M, hc = (c >> 1) & M, hd = (d >> 1) & M;
  const uint32_t p1 = ha + hb, p2 = ha + hc, p3 = ((p1 >> 1) & M) + (((hc + hd) >> 1) & M);
  return phase == 0 ? a : phase == 1 ? p1 : phase == 2 ? p2 : p3;
}
// ---- residual helpers ----------------------------------------------------------------------------
// pass 1 of area b by lane r (0..7): 8x8 -> coefficient group r; 4x4 -> sub-block r>>1, groups (r&1)*2+{0,1}.
// t may be c itself (in-place transpose): every read of the area is issued before its first write, and the 8
// lanes of an area always take the same branch (LDS executes a wave's instructions in order)
__device__ __forceinline__ void idct_pass1(const int *c, int *t, bool is8, int r) {
  int in[8], out[8];
  if (is8) {
#pragma unroll
    for (int m = 0; m < 8; m++) in[m] = c[8 * r + m];
    if (r == 0) in[0] += 32;
    mobi_bfly8(in, out);
    wave_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) t[8 * m + r] = out[m];
  } else {
    const int s = r >> 1, k0 = (r & 1) * 2;
#pragma unroll
    for (int m = 0; m < 8; m++) in[m] = c[16 * s + 4 * k0 + m]; // groups k0 and k0+1
    if (k0 == 0) in[0] += 32;
    mobi_bfly4(in, out);
    mobi_bfly4(in + 4, out + 4);
    wave_sync();
#pragma unroll
    for (int m = 0; m < 4; m++) { t[16 * s + 4 * m + k0] = out[m]; t[16 * s + 4 * m + k0 + 1] = out[4 + m]; }
  }
}
// pass 2 whose eight residuals stay with the lane, as four int16 pairs (saturated): 8x8 -> row r; 4x4 -> rows i0, i0 + 1 of block r >> 1
__device__ __forceinline__ uint4 idct_pass2_pk(const int *t, bool is8, int r) {
  int in[8], out[8];
  if (is8) {
#pragma unroll
    for (int m = 0; m < 8; m++) in[m] = t[8 * r + m];
    mobi_bfly8(in, out);
  } else {
    const int s = r >> 1, i0 = (r & 1) * 2;
#pragma unroll
    for (int m = 0; m < 8; m++) in[m] = t[16 * s + 4 * i0 + m]; // groups i0 and i0 + 1
    mobi_bfly4(in, out);
    mobi_bfly4(in + 4, out + 4);
  }
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  union { s16x2 v; uint32_t u; } p0, p1, p2, p3;
  p0.v = __builtin_amdgcn_cvt_pk_i16(out[0] >> 6, out[1] >> 6);
  p1.v = __builtin_amdgcn_cvt_pk_i16(out[2] >> 6, out[3] >> 6);
  p2.v = __builtin_amdgcn_cvt_pk_i16(out[4] >> 6, out[5] >> 6);
  p3.v = __builtin_amdgcn_cvt_pk_i16(out[6] >> 6, out[7] >> 6);
  return uint4{p0.u, p1.u, p2.u, p3.u};
}
} // namespace

// q = x / d, r = x % d with magic = floor(2^32 / d): the estimate is at most one short
__device__ __forceinline__ uint32_t fastdiv(uint32_t x, uint32_t d, uint32_t magic, uint32_t &r) {
  uint32_t q = __umulhi(x, magic);
  r = x - q * d;
  if (r >= d) { q++; r -= d; }
  return q;
}

// Stage-by-stage instruction counts (tools/exp_stages.sh): a --profiling build leaves the octet kernel after stage n when the launch
// asks for it (MobiReconArgs.reserved21, from MOBI_STOP_STAGE), decided at run time so that no code is optimised away.  Wrong pictures,
// on purpose; the default build has none of it.
#if defined(MOBI_PROFILING)
#define MOBI_STOP(n) do { if (A.reserved21 == (n)) return; } while (0)
/* the same for mobi_recon_intra (MOBI_INTRA_DBG = n << 8); the tag is still published: whoever waits for this macroblock must not spin */
#define MOBI_ISTOP(n) do { if ((dbg >> 8) == (n)) { if (I.valid && I.publish && l == 0) __hip_atomic_store(A.done + (size_t)clip * A.n_mbs + mb, A.step_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; } } while (0)
#else
#define MOBI_STOP(n) do { } while (0)
#define MOBI_ISTOP(n) do { } while (0)
#endif
namespace {
typedef const void __attribute__((address_space(1))) *gptr_t;
typedef void __attribute__((address_space(3))) *lptr_t;
// lane i lands at dst + IMM + i*16; dst must be wave-uniform (it travels in M0); IMM = constant byte offset added to BOTH
// the source address and the LDS address
#define MOBI_DMA16(src, dst, IMM) __builtin_amdgcn_global_load_lds((gptr_t)(src), (lptr_t)(dst), 16, IMM, 0)
// the same with the source as a wave-uniform base (an SGPR pair) + a 32-bit per-lane offset: no 64-bit address arithmetic on the vector
// unit.  lds = the LDS byte address (wave-uniform) lane 0 lands at.  (One wait state between a write of M0 and an LDS DMA.)
__device__ __forceinline__ void dma16_sv(const uint8_t *base, uint32_t voff, uint32_t lds) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds) : "memory", "m0");
#endif
}

__device__ __forceinline__ uint32_t lds32(const uint8_t *L, int byte_off) { return *(const uint32_t *)(L + byte_off); }
// pass 2 of one area by lane r, tracking the range of pred+residual instead of testing every sample.  Only the
// butterflies differ between one 8x8 transform (lane r = pixel row r) and four 4x4s (lane r = rows (r&1)*2, +1 of
// sub-block r>>1): both leave 8 residuals for two 4-pixel words, so the pixel update is one shared instruction stream
// (the 8 lanes of an area agree on the kind, the lanes of a wave do not).
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t sat_pk_u8(s16x2 v) { // two int16 -> two bytes, each clamped to 0..255 (the clamp table's identity range, MobiConst.cs:587)
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t d;
  asm("v_sat_pk_u8_i16 %0, %1" : "=v"(d) : "v"(v));
  return d;
#else
  const int a = v.x < 0 ? 0 : v.x > 255 ? 255 : v.x, b = v.y < 0 ? 0 : v.y > 255 ? 255 : v.y;
  return (uint32_t)a | ((uint32_t)b << 8);
#endif
}
// Two pixels per instruction: the residual pairs saturate to int16 on the way (v_cvt_pk_i16_i32, then a packed >> 6; the sum with
// a saturating add), which changes nothing the reference can see -- a residual that does not fit 16 bits before the shift is at least
// 512 in magnitude after it, the clamp table's domain is prediction + residual in [-64, 319] (MobiConst.cs:587), and lo / hi see the
// saturated sums.
__device__ __forceinline__ void idct_pass2_q(const int *t, bool is8, int r, uint8_t *wa, uint8_t *wb, s16x2 &lo, s16x2 &hi) {
  int in[8], out[8]; // wa, wb: the two 4-pixel words (4-byte aligned) the lane's eight residuals belong to
  if (is8) {
#pragma unroll
    for (int m = 0; m < 8; m++) in[m] = t[8 * r + m];
    mobi_bfly8(in, out);
  } else {
    const int s = r >> 1, i0 = (r & 1) * 2;
#pragma unroll
    for (int m = 0; m < 8; m++) in[m] = t[16 * s + 4 * i0 + m]; // groups i0 and i0 + 1
    mobi_bfly4(in, out);
    mobi_bfly4(in + 4, out + 4);
  }
  const uint32_t pw[2] = {*(const uint32_t *)wa, *(const uint32_t *)wb};
  uint32_t res[2];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    union { s16x2 v; uint32_t u; } p01, p23;
    p01.u = __builtin_amdgcn_perm(0u, pw[h], 0x0c010c00u); // bytes 0, 1 as two uint16
    p23.u = __builtin_amdgcn_perm(0u, pw[h], 0x0c030c02u);
    const s16x2 r01 = __builtin_amdgcn_cvt_pk_i16(out[4 * h], out[4 * h + 1]) >> (short)6, r23 = __builtin_amdgcn_cvt_pk_i16(out[4 * h + 2], out[4 * h + 3]) >> (short)6;
    const s16x2 s01 = __builtin_elementwise_add_sat(p01.v, r01), s23 = __builtin_elementwise_add_sat(p23.v, r23);
    lo = __builtin_elementwise_min(lo, __builtin_elementwise_min(s01, s23));
    hi = __builtin_elementwise_max(hi, __builtin_elementwise_max(s01, s23));
    res[h] = __builtin_amdgcn_perm(sat_pk_u8(s23), sat_pk_u8(s01), 0x05040100u);
  }
  *(uint32_t *)wa = res[0];
  *(uint32_t *)wb = res[1];
}
} // namespace

// ---- 16-bit packed transforms (r04): one lane carries the same row of TWO coded areas, one in each half of a dword ----
// The octet kernel is bound by vector instruction issue, and 44 % of its instructions were the residual stage: two passes of the 8-point
// butterfly per area row, eight lanes per area, sixteen areas in two half rounds.  v_pk_add_i16 / v_pk_sub_i16 / v_pk_ashrrev_i16 do two of
// them per instruction.  Exactness: the reference computes in int32 (MD.cs:3435-3798); 16-bit wrapping arithmetic gives the same bits as
// long as no intermediate leaves int16, and every intermediate of a pass is a sum of inputs with coefficients of magnitude <= 3/2 (plus
// the truncation of at most a handful of shifts), so after both passes |x| <= 9/4 * (sum of |coefficient| of the area + 32) + 104.  The
// scatter adds up |coefficient| per area (P_SUM); an octet with an area above MOBI_PK_LIMIT takes the 32-bit rounds instead (wave-uniform).
#define MOBI_PK_LIMIT 14000 /* 9/4 * (14000 + 32) + 104 = 31676 < 32768 */
// (r05, measured and not kept: the same bound says when the clamp table CANNOT be left -- sums below 1700 give |x >> 6| <= 64 -- and an octet
// of such areas could run its pixel update without the sixteen packed min / max per lane that track the range; with the second copy of the
// pixel update and the ballot that picks it the launch was no faster: profiles/r05_experiments.txt.)
namespace {
__device__ __forceinline__ void bfly8_pk(const s16x2 in[8], s16x2 out[8]) { // mobi_bfly8, two at a time
  const s16x2 a0 = in[0] + in[4], a1 = in[0] - in[4];
  const s16x2 a2 = in[2] + (in[6] >> (short)1), a3 = (in[2] >> (short)1) - in[6];
  const s16x2 e0 = a0 + a2, e1 = a1 + a3, e2 = a1 - a3, e3 = a0 - a2;
  const s16x2 b0 = in[1] + in[7] - in[3] - (in[3] >> (short)1);
  const s16x2 b1 = in[7] - in[1] + in[5] + (in[5] >> (short)1);
  const s16x2 b2 = in[5] - (in[7] + (in[7] >> (short)1)) - in[3];
  const s16x2 b3 = in[3] + in[5] + in[1] + (in[1] >> (short)1);
  const s16x2 o0 = b2 + (b3 >> (short)2), o3 = b3 - (b2 >> (short)2);
  const s16x2 o1 = b0 + (b1 >> (short)2), o2 = (b0 >> (short)2) - b1;
  out[0] = e0 + o3; out[7] = e0 - o3;
  out[1] = e1 + o2; out[6] = e1 - o2;
  out[2] = e2 + o1; out[5] = e2 - o1;
  out[3] = e3 + o0; out[4] = e3 - o0;
}
__device__ __forceinline__ void bfly4_pk(const s16x2 in[4], s16x2 out[4]) { // mobi_bfly4, two at a time
  const s16x2 a = in[0] + in[2], b = in[0] - in[2];
  const s16x2 c = (in[1] >> (short)1) - in[3], d = in[1] + (in[3] >> (short)1);
  out[0] = a + d; out[3] = a - d; out[1] = b + c; out[2] = b - c;
}
union PkRow { uint4 q[2]; s16x2 v[8]; };
// A pair tile = 64 dwords, dword i = coefficient i of area A (low half) and of area B (high half).  Lane r reads dwords 8r .. 8r + 7 in both
// passes and for both kinds: an 8x8 transform's group r, or groups k0, k0 + 1 of 4x4 block r >> 1 (16 * (r >> 1) + 4 * k0 = 8r).
__device__ __forceinline__ void idct_pass1_pk(uint32_t *c, bool is8, int r) {
  PkRow in;
  s16x2 out[8];
  in.q[0] = *(const uint4 *)(c + 8 * r);
  in.q[1] = *(const uint4 *)(c + 8 * r + 4);
  const s16x2 rnd = {32, 32};
  if (is8) {
    if (r == 0) in.v[0] += rnd;
    bfly8_pk(in.v, out);
    wave_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) *(s16x2 *)(c + 8 * m + r) = out[m];
  } else {
    const int s = r >> 1, k0 = (r & 1) * 2;
    if (k0 == 0) in.v[0] += rnd;
    bfly4_pk(in.v, out);
    bfly4_pk(in.v + 4, out + 4);
    wave_sync();
#pragma unroll
    for (int m = 0; m < 4; m++) *(uint2 *)(c + 16 * s + 4 * m + k0) = uint2{__builtin_bit_cast(uint32_t, out[m]), __builtin_bit_cast(uint32_t, out[4 + m])};
  }
}
// pass 2 and the pixel update of both areas: out[k] = residual k of the lane's row in area A (low half) and area B (high half); words wa
// hold samples 0..3, wb samples 4..7 (8x8: one row; 4x4: rows i0, i0 + 1 of block r >> 1).  The range of prediction + residual is tracked
// per half (the clamp table's domain, MobiConst.cs:587) instead of testing every sample.
__device__ __forceinline__ void idct_pass2_pk(const uint32_t *t, bool is8, int r, uint8_t *waA, uint8_t *wbA, uint8_t *waB, uint8_t *wbB, bool actB,
                                              s16x2 &lo, s16x2 &hi) {
  PkRow in;
  s16x2 out[8];
  in.q[0] = *(const uint4 *)(t + 8 * r);
  in.q[1] = *(const uint4 *)(t + 8 * r + 4);
  if (is8) {
    bfly8_pk(in.v, out);
  } else {
    bfly4_pk(in.v, out);
    bfly4_pk(in.v + 4, out + 4);
  }
  const uint32_t pA[2] = {*(const uint32_t *)waA, *(const uint32_t *)wbA}, pB[2] = {*(const uint32_t *)waB, *(const uint32_t *)wbB};
  uint32_t resA[2], resB[2];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    uint32_t sat[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      // sample k of both words as two uint16: byte k of A's word below, byte k of B's word above
      const s16x2 pp = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(pB[h], pA[h], 0x0c040c00u + (uint32_t)k * 0x00010001u));
      const s16x2 sum = __builtin_elementwise_add_sat(pp, out[4 * h + k] >> (short)6);
      lo = __builtin_elementwise_min(lo, sum);
      hi = __builtin_elementwise_max(hi, sum);
      sat[k] = sat_pk_u8(sum); // byte 0: A's sample, byte 1: B's
    }
    const uint32_t m01 = __builtin_amdgcn_perm(sat[1], sat[0], 0x05010400u), m23 = __builtin_amdgcn_perm(sat[3], sat[2], 0x05010400u); // A k, A k+1, B k, B k+1
    resA[h] = __builtin_amdgcn_perm(m23, m01, 0x05040100u);
    resB[h] = __builtin_amdgcn_perm(m23, m01, 0x07060302u);
  }
  *(uint32_t *)waA = resA[0];
  *(uint32_t *)wbA = resA[1];
  if (actB) {
    *(uint32_t *)waB = resB[0];
    *(uint32_t *)wbB = resB[1];
  }
}
} // namespace

// =====================================================================================================
// mobi_recon_inter8: one wavefront per octet (Kernels; how it got here: HISTORY.md)
// =====================================================================================================
//   A. every lane decodes its macroblock's descriptor; ONE fetch path for every macroblock with whole leaves (16x16, two 16x8, two
//      8x16): DMA rounds with a per-lane source address (global_load_lds_dwordx4), so the rows of a top/bottom pair and the halves of a
//      left/right pair simply come from the other leaf's position; level words travel in registers.
//   B. a lane owns 8 consecutive luma rows x 4 pixels (4 chroma rows x 4): it lies inside one leaf whatever the split.  CopyBlock
//      (MD.cs:424-452) by v_perm_b32 (byte window out of two dwords) and v_lerp_u8: (a>>1)+(b>>1) per byte is the byte average of
//      a & 0xFE and b & 0xFE; phase 0 uses the same instructions with mask 0xFF and b = a.
//      Deeper partition trees: the whole wave for one macroblock, a lane = one 2x2 cell of the MV cell map (r04).
//   C. residual on packed int16, the same row of TWO coded areas per lane, sixteen areas per round (r04); 32-bit rounds behind a guard.
//   D. the octet's tiles are contiguous: whole-line stores.
// The integer VALU is not one machine (tools/ubench/oprate.hip: two-operand adds / shifts / logic ops issue in ~2.7 cycles per wave,
// everything with three operands, byte selects, compares, SDWA, DPP and 16-bit packed forms in ~4.3), and the kernel is bound by its
// vector instruction count: tools/exp_stages.sh counts it stage by stage (MOBI_STOP).
namespace {
// LDS of one octet (10 KB: sixteen waves per CU).  While the windows are in flight / being interpolated:
//   P_L   luma windows: chunk (row pair p = 0..9, quadrant column s = 0..3) of macroblock g at p * 512 + s * 128 + g * 16; a chunk =
//         two rows x 8 samples of one quadrant (mobi_tile.h).  A whole leaf's window is 9 pairs x 3 columns (17 rows x 17..24 samples
//         from an even row and a column that is a multiple of 8); a TOP/BOTTOM pair keeps leaf A in pairs 0..4 and leaf B in 5..9,
//         a LEFT/RIGHT pair leaf A in columns 0, 1 and leaf B in 2, 3 (9 samples never span more than two quadrant columns).
//   P_C   chroma windows: chunk (row r = 0..9, column s = 0..1) at (r * 2 + s) * 128 + g * 16; a chunk = one row of [U 8 B | V 8 B]:
//         both planes of a leaf share one window.  TOP/BOTTOM: leaf A rows 0..4, leaf B rows 5..9.
//   P_C1  the same for the right half of a LEFT/RIGHT pair.
// After motion compensation (the windows are dead):
//   P_OUT  the octet's output samples, luma rows R = 0..15 and chroma rows as R = 16..23 (column = plane * 8 + sample) in ONE formula:
//          (row R, column c) of macroblock g at (R & 7) * 384 + (R >> 3) * 128 + ((g ^ (R & 7)) * 16) + c.  A row's eight macroblocks are
//          rotated by the row number, so that the 64 lanes of a motion-compensation store, the eight lanes that add one area's residual
//          (one row each) and the 16-byte reads of the final copy all spread over the banks; and a coded area's place is one constant
//          K = (area < 4 ? area >> 1 : 2) * 128 + (area & 1) * 8 whatever the plane (r03 kept chroma in a layout of its own: the residual
//          add spent 30 instructions per half round choosing between the two).
enum {
  P_L = 0,
  P_C = 5120,
  P_C1 = 7680,
  P_BYTES = 10240,
  P_OUT = 0,      // 3072 B
  P_COEF = 3072,  // coefficient tiles of P_TILE words: 8 tiles of int16 PAIRS (two areas each) per packed round, or P_ROUND tiles of int32
  P_TILE = 72,
#ifndef MOBI_PK_PAIRS
#define MOBI_PK_PAIRS 12
#endif
  P_PAIRS = MOBI_PK_PAIRS, // packed round: the level words are scattered ONCE for up to 24 coded areas (an octet has 14 on average in the generator's
                  // mix, more than 16 in one out of five: r04 scattered twice for those); the transforms take the pair tiles eight at a time
  P_ROUND = 22,   // 32-bit rounds (the fall-back when some area's coefficients are too large for 16-bit butterflies): three half rounds of eight
  P_SUM = P_COEF + P_PAIRS * P_TILE * 4, // sum of |coefficient| per coded area (48 words), behind the packed tiles; dead before a 32-bit round
  P_SC = 9728,    // dequant scales (320 B): on top of the chroma windows, once the chroma has been interpolated
  P_TAB = 10048,  // slot -> uint32: [6:4] g, [15:8] area * 8 + g, [31:16] K (above): 48 words
  P_INV = 9408    // area*8 + g -> uint32: [12:0] where the slot's coefficients start in the packed tiles (byte offset from P_COEF: pair tile
                  // slot >> 1, half slot & 1), [19:13] slot, [31:23] what selects the dequant scale: 0x0FC (one 8x8 transform: byte offset =
                  // 4 * position) or 0x13C (4x4 blocks: 256 + 4 * (position & 15)), see the scatter.  48 words behind the last coefficient tile.
};
static_assert(P_COEF + P_ROUND * P_TILE * 4 <= P_INV && P_INV + 192 <= P_SC && P_TAB + 192 <= P_BYTES && P_SUM + 192 <= P_INV, "inter LDS map");
__device__ __forceinline__ int out_px(int g, int R, int c) { return P_OUT + (R & 7) * 384 + (R >> 3) * 128 + ((g ^ (R & 7)) << 4) + c; }
__device__ __forceinline__ int out_y(int g, int R, int c) { return out_px(g, R, c); }
__device__ __forceinline__ int out_c(int g, int R, int pl, int x) { return out_px(g, 16 + R, pl * 8 + x); }
// N output rows of 4 pixels from N + 1 window rows (x0[i], x1[i] = the two aligned dwords holding row i's 5 bytes)
template <int N>
__device__ __forceinline__ void mc_rows(const uint32_t (&x0)[N + 1], const uint32_t (&x1)[N + 1], uint32_t sh, int phase, uint32_t *out) {
  // CopyBlock (MD.cs:424-452) as two stages that every phase runs: h = (a >> 1) + (b >> 1) per byte with b = a and no masking when the
  // phase has no horizontal half-pel ((a + a) >> 1 = a), then the same between a row and the row below it (or itself).
  // (x >> 1) + (y >> 1) per byte = v_lerp_u8(x & 0xFE, y & 0xFE): the byte average of the two, which cannot carry once the low bits are gone.
  const bool hor = (phase & 1) != 0, ver = (phase & 2) != 0;
  const uint32_t selA = 0x03020100u + sh * 0x01010101u, selB = hor ? selA + 0x01010101u : selA;
  const uint32_t Mh = hor ? 0xFEFEFEFEu : 0xFFFFFFFFu, Mv = ver ? 0xFEFEFEFEu : 0xFFFFFFFFu;
  uint32_t m[N + 1];
#pragma unroll
  for (int i = 0; i <= N; i++) {
    const uint32_t a = __builtin_amdgcn_perm(x1[i], x0[i], selA) & Mh, b = __builtin_amdgcn_perm(x1[i], x0[i], selB) & Mh;
    m[i] = __builtin_amdgcn_lerp(a, b, 0u) & Mv;
  }
#pragma unroll
  for (int i = 0; i < N; i++) out[i] = __builtin_amdgcn_lerp(m[i], ver ? m[i + 1] : m[i], 0u);
}
} // namespace

__device__ __forceinline__ unsigned long long prof_stamp() { // shader clock, pinned: nothing is scheduled across it
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");
  const unsigned long long t = __builtin_readcyclecounter();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
template <int PROF, int CWR, int FUSED>
__device__ __forceinline__ void recon_inter_oct(const MobiReconArgs &A, uint8_t *L, uint32_t clip, uint32_t mby, uint32_t ox, int lane) {
  unsigned long long pt[7] = {0, 0, 0, 0, 0, 0, 0};
  if (PROF) pt[0] = prof_stamp();
  const uint32_t oi = clip * A.qpc + mby * A.qpr + ox; // (index of the profiling record)
  const uint32_t mbx0 = ox * 8, mbw = (uint32_t)A.mbw;
  const int nmb = (int)(mbw - mbx0 < 8 ? mbw - mbx0 : 8);
  const int S = A.stride, lgS = 31 - __builtin_clz((unsigned)S);
  const uint32_t ysz = (uint32_t)S * (uint32_t)A.height;
  uint8_t *clip_base = A.planes + (size_t)clip * A.clip_bytes;
  const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)L; // the wave's LDS as an LDS address (for M0)
  const int off0 = (int)(mby * 16 * (uint32_t)S + mbx0 * 16); // the octet's first sample as a linear offset (the slow path's currency)
  const int g = lane & 7, j = lane >> 3; // adjacent lanes = adjacent macroblocks: chunk j of the 8 macroblocks lies side by side in LDS
  unsigned long long pa = 0, pb = 0;
  if (PROF) { asm volatile("" : : "s"(off0), "s"(clip)); pa = prof_stamp(); }

  // A wave asks for its descriptors, windows and level words ahead of its neighbours' arithmetic: the sooner its requests are out, the
  // shorter it holds its place (A/B on one box, tools/exp_prio.sh: 7.31 -> 7.20 ms per launch; the stores at the end as well: -0.3 %)
  __builtin_amdgcn_s_setprio(3);
  // ---- stage A: descriptor, then every global read of the octet ----
  const uint4 *dp = (const uint4 *)(A.desc + (clip * (uint32_t)A.n_mbs + mby * mbw + mbx0) + g); // the table has slack past the last octet
  const uint4 d = dp[0], d2 = dp[1];
  if (PROF) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pb = prof_stamp(); }
  const bool valid = g < nmb && (d.y & 1) == MOBI_MB_INTER;
  const int nl = (d.y >> 1) & 0x7F, kind2 = (d.y >> 26) & 3;
  const bool leaves = valid && (nl == 1 || kind2 != 0);       // whole leaves (16x16, two 16x8, two 8x16)
  const bool multi = valid && nl > 1 && kind2 == 0;           // deeper tree: MV cell map in the payload
  const bool tb = leaves && kind2 == MOBI_DUAL_TB, lr = leaves && kind2 == MOBI_DUAL_LR;
  const uint32_t cbp6 = valid ? (d.y >> 8) & 0x3F : 0, ncoef = cbp6 ? d.z & 0x3FF : 0;
  // (wave masks as ballots of ONE comparison each, combined on the scalar unit: a ballot of a compound condition goes through a 0 / 1
  // register and a second comparison)
  const unsigned long long v64 = __builtin_amdgcn_ballot_w64(g < nmb) & __builtin_amdgcn_ballot_w64((d.y & 1) == MOBI_MB_INTER); // = ballot(valid)
  const unsigned long long six = 0x0000FFFFFFFFFFFFull; // lanes with j < 6
  const unsigned long long mb64 = __builtin_amdgcn_ballot_w64(((d.y >> (8 + j)) & 1) != 0) & v64 & six;   // bit area*8 + g: the area is coded
  const unsigned long long tb64 = __builtin_amdgcn_ballot_w64(((d.y >> (14 + j)) & 1) != 0) & six;        // ... with one 8x8 transform
  const uint32_t m_lo = (uint32_t)mb64, m_hi = (uint32_t)(mb64 >> 32), t_lo = (uint32_t)tb64, t_hi = (uint32_t)(tb64 >> 32);
  const uint32_t inter_mask = (uint32_t)v64 & 0xFFu;
  const uint32_t multi_mask = (uint32_t)(v64 & __builtin_amdgcn_ballot_w64(nl > 1) & __builtin_amdgcn_ballot_w64(kind2 == 0)) & 0xFFu;
  if (inter_mask == 0) return; // nothing but intra macroblocks here
  auto slot_off = [&](uint32_t ref) {
    int sl = A.ring_base - (int)ref;
    sl = sl < 0 ? sl + 6 : sl;
    return __umul24((uint32_t)sl, A.slot_bytes); // slot_bytes < 2^24: checked by mobi_launch_inter
  };
  const uint32_t refA = slot_off((d.z >> 10) & 7), refB = slot_off((d.z >> 13) & 7);
  const int posA = (int)d.w, cposA = (int)d2.x, posB = (int)d2.y, cposB = (int)d2.z;
  const int phA = (d.z >> 16) & 3, cphA = (d.z >> 18) & 3, phB = (d.z >> 20) & 3, cphB = (d.z >> 22) & 3;
  // A leaf's window starts at its first sample's position: leaf B's is row 8 (TOP/BOTTOM) or column 8 (LEFT/RIGHT) of the macroblock
  const int topB = tb ? posB + (8 << lgS) : posB + 8, ctopB = tb ? cposB + (4 << lgS) : cposB + 4;
  // What is fetched is a fixed shape, not the window's exact needs (the kernel is bound by instruction issue since the planes are
  // tiled, and the exact shape costs more instructions than the few chunks it saves cost requests): 9 row pairs (TOP/BOTTOM: 5 + 5)
  // from an even row, 3 quadrant columns (LEFT/RIGHT: 2 + 2) from a multiple of 8; chroma 9 rows (5 + 5) x 2 columns.
  // A window that would run over the end of its plane row continues in the next row (or, chroma, in the other plane's half): the
  // reference's linear offsets mean exactly that (Stride == Width streams; vectors far outside the picture), and only the slow
  // path's per-dword addressing follows it.  Whole rows of chunks would not.
  // (The test is on the bytes the window NEEDS, not on the fixed shape that is fetched: a quadrant column beyond them may come from
  // the next row's first bytes -- in bounds, never looked at.  Width == Stride pictures have every right-most macroblock there.)
  const int wpx = lr ? 8 : 16, cwpx = lr ? 4 : 8;
  // (as one number per leaf -- by how many samples its windows run over, luma or chroma -- and one comparison: chains of || and && on
  // lane conditions compile to 0 / 1 registers and comparisons of those)
  const int overA = max((posA & (S - 1)) + wpx + (phA & 1) - S, (cposA & (S - 1)) + cwpx + (cphA & 1) - (S >> 1));
  const int overB = max((topB & (S - 1)) + wpx + (phB & 1) - S, (ctopB & (S - 1)) + cwpx + (cphB & 1) - (S >> 1));
  const bool wrap = leaves && max(overA, kind2 != 0 ? overB : 0) > 0;
  const bool win = leaves && !wrap;                           // fetched through the LDS windows
  const bool slow = multi || wrap;
  const uint32_t slow_mask = (uint32_t)__builtin_amdgcn_ballot_w64(slow) & 0xFFu;
  const bool any_lr = __builtin_amdgcn_ballot_w64(lr && win) != 0;
  {
    // luma rounds: lane (g, j) brings chunk (pair 2t + (j >> 2), column j & 3) of its macroblock.  The column is the lane's for all
    // rounds, so the leaf it serves changes only in a TOP/BOTTOM pair (pairs 5..9 are leaf B's).  No lane is masked off: one with
    // nothing to bring (column 3 of a whole leaf; the tenth pair; macroblocks that are not fetched this way) repeats a chunk that
    // another lane or itself brings anyway -- same line, no request -- or the clip's first bytes.
    const int s = j & 3, ph = j >> 2;
    const bool colB = lr && s >= 2;                           // LEFT/RIGHT: columns 2, 3 belong to leaf B
    const int sw = lr ? s & 1 : (s < 3 ? s : 2);
    auto chunk0 = [&](int top, uint32_t ref, uint32_t &u) {   // byte offset (inside the clip) of the column's chunk in tile row 0; u = half the window's first row
      const int yo = (top >> lgS) & 1;
      const uint32_t lin = (uint32_t)(((top - (yo << lgS)) & ~7) + 8 * sw);
      u = lin >> (lgS + 1);
      return ref + mobi_ty_col(lin & (uint32_t)(S - 1));
    };
    uint32_t uP, uB;
    const uint32_t bP = chunk0(colB ? topB : posA, colB ? refB : refA, uP), bB = chunk0(topB, refB, uB);
    // pair p of a window that starts at row 2u: row 2(u + p): tile row (u + p) >> 3, quadrant row ((u + p) >> 2) & 1, rows 2((u + p) & 3)
    auto rowpart = [&](uint32_t v) { return ((v & ~7u) << (lgS + 1)) + (((v & 7u) + (v & 4u)) << 4); };
    const uint32_t vP = uP + (uint32_t)ph, vB = uB + (uint32_t)ph - 5u; // round t: pair 2t + ph of P, pair 2t + ph - 5 of B
    // rounds 0, 1: pairs 0..3 (P); round 2: pairs 4, 5 (TOP/BOTTOM: 5 is B's first); rounds 3, 4: pairs 6..9 (TOP/BOTTOM: B's)
    const bool b2 = tb && ph, b34 = tb;
    const uint32_t base2 = b2 ? bB : bP, v2 = b2 ? vB : vP, base34 = b34 ? bB : bP, v34 = b34 ? vB : vP;
    const uint32_t v4 = (!tb && ph) ? v34 - 1u : v34;         // a whole leaf has no tenth pair: the ninth again
    uint32_t o[5] = {bP + rowpart(vP), bP + rowpart(vP + 2), base2 + rowpart(v2 + 4), base34 + rowpart(v34 + 6), base34 + rowpart(v4 + 8)};
    // (only the lanes of macroblocks fetched this way ask: one execution mask for all eight rounds; the others' places in LDS keep
    // whatever they held -- their motion compensation below runs on it and is overwritten or never stored)
    if (win) {
#pragma unroll
      for (int t = 0; t < 5; t++) dma16_sv(clip_base, o[t], lds0 + P_L + t * 1024);
    }
    // chroma rounds: lane (g, j) brings chunk (row 4t + (j >> 1), column j & 1): both planes of that row
    const int cs = j & 1, ch2 = j >> 1;
    auto cchunk0 = [&](int ctop, uint32_t ref, uint32_t &r) {
      const uint32_t lin = (uint32_t)((ctop & ~7) + 8 * cs);
      r = lin >> lgS;
      return ref + ysz + mobi_tc_x(lin & (uint32_t)(S - 1));
    };
    auto crowpart = [&](uint32_t r) { return ((r & ~7u) << lgS) + ((r & 7u) << 4); };
    uint32_t rA, rB;
    const uint32_t cA = cchunk0(cposA, refA, rA), cB = cchunk0(ctopB, refB, rB);
    const uint32_t qA = rA + (uint32_t)ch2, qB = rB + (uint32_t)ch2 - 5u; // round t: row 4t + ch2 of A, row 4t + ch2 - 5 of B
    // round 0: rows 0..3 (A); round 1: rows 4..7 (TOP/BOTTOM: 5.. are B's); round 2: rows 8..11 (TOP/BOTTOM: B's 3, 4; there is no row
    // past 8 resp. 9: those lanes bring the last one again)
    const bool cb1 = tb && ch2 >= 1;
    const uint32_t cbase1 = cb1 ? cB : cA, q1 = cb1 ? qB : qA, cbase2 = tb ? cB : cA;
    const uint32_t q2 = tb ? qB - (ch2 >= 2 ? (uint32_t)ch2 - 1u : 0u) : qA - (uint32_t)ch2;
    const uint32_t co[3] = {cA + crowpart(qA), cbase1 + crowpart(q1 + 4), cbase2 + crowpart(q2 + 8)};
    if (win) {
#pragma unroll
      for (int t = 0; t < 3; t++) dma16_sv(clip_base, co[t], lds0 + P_C + t * 1024);
    }
    if (any_lr) { // the right halves of LEFT/RIGHT pairs: rows 0..8 of leaf B's own window
      const uint32_t q3 = rB + (uint32_t)ch2;
      if (lr && win) {
        dma16_sv(clip_base, cB + crowpart(q3), lds0 + P_C1);
        dma16_sv(clip_base, cB + crowpart(q3 + 4), lds0 + P_C1 + 1024);
        dma16_sv(clip_base, cB + crowpart(q3 + 8 - (uint32_t)ch2), lds0 + P_C1 + 2048);
      }
    }
  }
  MOBI_STOP(1);
  const int quant = __builtin_amdgcn_readfirstlane((int)((d.y >> 20) & 63));
  const uint32_t *pay = A.payload + (size_t)clip * A.pay_clip_words; // (wave-uniform)
  const uint32_t *cw = pay + d.x + (multi ? MOBI_MV_CELLS : 0);
  uint32_t cwr[CWR]; // lane (g, j) scatters words j, j+8, j+16, ... of macroblock g; the first 8*CWR of them travel in registers
#pragma unroll
  for (int k = 0; k < CWR; k++) asm volatile("" : "=v"(cwr[k])); // (whatever is there: a word is only looked at when it was loaded)
  const uint32_t *cwj = cw + j;       // word 8k + j at a constant offset from here
  const int left = (int)ncoef - j;    // ... exists while 8k < left
  if (left > 0) cwr[0] = cwj[0];
  if (__builtin_amdgcn_ballot_w64(ncoef > 8) != 0) {
#pragma unroll
    for (int k = 1; k < 4; k++)
      if (left > 8 * k) cwr[k] = cwj[8 * k];
    if (CWR > 4 && __builtin_amdgcn_ballot_w64(ncoef > 32) != 0) {
#pragma unroll
      for (int k = 4; k < CWR; k++)
        if (left > 8 * k) cwr[k] = cwj[8 * k];
    }
  }
  // slow path (deeper trees; wrapping windows): the whole wave works for one such macroblock later.  A deeper tree is fetched cell by cell
  // (r04): lane = one cell of the 8 x 8 MV cell map = 2 x 2 luma samples and one sample of each chroma plane, one motion vector, one window
  // of 3 x 3 luma and 2 x 2 chroma bytes -- whatever the tree looks like (r03's lanes owned 4 x 1 samples: up to two luma and four chroma
  // windows each when the leaves were narrower than that, 300 instructions per such macroblock against 160 now).  A whole leaf whose
  // window wraps keeps the 4 x 1 lanes: lane = (row lane >> 2, pixels 4 * (lane & 3)) for luma, lanes 0..31 = (plane, row, 4 samples)
  // for chroma, vectors from the leaf records.  The cells of the first two deep trees travel with everything else (6 % of the octets
  // have two): one of their two round trips.
  const int yrow = lane >> 2, yc4 = (lane & 3) * 4;
  const int cv = (lane >> 4) & 1, crow = (lane & 15) >> 1, cc4 = (lane & 1) * 4;
  uint32_t cell0 = 0, cell1 = 0;
  auto load_cell = [&](int gm, uint32_t &c) {
    if ((multi_mask >> gm) & 1) c = (pay + __builtin_amdgcn_readlane(d.x, gm))[lane];
  };
  if (slow_mask) {
    load_cell(__builtin_ctz(slow_mask), cell0);
    const uint32_t m2 = slow_mask & (slow_mask - 1);
    if (m2) load_cell(__builtin_ctz(m2), cell1);
  }
  if (PROF) pt[1] = prof_stamp();
  MOBI_STOP(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wave_sync();
  if (PROF) pt[2] = prof_stamp();

  // slow path: issue the first one's pixel fetches now, consume them after the others' motion compensation
  struct Deep { // both kinds: `cells` says which
    bool cells;
    // a cell: rows r, r + 1, r + 2 of luma (two aligned dwords each), rows r, r + 1 of U and of V; byte shifts and CopyBlock phases
    uint2 yr[3], ur[2], vr[2];
    uint32_t ysh, csh;
    int yph, cph;
    // a wrapping leaf: one luma, one chroma window per lane
    Win wa, wq;
    int pha, phq;
  };
  auto deep_fetch = [&](Deep &D, int gm, uint32_t cell) {
    const int offm = off0 + gm * 16;
    D.cells = (multi_mask >> gm) & 1;
    if (D.cells) {
      const int cy = lane >> 3, cx = lane & 7;
      const int dx = mobi_cell_dx(cell), dy = mobi_cell_dy(cell), qx = dx >> 1, qy = dy >> 1;
      const uint32_t sl = slot_off((uint32_t)mobi_cell_ref(cell));
      const int lo = offm + ((2 * cy + (dy >> 1)) << lgS) + 2 * cx + (dx >> 1);                // the cell's first source sample (MD.cs:400-416)
      const int co = (offm >> 1) + ((cy + (qy >> 1)) << lgS) + cx + (qx >> 1);                 // ... in U; V = + Stride / 2 (+ 8 in a tile row)
      D.yph = (dx & 1) | ((dy & 1) << 1);
      D.cph = (qx & 1) | ((qy & 1) << 1);
      D.ysh = (uint32_t)lo & 3;
      D.csh = (uint32_t)co & 3;
      const uint32_t l4 = (uint32_t)lo & ~3u, c4 = (uint32_t)co & ~3u;
      // some lane's window within 8 bytes of the end of a plane row, or its U window not inside the U half (a vector far to the left
      // wraps into the previous row's V half, and "V = U + 8 in the tile row" only holds in the U half): every dword is mapped on its own
      const bool gen = (l4 & (uint32_t)(S - 1)) >= (uint32_t)(S - 8) || (c4 & (uint32_t)(S - 1)) >= (uint32_t)((S >> 1) - 8);
      if (__builtin_amdgcn_ballot_w64(gen) != 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
          D.yr[k].x = *(const uint32_t *)(clip_base + (sl + mobi_ty(l4 + (uint32_t)(k << lgS), lgS)));
          D.yr[k].y = *(const uint32_t *)(clip_base + (sl + mobi_ty(l4 + (uint32_t)(k << lgS) + 4, lgS)));
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const uint32_t a0 = sl + ysz + mobi_tc(c4 + (uint32_t)(k << lgS), lgS), a1 = sl + ysz + mobi_tc(c4 + (uint32_t)(k << lgS) + 4, lgS);
          const uint32_t b0 = sl + ysz + mobi_tc(c4 + (uint32_t)(k << lgS) + (uint32_t)(S >> 1), lgS), b1 = sl + ysz + mobi_tc(c4 + (uint32_t)(k << lgS) + (uint32_t)(S >> 1) + 4, lgS);
          D.ur[k] = uint2{*(const uint32_t *)(clip_base + a0), *(const uint32_t *)(clip_base + a1)};
          D.vr[k] = uint2{*(const uint32_t *)(clip_base + b0), *(const uint32_t *)(clip_base + b1)};
        }
      } else {
        // luma: the next dword is + 4 inside a quadrant row, else the next quadrant column's first; the row below + 8 inside a quadrant,
        // + 72 into the quadrant below, or the tile below's first row (as fetch_win_y)
        const uint32_t t = sl + mobi_ty(l4, lgS), row = l4 >> lgS;
        const uint32_t dc = (l4 & 4u) ? ((l4 & 8u) ? 188u : 60u) : 4u;
        auto down = [&](uint32_t r) { return (r & 7u) != 7u ? 8u : (r & 8u) ? ((16u << lgS) - 184u) : 72u; };
        const uint32_t t1 = t + down(row), t2 = t1 + down(row + 1);
        D.yr[0] = uint2{*(const uint32_t *)(clip_base + t), *(const uint32_t *)(clip_base + (t + dc))};
        D.yr[1] = uint2{*(const uint32_t *)(clip_base + t1), *(const uint32_t *)(clip_base + (t1 + dc))};
        D.yr[2] = uint2{*(const uint32_t *)(clip_base + t2), *(const uint32_t *)(clip_base + (t2 + dc))};
        // chroma: U and V of a sample are 8 bytes apart in the tile row; next dword + 4, or the next tile's (+ 124); row below + 16 or the tile below
        const uint32_t u = sl + ysz + mobi_tc(c4, lgS), crow0 = c4 >> lgS;
        const uint32_t cdc = (c4 & 4u) ? 124u : 4u, cdr = (crow0 & 7u) != 7u ? 16u : (8u << lgS) - 112u;
        D.ur[0] = uint2{*(const uint32_t *)(clip_base + u), *(const uint32_t *)(clip_base + (u + cdc))};
        D.vr[0] = uint2{*(const uint32_t *)(clip_base + (u + 8)), *(const uint32_t *)(clip_base + (u + cdc + 8))};
        D.ur[1] = uint2{*(const uint32_t *)(clip_base + (u + cdr)), *(const uint32_t *)(clip_base + (u + cdr + cdc))};
        D.vr[1] = uint2{*(const uint32_t *)(clip_base + (u + cdr + 8)), *(const uint32_t *)(clip_base + (u + cdr + cdc + 8))};
      }
    } else { // whole leaves whose windows wrap: the leaf records of lane gm
      const uint32_t w1m = __builtin_amdgcn_readlane(d.y, gm), w2m = __builtin_amdgcn_readlane(d.z, gm);
      const int pAm = (int)__builtin_amdgcn_readlane(d.w, gm), cAm = (int)__builtin_amdgcn_readlane(d2.x, gm);
      const int pBm = (int)__builtin_amdgcn_readlane(d2.y, gm), cBm = (int)__builtin_amdgcn_readlane(d2.z, gm);
      const int k2 = (w1m >> 26) & 3;
      const bool yBm = k2 == MOBI_DUAL_TB ? yrow >= 8 : k2 == MOBI_DUAL_LR ? yc4 >= 8 : false;
      const bool cBl2 = k2 == MOBI_DUAL_TB ? crow >= 4 : k2 == MOBI_DUAL_LR ? cc4 >= 4 : false;
      const int la = (yBm ? pBm : pAm) + (yrow << lgS) + yc4;
      D.pha = (w2m >> (yBm ? 20 : 16)) & 3;
      const uint32_t sa = slot_off((w2m >> (yBm ? 13 : 10)) & 7);
      const int lq = (cBl2 ? cBm : cAm) + cv * (S >> 1) + (crow << lgS) + cc4;
      D.phq = (w2m >> (cBl2 ? 22 : 18)) & 3;
      const uint32_t sq = slot_off((w2m >> (cBl2 ? 13 : 10)) & 7);
      // some lane's window within 8 bytes of the end of a plane row (chroma: of a plane's half): every dword is mapped on its own
      const bool gen = (((uint32_t)la & (uint32_t)(S - 1)) >= (uint32_t)(S - 8)) || (((uint32_t)lq & (uint32_t)((S >> 1) - 1)) >= (uint32_t)((S >> 1) - 8));
      const bool general = __builtin_amdgcn_ballot_w64(gen) != 0;
      D.wa = fetch_win_y(clip_base, sa, la, S, lgS, general);
      D.wq = fetch_win_c(clip_base, sq + ysz, lq, S, lgS, general);
    }
  };
  auto deep_finish = [&](int gm, const Deep &D) {
    if (D.cells) {
      const int cy = lane >> 3, cx = lane & 7;
      const uint32_t M = 0x7F7F7F7Fu;
      // luma: bytes x, x + 1 (a) and x + 1, x + 2 (b) of three rows; CopyBlock (MD.cs:424-452) for rows 0, 1 of the cell: two samples each
      uint32_t a[3], h[3];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        a[k] = cut(D.yr[k], D.ysh);
        const uint32_t b = cut1(D.yr[k], D.ysh);
        h[k] = ((a[k] >> 1) & M) + ((b >> 1) & M);
      }
      uint32_t yo[2];
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const uint32_t v = ((a[k] >> 1) & M) + ((a[k + 1] >> 1) & M), p3 = ((h[k] >> 1) & M) + ((h[k + 1] >> 1) & M);
        yo[k] = D.yph == 0 ? a[k] : D.yph == 1 ? h[k] : D.yph == 2 ? v : p3;
      }
      *(uint16_t *)(L + out_y(gm, 2 * cy, 2 * cx)) = (uint16_t)yo[0];
      *(uint16_t *)(L + out_y(gm, 2 * cy + 1, 2 * cx)) = (uint16_t)yo[1];
      // chroma: the U sample in byte 0, the V sample in byte 1 of every operand
      const uint32_t au = cut(D.ur[0], D.csh), bu = cut1(D.ur[0], D.csh), cu = cut(D.ur[1], D.csh), du = cut1(D.ur[1], D.csh);
      const uint32_t av = cut(D.vr[0], D.csh), bv = cut1(D.vr[0], D.csh), cvv = cut(D.vr[1], D.csh), dv = cut1(D.vr[1], D.csh);
      const uint32_t ca = __builtin_amdgcn_perm(av, au, 0x0c0c0400u), cb = __builtin_amdgcn_perm(bv, bu, 0x0c0c0400u);
      const uint32_t cc = __builtin_amdgcn_perm(cvv, cu, 0x0c0c0400u), cd = __builtin_amdgcn_perm(dv, du, 0x0c0c0400u);
      const uint32_t hca = (ca >> 1) & M, hcb = (cb >> 1) & M, hcc = (cc >> 1) & M, hcd = (cd >> 1) & M;
      const uint32_t c1 = hca + hcb, c2 = hca + hcc, c3 = ((c1 >> 1) & M) + (((hcc + hcd) >> 1) & M);
      const uint32_t uvp = D.cph == 0 ? ca : D.cph == 1 ? c1 : D.cph == 2 ? c2 : c3;
      L[out_c(gm, cy, 0, cx)] = (uint8_t)uvp;
      L[out_c(gm, cy, 1, cx)] = (uint8_t)(uvp >> 8);
    } else {
      const uint32_t va = mc4_select(D.wa, D.pha), cpred = mc4_select(D.wq, D.phq);
      *(uint32_t *)(L + out_y(gm, yrow, yc4)) = va;
      if (lane < 32) *(uint32_t *)(L + out_c(gm, crow, cv, cc4)) = cpred;
    }
  };
  Deep D0;
  if (slow_mask) deep_fetch(D0, __builtin_ctz(slow_mask), cell0);

  __builtin_amdgcn_s_setprio(0); // (behind the deep trees' requests too: 7.17 -> 7.10 ms; priority during the motion compensation costs: 7.27)
  MOBI_STOP(3);
  // ---- stage B: motion compensation.  Lane (g, rr = j >> 2, q = j & 3) = luma rows 8rr..8rr+7, pixels 4q..4q+3; lane (g, pl = j >> 2,
  // ch = (j >> 1) & 1, qc = j & 1) = plane pl, chroma rows 4ch..4ch+3, samples 4qc..4qc+3: either lies inside one leaf whatever the split ----
  uint32_t mcv[12];
  const int rr = j >> 2, q = j & 3;
  const int pl = j >> 2, ch = (j >> 1) & 1, qc = j & 1;
  {
    // chroma first: its windows make room for the dequant scales
    const bool cBl = (tb && ch) || (lr && qc);
    const int ctop = cBl ? ctopB : cposA, cph = cBl ? cphB : cphA;
    const int cxo = ctop & 7, d0 = (cxo + (lr ? 0 : 4 * qc)) >> 2, d1 = d0 + 1;
    const int base = ((lr && qc) ? P_C1 : P_C) + ((tb && ch) ? 5 : 4 * ch) * 256 + g * 16 + pl * 8;
    const int a0 = base + (d0 >> 1) * 128 + (d0 & 1) * 4, a1 = base + (d1 >> 1) * 128 + (d1 & 1) * 4;
    uint32_t x0[5], x1[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { x0[k] = lds32(L, a0 + k * 256); x1[k] = lds32(L, a1 + k * 256); }
    mc_rows<4>(x0, x1, (uint32_t)cxo & 3u, cph, mcv + 8);
  }
  MOBI_STOP(4);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane < MOBI_SCALE_STRIDE / 4) MOBI_DMA16((const uint8_t *)(A.scale + quant * MOBI_SCALE_STRIDE) + lane * 16, L + P_SC, 0);
  {
    // window row w of the lane's leaf sits in pair (yo + w) >> 1, row (yo + w) & 1 of it (yo = the window's first row is odd)
    const bool rB = tb && rr, qB = lr && q >= 2;
    const int ytop = rB || qB ? topB : posA, yph = rB || qB ? phB : phA;
    const int yo = (ytop >> lgS) & 1, xo = ytop & 7;
    const int d0 = (xo + 4 * (qB ? q - 2 : q)) >> 2, d1 = d0 + 1;
    const int base = P_L + (rB ? 5 : 4 * rr) * 512 + g * 16 + (qB ? 256 : 0);
    const int a0 = base + (d0 >> 1) * 128 + (d0 & 1) * 4, a1 = base + (d1 >> 1) * 128 + (d1 & 1) * 4;
    const int ev = yo * 8, od = yo ? 512 : 8;
    uint32_t x0[9], x1[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
      const int o = (k >> 1) * 512 + ((k & 1) ? od : ev);
      x0[k] = lds32(L, a0 + o);
      x1[k] = lds32(L, a1 + o);
    }
    mc_rows<8>(x0, x1, (uint32_t)xo & 3u, yph, mcv);
  }
  MOBI_STOP(5);
  wave_sync();
  {
    // out_px with the row's low bits constant: (g ^ k) << 4 = (g << 4) ^ (k << 4), and nothing else of the address lives in bits 4..6
    const int by = P_OUT + rr * 128 + (g << 4) + 4 * q, bc = P_OUT + 256 + ch * (4 * 384) + ((g << 4) ^ (ch << 6)) + pl * 8 + 4 * qc;
#pragma unroll
    for (int k = 0; k < 8; k++) *(uint32_t *)(L + ((by ^ (k << 4)) + k * 384)) = mcv[k];            // = out_y(g, 8 * rr + k, 4 * q)
#pragma unroll
    for (int k = 0; k < 4; k++) *(uint32_t *)(L + ((bc ^ (k << 4)) + k * 384)) = mcv[8 + k];        // = out_c(g, 4 * ch + k, pl, 4 * qc)
  }
  if (PROF) pt[3] = prof_stamp();
  MOBI_STOP(6);
  if (slow_mask) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the scales too)
    deep_finish(__builtin_ctz(slow_mask), D0);
    uint32_t mm = slow_mask & (slow_mask - 1);
    bool second = true;
    while (mm) { // a second, third ... such macroblock in the same octet: exposed round trips (rare)
      const int gm = __builtin_ctz(mm);
      mm &= mm - 1;
      uint32_t cell = cell1;
      if (!second) load_cell(gm, cell);
      second = false;
      Deep Dn;
      deep_fetch(Dn, gm, cell);
      asm volatile("" ::: "memory");
      deep_finish(gm, Dn);
    }
  }
  if (PROF) pt[4] = prof_stamp();
  MOBI_STOP(7);

  // ---- stage C: residual ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the scales
  wave_sync();
  if (m_lo | m_hi) {
    // slots: the 8x8 areas first, then -- from an even slot, so that the two areas of a packed pair are of one kind -- the areas made of
    // 4x4 blocks; inside a kind by area, then macroblock.  (n8 odd: slot n8 stays empty.)
    const uint32_t m8_lo = m_lo & t_lo, m8_hi = m_hi & t_hi, m4_lo = m_lo & ~t_lo, m4_hi = m_hi & ~t_hi;
    const int n8_lo = __builtin_popcount(m8_lo), n8 = n8_lo + __builtin_popcount(m8_hi);
    const int n4_lo = __builtin_popcount(m4_lo), first4 = (n8 + 1) & ~1, n_slots = first4 + n4_lo + __builtin_popcount(m4_hi);
    {
      const bool hi = lane >= 32;
      const int kk = lane & 31;
      if (((hi ? m_hi : m_lo) >> kk) & 1) {
        const bool is8 = ((hi ? t_hi : t_lo) >> kk) & 1;
        const uint32_t flip = is8 ? 0u : 0xFFFFFFFFu;
        const uint32_t mask = hi ? m_hi & (t_hi ^ flip) : m_lo & (t_lo ^ flip);
        const int first = (is8 ? 0 : first4) + (hi ? (is8 ? n8_lo : n4_lo) : 0);
        const int slot = first + __builtin_popcount(mask & ((1u << kk) - 1u));
        const int a = lane >> 3;
        const uint32_t K = (uint32_t)((a < 4 ? a >> 1 : 2) * 128 + (a & 1) * 8);
        *(uint32_t *)(L + P_TAB + 4 * slot) = ((uint32_t)(lane & 7) << 4) | ((uint32_t)lane << 8) | (K << 16);
        *(uint32_t *)(L + P_INV + 4 * lane) = (uint32_t)((slot >> 1) * (P_TILE * 4) + (slot & 1) * 2) | ((uint32_t)slot << 13) | ((is8 ? 0x0FCu : 0x13Cu) << 23); // area * 8 + g -> slot, for the scatter
      }
      if (lane < 48) *(uint32_t *)(L + P_SUM + 4 * lane) = 0u;
    }
    wave_sync();
    MOBI_STOP(8);
    s16x2 lo = {0, 0}, hi = {0, 0};
    const int r = lane & 7, grp = lane >> 3;
    // one pass over the macroblock's level words: PK = into the int16 pair tiles of slots [base, base + 16) (and, in the first round, the
    // per-area sums of |coefficient|), else into the int32 tiles of slots [base, base + P_ROUND)
    auto scatter_all = [&](auto pk, int base) {
      constexpr bool PK = decltype(pk)::value;
      const int base_off = PK ? (base >> 1) * (P_TILE * 4) : 0; // (wave-uniform)
      auto scatter = [&](uint32_t e) {
        const uint32_t t = e & 0x1FF, t4 = t << 2, p4 = t4 & 0xFCu, kk4 = ((t >> 6) * 8 + (uint32_t)g) << 2;
        const int level = (int32_t)e >> 16;
        const uint32_t inv = lds32(L, P_INV + (int)kk4);
        const int scale = (int)lds32(L, P_SC + (int)((t4 | 0x100u) & (inv >> 23))); // scale8[p] or scale4[p & 15] (80 words: 64 + 16)
        const int v = __mul24(scale, level);
        if (PK) {
          const uint32_t off = (inv & 0x1FFFu) - (uint32_t)base_off;
          if (off < (uint32_t)(P_PAIRS * P_TILE * 4)) *(int16_t *)(L + P_COEF + off + p4) = (int16_t)v;
          if (base == 0) { // (a word is looked at once per round; the sums are complete after the first)
            const int av = v < 0 ? -v : v;
            __hip_atomic_fetch_add((uint32_t *)(L + P_SUM + kk4), (uint32_t)(av > 0xFFFF ? 0xFFFF : av), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          }
        } else {
          const int slot = (int)((inv >> 13) & 0x7Fu) - base;
          if ((unsigned)slot < (unsigned)P_ROUND) *(int *)(L + P_COEF + slot * (P_TILE * 4) + p4) = v;
        }
      };
#pragma unroll
      for (int k = 0; k < CWR; k++) {
        int lf = left;
        asm volatile("" : "+v"(lf)); // (compared where it is needed: hoisted, the sixteen comparisons of this unrolled loop become sixteen
        const bool mine = lf > 8 * k; //  0 / 1 registers in front of it and sixteen more comparisons inside)
        if (k && __builtin_amdgcn_ballot_w64(mine) == 0) break;
        uint32_t e = cwr[k];
        asm volatile("" : "+v"(e));
        if (mine) scatter(e);
      }
      // beyond the registers (dense macroblocks): 64 more words of a macroblock per round trip, eight loads in flight per lane
      for (uint32_t i = 8u * CWR + (uint32_t)j; __builtin_amdgcn_ballot_w64(i < ncoef) != 0; i += 64) {
        uint32_t tw[8];
#pragma unroll
        for (int k = 0; k < 8; k++) tw[k] = i + 8u * k < ncoef ? cw[i + 8u * k] : 0u;
#pragma unroll
        for (int k = 0; k < 8; k++)
          if (i + 8u * k < ncoef) scatter(tw[k]);
      }
    };
    auto zero_tiles = [&](int n_tiles) { // (wave-uniform: only the tiles in use)
      const uint4 z = uint4{0, 0, 0, 0};
      for (int o = lane * 16; o < n_tiles * P_TILE * 4; o += 1024) *(uint4 *)(L + P_COEF + o) = z;
    };
    // where the lane's two words of an area are: 8x8 -> row r, samples 0..3 and 4..7; 4x4 -> rows i0, i0 + 1 of block r >> 1.
    // address = C + ((g ^ x) << 4) + K(area) in the layout of out_px
    const int s4 = r >> 1, rowa4 = (s4 >> 1) * 4 + (r & 1) * 2, cola4 = (s4 & 1) * 4;
    const int Ca8 = P_OUT + r * 384, Cb8 = Ca8 + 4, x8 = r << 4;
    const int Ca4 = P_OUT + rowa4 * 384 + cola4, Cb4 = P_OUT + (rowa4 + 1) * 384 + cola4, xa4 = rowa4 << 4, xb4 = (rowa4 + 1) << 4;
    bool wide = false;  // (wave-uniform) some area's coefficients are too large for the 16-bit butterflies
    for (int base = 0; base < n_slots && !wide; base += 2 * P_PAIRS) {
      zero_tiles(n_slots - base < 2 * P_PAIRS ? (n_slots - base + 1) >> 1 : P_PAIRS);
      wave_sync();
      scatter_all(std::true_type{}, base);
      wave_sync();
      if (base == 0) {
        const uint32_t sm = lane < 48 ? lds32(L, P_SUM + 4 * lane) : 0u;
        if (__builtin_amdgcn_ballot_w64(sm > (uint32_t)MOBI_PK_LIMIT) != 0) { wide = true; break; }
      }
      MOBI_STOP(9);
      for (int sub = 0; sub < P_PAIRS && base + 2 * sub < n_slots; sub += 8) { // the pair tiles eight at a time: one per group of eight lanes
        const int slot0 = base + 2 * (sub + grp);  // the group's pair: slots slot0 (A), slot0 + 1 (B)
        const bool act = sub + grp < P_PAIRS && slot0 < n_slots, is8g = slot0 < first4;
        const bool actB = slot0 + 1 < n_slots && slot0 + 1 != n8;
        uint32_t *tile = (uint32_t *)(L + P_COEF) + P_TILE * (sub + grp);
        if (act) idct_pass1_pk(tile, is8g, r);
        wave_sync();
        MOBI_STOP(10);
        if (act) {
          const uint32_t recA = lds32(L, P_TAB + 4 * slot0), recB = actB ? lds32(L, P_TAB + 4 * slot0 + 4) : recA;
          const int Ca = is8g ? Ca8 : Ca4, Cb = is8g ? Cb8 : Cb4, xa = is8g ? x8 : xa4, xb = is8g ? x8 : xb4;
          const int gA = (int)(recA & 0x70u), KA = (int)(recA >> 16), gB = (int)(recB & 0x70u), KB = (int)(recB >> 16);
          uint8_t *wa = L + (Ca + (gA ^ xa) + KA), *wb = L + (Cb + (gA ^ xb) + KA), *wc = L + (Ca + (gB ^ xa) + KB), *wd = L + (Cb + (gB ^ xb) + KB);
          idct_pass2_pk(tile, is8g, r, wa, wb, wc, wd, actB, lo, hi);
        }
        wave_sync();
        MOBI_STOP(11);
      }
    }
    if (wide) { // the same in int32, eight lanes per area, P_ROUND areas per round (r03's residual stage)
      const int hole = (n8 & 1) ? n8 : -1; // (n8 odd: slot n8 is empty)
      for (int base = 0; base < n_slots; base += P_ROUND) {
        zero_tiles(n_slots - base < P_ROUND ? n_slots - base : P_ROUND);
        wave_sync();
        scatter_all(std::false_type{}, base);
        wave_sync();
        int *coef = (int *)(L + P_COEF);
        uint32_t recx[3];
        bool actx[3], is8x[3];
#pragma unroll
        for (int h = 0; h < 3; h++) {
          const int idx = base + 8 * h + grp;
          actx[h] = idx < n_slots && idx != hole && 8 * h + grp < P_ROUND;
          recx[h] = actx[h] ? lds32(L, P_TAB + 4 * idx) : 0u;
          is8x[h] = idx < n8;
        }
#pragma unroll
        for (int h = 0; h < 3; h++) {
          int *tile = coef + P_TILE * (8 * h + grp);
          if (actx[h]) idct_pass1(tile, tile, is8x[h], r);
        }
        wave_sync();
#pragma unroll
        for (int h = 0; h < 3; h++) {
          if (actx[h]) {
            const bool is8 = is8x[h];
            const int Ca = is8 ? Ca8 : Ca4, Cb = is8 ? Cb8 : Cb4, xa = is8 ? x8 : xa4, xb = is8 ? x8 : xb4;
            const int gg = (int)(recx[h] & 0x70u), KK = (int)(recx[h] >> 16);
            idct_pass2_q(coef + P_TILE * (8 * h + grp), is8, r, L + (Ca + (gg ^ xa) + KK), L + (Cb + (gg ^ xb) + KK), lo, hi);
          }
        }
        wave_sync();
      }
    }
    if (lo.x < -64 || lo.y < -64 || hi.x > 319 || hi.y > 319) atomicOr(&A.fault[clip], 1); // clamp table domain (MobiConst.cs:587)
  }
  if (PROF) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); pt[5] = prof_stamp(); }

  MOBI_STOP(12);
  __builtin_amdgcn_s_setprio(3);
  // ---- stage D: the octet's tiles are contiguous: 2 KB of luma, 1 KB of chroma, whole lines ----
  // Intra macroblocks' places too (whatever LDS holds there; mobi_recon_intra overwrites them), and behind the picture's last
  // macroblock (848 = 53 macroblocks: the seventh octet holds five) the zeros the padding already holds: HBM turns every store
  // below 64 B into a read-modify-write (tools/ubench/pwrite.hip), and a run with a hole has such ends.
  {
    uint8_t *y0 = clip_base + (uint32_t)A.ring_base * A.slot_bytes;
    uint8_t *ty0 = y0 + mobi_tile_y(mbx0, mby, lgS), *tc0 = y0 + ysz + mobi_tile_c(mbx0, mby, lgS);
    const bool full = nmb == 8; // (wave-uniform: only a picture's last octet can be short)
    if (FUSED) {
      // One launch for the whole step (mobi_recon_step, small batches): the intra macroblocks' waves run beside this one.  Nothing is
      // written over an intra macroblock's place (its wave may be there first); the inter macroblocks' samples go through to memory
      // (sc1), are drained, and every inter macroblock gets this step's tag, which the intra waves that read it as halo poll.
      // (An asm store of more than 64 bits needs its own two wait states before a vector instruction may write its data registers: the
      // compiler pads its own stores, not these -- without the s_nop the next round's moves into the tuple changed the first dword of
      // the lanes whose data are read last.)
#pragma unroll
      for (int it = 0; it < 2; it++) {
        const int i = lane + 64 * it, gq = i >> 4, quad = (i >> 2) & 3, R0 = (quad >> 1) * 8 + 2 * (i & 3), c0 = (quad & 1) * 8;
        const uint2 v0 = *(const uint2 *)(L + out_y(gq, R0, c0)), v1 = *(const uint2 *)(L + out_y(gq, R0 + 1, c0));
        const u32x4 v = {v0.x, v0.y, v1.x, v1.y};
        const uint8_t *dst = ty0 + i * 16;
        if ((inter_mask >> gq) & 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
      }
      {
        const int gq = lane >> 3, R = lane & 7;
        const uint4 c4 = *(const uint4 *)(L + out_c(gq, R, 0, 0));
        const u32x4 vc = {c4.x, c4.y, c4.z, c4.w};
        const uint8_t *dst = tc0 + lane * 16;
        if ((inter_mask >> gq) & 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(vc) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the samples have left this CU before the tags do
      if (lane < 8 && ((inter_mask >> lane) & 1))
        __hip_atomic_store(A.done + (size_t)clip * A.n_mbs + mby * mbw + mbx0 + lane, A.step_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
#pragma unroll
    for (int it = 0; it < 2; it++) {
      const int i = lane + 64 * it, gq = i >> 4, quad = (i >> 2) & 3, R0 = (quad >> 1) * 8 + 2 * (i & 3), c0 = (quad & 1) * 8;
      const uint2 v0 = *(const uint2 *)(L + out_y(gq, R0, c0)), v1 = *(const uint2 *)(L + out_y(gq, R0 + 1, c0));
      if (full) *(uint4 *)(ty0 + i * 16) = uint4{v0.x, v0.y, v1.x, v1.y};
      else *(uint4 *)(ty0 + i * 16) = gq < nmb ? uint4{v0.x, v0.y, v1.x, v1.y} : uint4{0, 0, 0, 0};
    }
    {
      const int gq = lane >> 3, R = lane & 7;
      const uint4 vc = *(const uint4 *)(L + out_c(gq, R, 0, 0));
      if (full) *(uint4 *)(tc0 + lane * 16) = vc;
      else *(uint4 *)(tc0 + lane * 16) = gq < nmb ? vc : uint4{0, 0, 0, 0};
    }
    }
  }
  if (PROF && lane == 0) { // MOBI_DEBUG=9: where a wave's life goes (shader clock): A issue, fetch wait, MC, deep trees, residual, store issue
    pt[6] = prof_stamp();   // one record per octet (the buffer holds 16 bytes per macroblock)
    unsigned long long *rec = A.prof + (size_t)oi * 8;
#pragma unroll
    for (int k = 0; k < 6; k++) rec[k] = pt[k + 1] - pt[k];
    rec[6] = 1ull | ((pa - pt[0]) << 8) | ((pb - pa) << 32); // + kernel arguments, descriptor
    rec[7] = (unsigned long long)(__builtin_popcount(m_lo) + __builtin_popcount(m_hi));
  }
}
// One wave per workgroup.  Tried and measured (r2, 24576 clips 640x480, same box): workgroups of 2 / 3 / 5 / 6 vertically adjacent
// octets, so that waves whose reference windows overlap share this CU's L1 -- 8.0 / 8.2 / 11.4 / 11.2 ms against 8.0 ms for
// single waves (a workgroup's LDS and wave slots come and go as a block); non-temporal window fetches -- 8.5 ms (chroma only)
// and 9.8 ms (all): the L1 hits between the macroblocks of an octet are worth more than the L1 they pollute; non-temporal row
// stores (so that output lines do not compete with window lines for the L2): 7.76 against 7.77 ms, nothing.
// Persistent waves (a wave walks many octets and asks for the next one's descriptors while it works -- a fresh wave spends 11 % of its
// life waiting for its own, MOBI_DEBUG=9): the loop around this much inlined code spills (24 VGPRs, 26 SGPRs at best: the kernel
// arguments stay live across it), and as a real function the callee-saved registers go through scratch.  Not kept.  Touching the
// descriptors of the octet an XCD starts 512 ... 8192 workgroups later (a DMA of two lines into unused LDS): 2.81 against 2.71 ms per
// 8192 clips -- two more requests per wave in a kernel bound by requests cost more than the shorter wait gives.
#define MOBI_OCT_KERNEL(NAME, WAVES, PROF, NCWR)                                                      \
  extern "C" __global__ __launch_bounds__(64, WAVES) void NAME(MobiReconArgs A) {                      \
    __shared__ __attribute__((aligned(16))) uint8_t lds[P_BYTES];                                      \
    const uint32_t oi = (blockIdx.x & 7) * A.inter_per_xcd + (blockIdx.x >> 3);                        \
    if (oi >= A.qpc * (uint32_t)A.n_clips) return;                                                     \
    uint32_t rem, ox;                                                                                  \
    const uint32_t clip = fastdiv(oi, A.qpc, A.magic_qpc, rem); /* qpr / qpc: octets per macroblock row / per clip */ \
    const uint32_t mby = fastdiv(rem, A.qpr, A.magic_qpr, ox);                                         \
    recon_inter_oct<PROF, NCWR, 0>(A, lds, clip, mby, ox, (int)threadIdx.x);                              \
  }
// 4 waves per SIMD (108 VGPRs, no spills; 5 waves = 96 VGPRs spill 5 registers since the windows became 16-byte aligned and
// measure the same) with 128 level words per macroblock in registers (96: 848x480 with its dense blocks 8 % slower; 192: no better).
// Re-measured at the end of r02 on one box: 5 waves with 64 level words (96 VGPRs, one spill) 2.696 ms per 8192 clips of 640x480
// against 2.696, and 2.81 against 2.50 on 848x480; 5 waves with 96 words (5 spills) 2.84.  Fewer waves do cost (15 per CU +3 %,
// 12 per CU +12 %, MOBI_LDS_PAD); more do not pay.  Output rows padded in LDS (pitch 144 / 80 instead of 128 / 64, so that the eight
// lanes of an area do not meet in one bank when they add the residual: 55 % of the LDS cycles are bank conflicts): 2.754 against 2.758 --
// LDS time is not on the critical path of a kernel that waits for memory requests.
MOBI_OCT_KERNEL(mobi_recon_inter8, 4, 0, 16)
#if defined(MOBI_PROFILING)
MOBI_OCT_KERNEL(mobi_recon_inter8_prof, 4, 1, 16) // (in-kernel cycle records, MOBI_DEBUG=9: the profiling twin of the library only)
#endif

// =====================================================================================================
// intra macroblocks, FOUR per wavefront (sixteen lanes each)
// =====================================================================================================
// r01 / early r02 ran one macroblock per wave: the block list of a macroblock is serial (every block predicts from the ones
// before it), so the wave walked it with scalar control -- 783 scalar + 611 vector instructions per macroblock, and a SIMD issues
// one scalar instruction every four clocks: the kernel was bound by scalar issue.  Here a macroblock is a ROW of 16 lanes (the DPP
// row), a wave carries four independent macroblocks of the same dependency level, and the serial list is data: one step per
// unsplit area (an 8x8 block's predictors read only samples outside the block: 16 lanes x 4 samples do it at once), four per split
// area (4x4 blocks, one sample per lane), 6..24 steps, two descriptor words per step, built once by the lane that holds the block's
// record and read back by all 16.  The wave runs as many steps as its longest macroblock has (95 % of the areas are unsplit in the
// generator's mix); the wave-level branches inside a step ask "does any of the four need a plane / a DC / an 8x8 / a 4x4 now".
namespace {
// LDS of one macroblock, 1888 bytes, two lives:
//   while the residuals are made    [0, 1536) coefficients, int32 [6][64]               [1536, 1856) dequant scales
//   from then on                    [0, 768) residuals, int16 [6][64]   [768, 1888) the three tiles, the step descriptors in their slack
// (pass 2 of the transforms writes the int16 residuals of areas 2k, 2k + 1 over the int32 words of area k, which is done with by then;
// r02's first version kept all three side by side: 2848 bytes, 14 waves per CU instead of 19, and the launch is latency-bound
// wherever it is not bound by its stores.)
enum { IQ_TILE = 768,                                                      // bytes
       IQ_TCU = 17 * TP, IQ_TCV = IQ_TCU + 9 * TP,                         // chroma tiles behind the luma tile (bytes from the tile's start)
       IQ_SCALE = 1536 / 4, IQ_WORDS = (IQ_TILE + IQ_TCV + 9 * TP) / 4, // words
       IQ_TAB = 1856 };                                                   // bytes: 32 free ones between the scales and the end, while the residuals are made
static_assert(IQ_WORDS * 4 == 1888 && IQ_SCALE * 4 + 320 <= IQ_TAB && IQ_TAB + 24 <= IQ_WORDS * 4 && 4 * IQ_WORDS * 4 <= 6 * 1280, "intra LDS map");
// The step descriptors (two words per step, at most 24 steps) live in the chroma tiles' slack: a chroma tile row holds columns -4..15 in its
// first 20 bytes and nothing in the other 12 -- three words per row, 54 in the 18 rows.  (r04: they had 192 bytes of their own behind the
// tiles; without them a wave's LDS is 7552 bytes = six allocation granules of 1280 instead of seven: the LDS would hold 21 waves per CU instead of 18; at 85 registers the SIMDs hold 20.)
__device__ __forceinline__ int step_word(int k) { const int row = (k * 43) >> 7; return IQ_TILE + IQ_TCU + row * TP + 20 + 4 * (k - 3 * row); } // byte offset of word k < 54
// step descriptor, word 0
enum { SD_O = 0,           // [10:0]  byte offset of the block's top-left sample inside the macroblock's tiles
       SD_TAP = 11,        // [20:11] first tap table entry of this step (+ lane)
       SD_IS4 = 1 << 21,   // 4x4 block (lane = y * 4 + x); else an 8x8 block (lane = row * 2 + half row: four samples)
       SD_CODED = 1 << 22, // add the residual
       SD_KTAP = 1 << 23, SD_KDC = 1 << 24, SD_KPLANE = 1 << 25, // predictor kind; none of them: what is there stays (plane passes, mode 9)
       SD_TA = 1 << 26, SD_LA = 1 << 27,                         // DC: row above / column to the left available (MD.cs:1923-1924)
       SD_P4 = 1 << 30 };  // the plane is a 4x4 one
// word 1: [8:0] index of the step's first residual (+ lane's), [13:9] 1 + the plane parameter's index among the macroblock's wide parameters
//         (mobi_cmd.h: a parameter that does not fit 16 bits; 0: it does), [31:16] plane parameter

__device__ __forceinline__ uint32_t ldg_u8_sc1(const uint8_t *p) { // past this CU's L1; valid after vm_wait*
  uint32_t v = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("global_load_ubyte %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
#endif
  return v;
}
// LDS address + a signed 16-bit half of a tap-table word, in one instruction (SDWA: the operand selects the half and sign-extends it):
// the directional predictors read four neighbour samples per predicted sample at tile offsets that come as int16 pairs
__device__ __forceinline__ uint32_t tap_addr(uint32_t base, uint32_t packed, int hi) {
  uint32_t d = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  if (hi) asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(d) : "v"(base), "v"(packed));
  else asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(d) : "v"(base), "v"(packed));
#endif
  return d;
}
typedef const uint8_t __attribute__((address_space(3))) *lds_u8p;
__device__ __forceinline__ uint32_t tap4(uint32_t base, uint32_t w0, uint32_t w1) { // (t0 + t1 + t2 + t3 + 2) >> 2 over the four taps of two table words
  const uint32_t a = *(lds_u8p)tap_addr(base, w0, 0), b = *(lds_u8p)tap_addr(base, w0, 1), c = *(lds_u8p)tap_addr(base, w1, 0), d = *(lds_u8p)tap_addr(base, w1, 1);
  return (a + b + (c + d + 2u)) >> 2;
}
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2 ldg_x2_sc1(const uint8_t *p) {
  u32x2 v = {0, 0};
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
#endif
  return v;
}
// The wait sits right behind its loads and names every destination: an asm load's register is only reserved up to the asm
// statement, so a wait further away lets the register allocator reuse it while the data is still in flight (seen in r02).
__device__ __forceinline__ void vm_wait(u32x2 &a, uint32_t &b, uint32_t &c) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c) : : "memory");
#endif
}
__device__ __forceinline__ void vm_wait7(uint32_t (&v)[7]) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]) : : "memory");
#endif
}
struct QNb { // neighbour samples of the block at tile offset o
  const uint8_t *t;
  int o;
  __device__ __forceinline__ int operator()(int dy, int dx) const { return t[o + dy * TP + dx]; }
};
__device__ __forceinline__ int row_sum16(int v) { // sum over the 16 lanes of a DPP row, left in all of them
#if defined(__HIP_DEVICE_COMPILE__)
  v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true); // row_ror:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, true); // row_ror:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x122, 0xF, 0xF, true); // row_ror:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x121, 0xF, 0xF, true); // row_ror:1
#endif
  return v;
}
__device__ __forceinline__ uint32_t quad_first(uint32_t v) { // lane 4k's value in lanes 4k..4k+3
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xF, 0xF, true); // quad_perm:[0,0,0,0]
#else
  return v;
#endif
}

// What a row of lanes needs to know to start on its macroblock.  The host-built launch list (mobi_abi.cpp, LevelPlan) carries it
// as one 16-byte item, so the first load already says where the records are and whether anybody has to be waited for.
struct QItem {
  bool valid;        // false: padding (levels are padded to whole waves); the row works on zeros and stores nothing
  uint32_t clip, mb;
  uint32_t w1;       // MbDesc.w1
  uint32_t pay;      // MbDesc.payload_off
  uint32_t w3;       // MbDesc.w3: [0] 16x16 plane present, [4] its parameter is a wide one, [31:16] its parameter
  uint32_t ncoef;
  bool has_deps;     // some macroblock its halo reads is an intra one of this step: poll the tags, read the halo afterwards
  bool publish;      // an intra macroblock of this step may read these pixels: write through, drain, publish the tag
};
} // namespace

template <int FUSED = 0> // FUSED: the step's inter macroblocks run in the same launch (mobi_recon_step): wait for the ones the halo reads as well
__device__ __forceinline__ void recon_intra_quad(const MobiReconArgs &A, uint32_t *Lw, const QItem &I, int lane, int dbg = 0) {
  const int l = lane & 15;
  uint32_t *G = Lw + (lane >> 4) * IQ_WORDS;
  uint8_t *tile = (uint8_t *)G + IQ_TILE;
  int *coef = (int *)G;
  const int16_t *res16 = (const int16_t *)G;
  uint8_t *Gb = (uint8_t *)G;
  const int S = A.stride, lgS = 31 - __builtin_clz((unsigned)S), mbw = A.mbw;
  const uint32_t clip = I.clip, mb = I.mb, w1 = I.w1, w3 = I.w3, ncoef = I.ncoef;
  const int t8 = (w1 >> 14) & 0x3F;
  const uint32_t *rec = A.payload + (size_t)I.clip * A.pay_clip_words + I.pay;
  const int mby = (int)(((float)mb + 0.5f) / (float)mbw), mbx = (int)mb - mby * mbw; // mb < 8192: the quotient is never within rounding of an integer
  const int off = ((mby * 16) << lgS) + mbx * 16;                                       // < 2^20: the macroblock's linear offset (MD.cs:212-217)
  uint8_t *y0 = A.planes + (size_t)clip * A.clip_bytes + (size_t)(A.ring_base % 6) * A.slot_bytes;
  uint8_t *uv0 = y0 + (size_t)S * A.height;
  uint8_t *ty = y0 + mobi_tile_y((uint32_t)mbx, (uint32_t)mby, lgS), *tc = uv0 + mobi_tile_c((uint32_t)mbx, (uint32_t)mby, lgS); // its tiles (mobi_tile.h)
  const uint2 *taps = (const uint2 *)(A.scale + MOBI_SCALE_ROWS * MOBI_SCALE_STRIDE); // 4480 B every wave reads: they stay in the L1

  __builtin_amdgcn_s_setprio(3); // (as in the inter kernel: requests first; 1.173 -> 1.162 ms)
  // ---- everything that can be asked for at once: block records, dequant scales, the first 64 level words, the halo ----
  const uint32_t recA = I.valid ? rec[l] : 0u, recB = I.valid && l < MOBI_INTRA_RECORDS - 16 ? rec[16 + l] : 0u;
  const uint4 *sc_g = (const uint4 *)(A.scale + ((w1 >> 20) & 63) * MOBI_SCALE_STRIDE);
  const uint4 sc0 = sc_g[l], sc1 = sc_g[16 + (l & 3)];
  uint32_t cw[8]; // the first 128 level words of the macroblock
#pragma unroll
  for (int k = 0; k < 8; k++) cw[k] = (uint32_t)(l + 16 * k) < ncoef && !(dbg & 32) ? rec[MOBI_INTRA_RECORDS + l + 16 * k] : 0u;
  // Halo.  Away from the picture's left, right and top edges ownership needs no arithmetic: the row above (left, above, above-right
  // macroblocks) and the column to the left are raster-earlier; everything to the right in the macroblock's own rows is
  // raster-later and reads the fresh plane's 0.  Row above = the last rows of the neighbours' bottom quadrants / last chroma rows: ten
  // 8-byte loads (luma columns -4..23, U and V -4..15 each; the tiles keep column c at byte 4 + c).  Left column: two bytes per lane out
  // of the left neighbour's right quadrants (2 lines) and its chroma tile (1 line) -- in the reference's linear planes they were 32
  // different lines, which is what r02's edge side buffer was for.
  // At the picture's edges too, as long as the planes have padding columns (Width < Stride: every stream but 256- and 512-wide ones):
  // what lies outside the picture is then the padding's zeros or a negative offset -- nobody's, so it reads as the fresh plane's 0 --
  // and the pieces a macroblock does have are where they are for an interior one.  (Width == Stride: the reference's linear offsets
  // wrap into the neighbouring rows' pixels; those macroblocks ask for every halo sample who owns its address, below.)
  const bool interior = I.valid && ((mbx >= 1 && mbx + 1 < mbw && mby >= 1) || A.width < S);
  const bool hasW = mby >= 1 && (l == 0 || l == 4 || l == 5 ? mbx >= 1 : l == 3 || l == 8 || l == 9 ? mbx + 1 < mbw : true); // the lane's piece of the row above exists
  const bool hasL = mbx >= 1;
  const uint8_t *wp = ty, *b0p = ty, *b1p = ty; // lanes with nothing to fetch read the macroblock's own first sample and drop it
  if (interior) {
    const int up = S << 4;                                 // a tile row of luma tiles in bytes (S / 16 tiles of 256 B); chroma: half
    if (hasW) {
      if (l == 0) wp = ty - up - 256 + 192 + 56;             // above-left, BR quadrant, row 7: columns -8..-1
      else if (l < 3) wp = ty - up + (l == 1 ? 128 : 192) + 56; // above, BL / BR, row 7
      else if (l == 3) wp = ty - up + 256 + 128 + 56;        // above-right, BL, row 7
      else if (l < 10) wp = tc - (up >> 1) + ((l - 4) >> 1) * 128 - 128 + 112 + (l & 1) * 8; // chroma row 7 of above-left / above / above-right: U, V
    }
    if (hasL) {
      b0p = ty - 256 + (1 + 2 * (l >> 3)) * 64 + (l & 7) * 8 + 7;
      b1p = tc - 128 + (l & 7) * 16 + (l >> 3) * 8 + 7;
    }
  }
  // where the 8 bytes go: column c at byte 4 + c of tile row 0; the above-left pieces keep their last four columns only
  const int wdst = l < 4 ? (l == 0 ? 0 : 8 * l - 4) : ((l & 1) ? IQ_TCV : IQ_TCU) + (l < 6 ? 0 : l < 8 ? 4 : 12);
  const bool whalf = l == 0 || l == 4 || l == 5;
  const int b0dst = (l + 1) * TP + 3, b1dst = (l < 8 ? IQ_TCU : IQ_TCV) + ((l & 7) + 1) * TP + 3;
  // ordinary loads, in flight beside the records -- for the rows that will not have to wait for a producer: the others load below, behind
  // their producers' tags, and would drop these (every macroblock of an I-frame but the first; every row of a one-launch step)
  uint2 w_early = uint2{0, 0};
  uint32_t b0_early = 0, b1_early = 0;
  if (!(dbg & 16) && !(I.valid && (FUSED || I.has_deps))) { w_early = *(const uint2 *)wp; b0_early = *b0p; b1_early = *b1p; }

  __builtin_amdgcn_s_setprio(0);
  { // zero the coefficients; dequant scales behind them
    uint4 *G4 = (uint4 *)G;
    const uint4 z = uint4{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 6; k++) G4[l + 16 * k] = z;
    G4[IQ_SCALE / 4 + l] = sc0;
    if (l < 4) G4[IQ_SCALE / 4 + 16 + l] = sc1;
  }

  // (the dependency list is asked for with everything else; the wait for the producers themselves comes as late as it can: below,
  // in front of the halo's placement)
  const bool waits = I.valid && (FUSED || I.has_deps);
  uint32_t wd = 0;
  if (waits && l < MOBI_INTRA_DEPS) wd = (&(A.desc + (size_t)clip * A.n_mbs + mb)->w4)[l >> 1];
  MOBI_ISTOP(1);

  // ---- dequantise and scatter the level words ----
  wave_sync();
  auto scatter = [&](uint32_t e) {
    const int t = e & 0x1FF, level = (int32_t)e >> 16, p = t & 63;
    const int si = ((t8 >> (t >> 6)) & 1) ? p : 64 + (p & 15);
    coef[t] = __mul24((int)G[IQ_SCALE + si], level);
  };
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const bool mine = (uint32_t)(l + 16 * k) < ncoef && !(dbg & 4);
    if (k && __builtin_amdgcn_ballot_w64(mine) == 0) break; // (nobody in the wave has that many)
    if (mine) scatter(cw[k]);
  }
  for (uint32_t base = 128; __builtin_amdgcn_ballot_w64(base < ncoef) != 0; base += 128) { // dense macroblocks: 128 more per round trip
#pragma unroll
    for (int k = 0; k < 8; k++) cw[k] = base + (uint32_t)(l + 16 * k) < ncoef ? rec[MOBI_INTRA_RECORDS + base + l + 16 * k] : 0u;
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (base + (uint32_t)(l + 16 * k) < ncoef) scatter(cw[k]);
  }
  wave_sync();
  MOBI_ISTOP(2);

  // ---- residuals of the coded areas.  Eight lanes per area; the coded areas of all four macroblocks are taken together, eight per
  // round (a P-frame's intra macroblocks have ~2 of 6 coded: one round instead of the three that "two areas of each macroblock per
  // round" took; an I-frame's ~3: two).  Pass 1 in place; pass 2 leaves int16 residuals (saturated: whatever does not fit is a
  // clamp-table fault anyway) in registers until every round has read its coefficients, then over the words of areas 0..2 ----
  {
    const uint32_t cb = (I.valid && !(dbg & 1)) ? (w1 >> 8) & 0x3Fu : 0u;
    const uint32_t M = (uint32_t)__builtin_amdgcn_readlane((int)cb, 0) | ((uint32_t)__builtin_amdgcn_readlane((int)cb, 16) << 6) |
                       ((uint32_t)__builtin_amdgcn_readlane((int)cb, 32) << 12) | ((uint32_t)__builtin_amdgcn_readlane((int)cb, 48) << 18); // bit mb * 6 + area
    const int n_act = __builtin_popcount(M);
    uint8_t *tab = (uint8_t *)Lw + IQ_TAB; // k-th coded area -> its bit number (in the first macroblock's free bytes behind its scales)
    if (lane < 24 && ((M >> lane) & 1)) tab[__builtin_popcount(M & ((1u << lane) - 1u))] = (uint8_t)lane;
    wave_sync();
    const int r = lane & 7;
    uint4 q[3];
    int idx[3];
    bool is8r[3];
#pragma unroll
    for (int rd = 0; rd < 3; rd++) {
      q[rd] = uint4{0, 0, 0, 0};
      idx[rd] = -1;
      is8r[rd] = false;
      if (rd * 8 < n_act) {
        const int k = rd * 8 + (lane >> 3);
        const bool act = k < n_act;
        const int id = act ? tab[k] : 0, mbi = (id * 43) >> 8, a = id - 6 * mbi; // (id / 6 for id < 24)
        const int t8x = __builtin_amdgcn_ds_bpermute(mbi << 6, t8);               // that macroblock's 8x8 / 4x4 mask
        const bool is8a = (t8x >> a) & 1;
        int *cx = (int *)(Lw + mbi * IQ_WORDS) + 64 * a;
        if (act) idct_pass1(cx, cx, is8a, r);
        wave_sync();
        if (act) {
          q[rd] = idct_pass2_pk(cx, is8a, r);
          idx[rd] = id;
          is8r[rd] = is8a;
        }
      }
    }
    wave_sync(); // every lane has read its coefficients: the residuals may land on them
#pragma unroll
    for (int rd = 0; rd < 3; rd++) {
      if (rd * 8 < n_act && idx[rd] >= 0) {
        const int mbi = (idx[rd] * 43) >> 8, a = idx[rd] - 6 * mbi;
        uint8_t *dst = (uint8_t *)(Lw + mbi * IQ_WORDS) + 128 * a; // the area's int16 [8][8]
        if (is8r[rd]) {
          *(uint4 *)(dst + 16 * r) = q[rd];
        } else { // 4x4 blocks: lane r made rows i0, i0 + 1 of block r >> 1
          const int sb = r >> 1, row0 = (sb >> 1) * 4 + (r & 1) * 2, c0 = (sb & 1) * 4;
          *(uint2 *)(dst + 16 * row0 + 2 * c0) = uint2{q[rd].x, q[rd].y};
          *(uint2 *)(dst + 16 * (row0 + 1) + 2 * c0) = uint2{q[rd].z, q[rd].w};
        }
      }
    }
    wave_sync();
  }
  MOBI_ISTOP(3);

  // ---- tiles: zero (what nobody owns yet reads 0, as the reference's fresh plane does), then the halo ----
  {
    uint4 *T4 = (uint4 *)tile; // 1120 bytes
    const uint4 z = uint4{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++) T4[l + 16 * k] = z;
    if (l < (IQ_TCV + 9 * TP) / 16 - 64) T4[64 + l] = z;
  }
  wave_sync();
  MOBI_ISTOP(4);

  // ---- the schedule: one step per unsplit area, four per split one, in decode order.  Lane 4a + s of the row owns the candidate
  // (area a, block s); its place in the list follows from how many areas before a are split. ----
  int n_iter;
  {
    const uint64_t balA = __builtin_amdgcn_ballot_w64(((recA >> 5) & 1) != 0 && (l & 3) == 0);          // areas 0..3: lanes 0, 4, 8, 12 of the row
    const uint64_t balB = __builtin_amdgcn_ballot_w64(((recB >> 5) & 1) != 0 && (l & 3) == 0 && l < 8); // areas 4, 5: lanes 0, 4
    const uint32_t mA = (uint32_t)(balA >> (lane & 48)) & 0x1111u, mB = (uint32_t)(balB >> (lane & 48)) & 0x11u;
    const uint32_t splits = (mA & 1) | ((mA >> 3) & 2) | ((mA >> 6) & 4) | ((mA >> 9) & 8) | ((mB & 1) << 4) | ((mB >> 4) << 5);
    const int nst = 6 + 3 * __builtin_popcount(splits);
    n_iter = max(max(__builtin_amdgcn_readlane(nst, 0), __builtin_amdgcn_readlane(nst, 16)), max(__builtin_amdgcn_readlane(nst, 32), __builtin_amdgcn_readlane(nst, 48)));
    // rows with fewer steps than the longest of the four idle through the rest: a descriptor that does nothing
    const uint2 idle = uint2{(uint32_t)(TP + 4), 0u};
    auto put_step = [&](int t, uint2 d) {
      *(uint32_t *)(Gb + step_word(2 * t)) = d.x;
      *(uint32_t *)(Gb + step_word(2 * t + 1)) = d.y;
    };
    put_step(l, idle);
    if (l < 8) put_step(16 + l, idle);
    wave_sync();
    auto build = [=](int t, uint32_t rown, uint32_t r0, int &pos) -> uint2 {
      const int a = t >> 2, s = t & 3;
      const bool split = (r0 >> 5) & 1, pre = (r0 >> 6) & 1;
      pos = (s == 0 || split) ? a + 3 * __builtin_popcount(splits & ((1u << a) - 1u)) + s : -1;
      const uint32_t rs = split ? rown : r0;
      const int mode = (int)(rs & 15);
      const bool coded = (rs >> 4) & 1, luma = a < 4;
      const int by = (luma ? (a >> 1) * 8 : 0) + (split ? (s >> 1) * 4 : 0), bx = (luma ? (a & 1) * 8 : 0) + (split ? (s & 1) * 4 : 0);
      const int o_blk = (luma ? 0 : a == 4 ? IQ_TCU : IQ_TCV) + (by + 1) * TP + 4 + bx;
      const bool k_dc = mode == 3, k_tap = mode < 2 || (mode >= 4 && mode <= 8);
      const bool plane_blk = mode == 2, plane_pre = pre && s == 0;
      const uint32_t param = plane_blk ? rs >> 16 : r0 >> 16;
      const int mi = !k_tap ? 0 : mode < 2 ? mode : mode - 2;
      const int tapbase = split ? MOBI_TAP_4X4 + mi * 16 : mi * 64;
      const int boff = (luma ? off : (off >> 1) + (a - 4) * (S >> 1)) + by * S + bx;
      const bool vfix = !luma && (boff & (S - 1)) >= (S >> 1);                                              // MD.cs:1886
      const bool la = ((boff - (vfix ? (S >> 1) : 0)) & (S - 1)) != 0, ta = boff >= S;                      // :1923-1924
      const int ri = a * 64 + (split ? (s >> 1) * 32 + (s & 1) * 4 : 0);
      const uint32_t widx = ((plane_blk ? rs : plane_pre ? r0 : 0u) & MOBI_REC_WIDE) ? (uint32_t)(a * 4 + (split ? s : 0) + 1) : 0u;
      uint32_t d0 = (uint32_t)o_blk | ((uint32_t)tapbase << SD_TAP);
      if (split) d0 |= SD_IS4;
      if (coded) d0 |= SD_CODED;
      if (k_tap) d0 |= SD_KTAP;
      if (k_dc) d0 |= SD_KDC;
      if (plane_blk || plane_pre) d0 |= SD_KPLANE;
      if (ta) d0 |= SD_TA;
      if (la) d0 |= SD_LA;
      if (plane_blk && split) d0 |= SD_P4;
      return uint2{d0, (uint32_t)ri | (widx << 9) | (param << 16)};
    };
    int posA, posB;
    const uint2 dA = build(l, recA, quad_first(recA), posA);
    const uint2 dB = build(16 + (l & 7), recB, quad_first(recB), posB);
    if (posA >= 0) put_step(posA, dA);
    if (l < 8 && posB >= 0) put_step(posB, dB);
  }
  wave_sync();
  MOBI_ISTOP(5);

  // All dependency levels of a frame step run in ONE launch: items are sorted by level, workgroups are dispatched in order, and a
  // row waits HERE until the intra macroblocks its halo reads (MbDesc.w4..w7) carry this step's tag -- with its level words scattered,
  // its residuals transformed and its step list built (r04: the wait stood in front of all that; in a raster chain -- an I-frame, a
  // small batch -- every link then started its own work only when its producer was done).  Hand-off across CUs: the
  // producer stores pixels write-through (sc1), drains them, then publishes its tag; the consumer polls the tag with agent-scope
  // loads and reads the halo with sc1 loads, so neither a stale L1 line nor a dirty L2 line can sit in between.
  u32x2 wv = {w_early.x, w_early.y};
  uint32_t b0 = b0_early, b1 = b1_early;
  if (__builtin_amdgcn_ballot_w64(waits) != 0) {
    if (waits && l < MOBI_INTRA_DEPS) {
      const uint32_t dep = (wd >> (16 * (l & 1))) & 0xFFFFu;
      if (dep != MOBI_DEP_NONE && (FUSED || !(dep & MOBI_DEP_INTER))) { // (two launches: the inter macroblocks ran in the one before)
        const uint32_t *f = A.done + (size_t)clip * A.n_mbs + (dep & 0x1FFFu);
        int spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != A.step_tag) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > (1 << 21)) { atomicOr(&A.fault[clip], 2); break; } // a producer that never ran: report, do not hang
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    u32x2 wl = ldg_x2_sc1(wp);
    uint32_t c0 = ldg_u8_sc1(b0p), c1 = ldg_u8_sc1(b1p);
    vm_wait(wl, c0, c1);
    if (waits) { wv = wl; b0 = c0; b1 = c1; }
  }
  if (interior) {
    if (l < 10 && hasW) {
      if (!whalf) *(uint32_t *)(tile + wdst) = wv.x;
      *(uint32_t *)(tile + wdst + (whalf ? 0 : 4)) = wv.y;
    }
    if (hasL) {
      tile[b0dst] = (uint8_t)b0;
      tile[b1dst] = (uint8_t)b1;
    }
  }
  // At the picture's edges the linear offsets of the reference wrap into the previous / next row or fall into the padding: every
  // halo sample asks who owns its address (217 luma + 178 chroma samples, 7 per lane and round).  One macroblock in twelve; the
  // host sorts them to the end of their level so that few waves come here.
  if (__builtin_amdgcn_ballot_w64(I.valid && !interior) != 0) {
    const Geo g{A.width, A.height, S, mbw, lgS};
    const bool mine = I.valid && !interior;
#pragma unroll 1
    for (int round = 0; round < 4; round++) {
      const bool chroma = round >= 2;
      uint32_t hv[7];
      int hp[7];
#pragma unroll
      for (int k = 0; k < 7; k++) {
        const int i = l + 16 * (7 * (round & 1) + k);
        int r, c, a, o, pos;
        bool in;
        if (!chroma) {
          if (i < 25) { r = -1; c = i - 1; }
          else if (i < 41) { r = i - 25; c = -1; }
          else { r = (i - 41) >> 3; c = 16 + ((i - 41) & 7); }
          a = off + r * S + c;
          o = g.owner_luma(a);
          in = i < 25 + 16 + 128;
          pos = (r + 1) * TP + 4 + c;
        } else {
          const int v = i >= 89, j = v ? i - 89 : i;
          if (j < 17) { r = -1; c = j - 1; }
          else if (j < 25) { r = j - 17; c = -1; }
          else { r = (j - 25) >> 3; c = 8 + ((j - 25) & 7); }
          a = (off >> 1) + v * (S >> 1) + r * S + c;
          o = g.owner_chroma(a);
          in = i < 2 * (17 + 8 + 64);
          pos = (v ? IQ_TCV : IQ_TCU) + (r + 1) * TP + 4 + c;
        }
        const bool take = mine && in && o >= 0 && o < (int)mb;
        hp[k] = take ? pos : -1;
        const uint32_t ta = take ? (uint32_t)a : 0u; // not ours to read: load the plane's first sample instead, and drop it
        hv[k] = ldg_u8_sc1(chroma ? uv0 + mobi_tc(ta, lgS) : y0 + mobi_ty(ta, lgS));
      }
      vm_wait7(hv);
#pragma unroll
      for (int k = 0; k < 7; k++)
        if (hp[k] >= 0) tile[hp[k]] = (uint8_t)hv[k];
    }
  }
  wave_sync();

  // ---- 16x16 plane (MD.cs:3017-3166): 64 words, four per lane ----
  if (__builtin_amdgcn_ballot_w64((w3 & 1) != 0) != 0) {
    if (w3 & 1) {
      const QNb nb{tile, TP + 4};
      const int param = (w3 & MOBI_W3_WIDE) ? (int)rec[MOBI_INTRA_RECORDS + ncoef + 24] : (int)(int16_t)(w3 >> 16);
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int w = u * 16 + l, yy = w >> 2, x0 = (w & 3) * 4;
        *(uint32_t *)(tile + TP + 4 + yy * TP + x0) = mobi_plane_word(16, param, yy, x0, nb);
      }
    }
    wave_sync();
  }

  // ---- the steps.  An 8x8 block is one step, four samples per lane (its predictors read only samples outside the block); a 4x4
  // block one sample per lane ----
  const uint32_t tile_lds = (uint32_t)(size_t)(lptr_t)tile; // the tile as an LDS address
  const int po8 = (l >> 1) * TP + (l & 1) * 4, ro8 = (l >> 1) * 8 + (l & 1) * 4;
  const int po4 = (l >> 2) * TP + (l & 3), ro4 = (l >> 2) * 8 + (l & 3);
  int fault = 0;
  s16x2 flo = {0, 0}, fhi = {0, 0}; // range of prediction + residual over the 8x8 steps (clamp table domain [-64, 319], MobiConst.cs:587)
  // The tap table entries of a directional block do not depend on pixels: they are asked for one step ahead (two register sets taken in
  // turn: r03 rotated three sets through eleven 64-bit moves per step).
  auto load_step = [&](int t, uint2 &d, uint4 &ea, uint4 &eb) {
    d = uint2{*(const uint32_t *)(Gb + step_word(2 * t)), *(const uint32_t *)(Gb + step_word(2 * t + 1))};
    const uint2 *tp = taps + (((d.x >> SD_TAP) & 0x3FF) + ((d.x & SD_IS4) ? l : 4 * l));
    ea = *(const uint4_a4 *)tp;
    eb = *(const uint4_a4 *)(tp + 2);
  };
  auto do_step = [&](const uint2 d, const uint4 ea, const uint4 eb) {
    const int o = (int)(d.x & 0x7FF);
    const bool is4 = (d.x & SD_IS4) != 0;
    if (__builtin_amdgcn_ballot_w64((d.x & SD_KPLANE) != 0) != 0) { // plane with delta, 8x8 (MD.cs:3168-3251) or 4x4 (:3253-3327): a lane makes a word of four samples
      const bool p4 = (d.x & SD_P4) != 0;
      if ((d.x & SD_KPLANE) && l < (p4 ? 4 : 16)) {
        const int yy = p4 ? l : l >> 1, x0 = p4 ? 0 : (l & 1) * 4;
        const uint32_t widx = (d.y >> 9) & 31u;
        const int param = widx ? (int)rec[MOBI_INTRA_RECORDS + ncoef + widx - 1] : (int)(int16_t)(d.y >> 16);
        *(uint32_t *)(tile + o + yy * TP + x0) = mobi_plane_word(p4 ? 4 : 8, param, yy, x0, QNb{tile, o});
      }
      wave_sync();
    }
    int dcv = 0;
    if (__builtin_amdgcn_ballot_w64((d.x & SD_KDC) != 0) != 0) { // DC with availability (MD.cs:1920-2022, :2501-2580)
      const int n = is4 ? 4 : 8, lgn = is4 ? 2 : 3;
      const bool top = l < n, ta = (d.x & SD_TA) != 0, la = (d.x & SD_LA) != 0;
      const int i = top ? l : l - n;
      const int nbv = tile[o + (top ? i - TP : i * TP - 1)];
      const int sum = row_sum16((l < 2 * n && (top ? ta : la)) ? nbv : 0);
      dcv = (ta && la) ? (sum + n) >> (lgn + 1) : (ta || la) ? (sum + (n >> 1)) >> lgn : 0x80;
    }
    const bool wr = (d.x & (SD_KTAP | SD_KDC | SD_CODED)) != 0; // else: what a plane pass left there stays
    const bool w8 = wr && !is4, w4 = wr && is4;
    if (__builtin_amdgcn_ballot_w64(w8) != 0) {
      if (w8) {
        uint8_t *px = tile + o + po8;
        uint32_t word = *(const uint32_t *)px;
        if (d.x & SD_KTAP) { // the directional predictors: four neighbour samples per predicted sample, named by the tap table
          const uint32_t tb = tile_lds + (uint32_t)o;
          const uint32_t t0 = tap4(tb, ea.x, ea.y), t1 = tap4(tb, ea.z, ea.w), t2 = tap4(tb, eb.x, eb.y), t3 = tap4(tb, eb.z, eb.w);
          word = t0 | (t1 << 8) | (t2 << 16) | (t3 << 24);
        } else if (d.x & SD_KDC) {
          word = (uint32_t)dcv * 0x01010101u;
        }
        if (d.x & SD_CODED) { // two samples per instruction, as in the inter kernel (idct_pass2_q)
          const uint2 rr = *(const uint2 *)(res16 + (d.y & 0x1FF) + ro8);
          union { s16x2 v; uint32_t u; } p01, p23, r01, r23;
          p01.u = __builtin_amdgcn_perm(0u, word, 0x0c010c00u);
          p23.u = __builtin_amdgcn_perm(0u, word, 0x0c030c02u);
          r01.u = rr.x;
          r23.u = rr.y;
          const s16x2 s01 = __builtin_elementwise_add_sat(p01.v, r01.v), s23 = __builtin_elementwise_add_sat(p23.v, r23.v);
          flo = __builtin_elementwise_min(flo, __builtin_elementwise_min(s01, s23));
          fhi = __builtin_elementwise_max(fhi, __builtin_elementwise_max(s01, s23));
          word = __builtin_amdgcn_perm(sat_pk_u8(s23), sat_pk_u8(s01), 0x05040100u);
        }
        *(uint32_t *)px = word;
      }
    }
    if (__builtin_amdgcn_ballot_w64(w4) != 0) {
      if (w4) {
        uint8_t *px = tile + o + po4;
        int p = *px;
        if (d.x & SD_KTAP) p = (int)tap4(tile_lds + (uint32_t)o, ea.x, ea.y);
        else if (d.x & SD_KDC) p = dcv;
        if (d.x & SD_CODED) p = mobi_add_clamp(p, (int)res16[(d.y & 0x1FF) + ro4], &fault);
        *px = (uint8_t)p;
      }
    }
    wave_sync();
  };
  if (dbg & 2) n_iter = 0;
  MOBI_ISTOP(6);
  uint2 dA, dB = uint2{0, 0};
  uint4 eaA, ebA, eaB = uint4{0, 0, 0, 0}, ebB = uint4{0, 0, 0, 0};
  load_step(0, dA, eaA, ebA);
#pragma unroll 1
  for (int t = 0; t < n_iter; t += 2) {
    if (t + 1 < n_iter) load_step(t + 1, dB, eaB, ebB);
    do_step(dA, eaA, ebA);
    if (t + 1 >= n_iter) break;
    if (t + 2 < n_iter) load_step(t + 2, dA, eaA, ebA);
    do_step(dB, eaB, ebB);
  }
  if (flo.x < -64 || flo.y < -64 || fhi.x > 319 || fhi.y > 319) fault = 1;
  MOBI_ISTOP(7);
  if (fault && I.valid && !dbg) atomicOr(&A.fault[clip], 1);

  // ---- store.  Write-through (sc1), drained and followed by the tag only when an intra macroblock of this step may be waiting for
  // these pixels on another CU; plain stores otherwise (the next launch is a kernel boundary away). ----
  const bool anyp = __builtin_amdgcn_ballot_w64(I.valid && I.publish) != 0;
  if (I.valid && !(dbg & 8)) {
    // the macroblock's tiles are contiguous: lane l stores chunk l of the luma tile (quadrant l >> 2, rows 2 * (l & 3) and the next one)
    // and, lanes 0..7, row l of the chroma tile ([U | V]): three whole 128-byte lines per macroblock
    const int quad = l >> 2, R0 = (quad >> 1) * 8 + 2 * (l & 3), c0 = (quad & 1) * 8;
    const uint8_t *src = tile + (R0 + 1) * TP + 4 + c0;
    const u32x4 v = {*(const uint32_t *)src, *(const uint32_t *)(src + 4), *(const uint32_t *)(src + TP), *(const uint32_t *)(src + TP + 4)};
    uint8_t *dst = ty + l * 16;
    const uint8_t *su = tile + IQ_TCU + ((l & 7) + 1) * TP + 4, *sv = tile + IQ_TCV + ((l & 7) + 1) * TP + 4;
    const u32x4 vc = {*(const uint32_t *)su, *(const uint32_t *)(su + 4), *(const uint32_t *)sv, *(const uint32_t *)(sv + 4)};
    uint8_t *dstc = tc + (l & 7) * 16;
#if defined(__HIP_DEVICE_COMPILE__)
    if (anyp) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
    if (l < 8) {
      if (anyp) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dstc), "v"(vc) : "memory");
      else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(dstc), "v"(vc) : "memory");
    }
#endif
  }
  if (anyp) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the pixels have left this CU before the tag does
    if (I.valid && I.publish && l == 0) __hip_atomic_store(A.done + (size_t)clip * A.n_mbs + mb, A.step_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Launch list built on the host (mobi_abi.cpp, LevelPlan): 16 bytes per intra macroblock, sorted by dependency level, every level
// padded to a whole number of waves with null items (x = ~0).
//   x = clip << 13 | mb   y = MbDesc.w1   z = MbDesc.payload_off (inside this step's arena)
//   w = [0] 16x16 plane present  [1] has intra dependencies  [2] has intra dependents  [3] unused (r02: the left neighbour's last
//       column is in the edge side buffer)  [14:5] number of level words  [31:16] plane parameter
extern "C" __global__ __launch_bounds__(64) void mobi_recon_intra(MobiReconArgs A, const uint4 *items, int n_items, int dbg) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[4 * IQ_WORDS];
  const int lane = threadIdx.x;
  const uint4 item = items[blockIdx.x * 4 + (lane >> 4)];
  const bool valid = item.x != 0xFFFFFFFFu;
  const QItem I{valid, valid ? item.x >> 13 : 0u, valid ? item.x & 0x1FFFu : 0u, item.y, item.z, item.w & (0xFFFF0001u | MOBI_W3_WIDE), (item.w >> 5) & 0x3FFu,
                (item.w & 2) != 0, (item.w & 4) != 0};
  recon_intra_quad(A, lds, I, lane, dbg);
}

// Small batches: the whole frame step in ONE launch.  Workgroups [0, n_inter) are octets of inter macroblocks, the rest fours of intra
// items as above.
// ASSUMPTION: workgroups start in blockIdx order per XCD, so every octet workgroup holds a wave slot before an intra four that
// waits for it can fill the chip.  HIP does not promise that order; the hardware dispatcher has always kept it.  Should a driver change it,
// the waits do not hang: a four gives up after 2^21 polls and sets fault bit 2 -- and mobi_batch_decode then runs the SAME step again as
// two launches (mobi_abi.cpp, launch_plan's retry), which need no order at all.  The padding tiles behind a picture's last macroblock are
// not rewritten here (the two-launch octet kernel stores zeros there): they hold the zeros mobi_batch_create cleared the arena with.  Two launches cost a kernel boundary (the second waits until the first has drained, then starts cold) -- at 8 clips per
// GPU (BASELINE config 4) that boundary is a fifth of the step; here an intra macroblock starts as soon as the inter macroblocks its halo
// reads carry the step's tag.  Inter workgroups never wait and are dispatched first, so the waits cannot deadlock.
extern "C" __global__ __launch_bounds__(64, 4) void mobi_recon_step(MobiReconArgs A, const uint4 *items, uint32_t n_inter) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[P_BYTES];
  static_assert(P_BYTES >= 4 * IQ_WORDS * 4, "the intra fours fit the octet's LDS");
  if (blockIdx.x < n_inter) {
    const uint32_t oi = (blockIdx.x & 7) * A.inter_per_xcd + (blockIdx.x >> 3);
    if (oi >= A.qpc * (uint32_t)A.n_clips) return;
    uint32_t rem, ox;
    const uint32_t clip = fastdiv(oi, A.qpc, A.magic_qpc, rem);
    const uint32_t mby = fastdiv(rem, A.qpr, A.magic_qpr, ox);
    recon_inter_oct<0, 16, 1>(A, lds, clip, mby, ox, (int)threadIdx.x);
    return;
  }
  const int lane = threadIdx.x;
  const uint4 item = items[(blockIdx.x - n_inter) * 4 + (lane >> 4)];
  const bool valid = item.x != 0xFFFFFFFFu;
  const QItem I{valid, valid ? item.x >> 13 : 0u, valid ? item.x & 0x1FFFu : 0u, item.y, item.z, item.w & (0xFFFF0001u | MOBI_W3_WIDE), (item.w >> 5) & 0x3FFu,
                (item.w & 2) != 0, (item.w & 4) != 0};
  recon_intra_quad<1>(A, (uint32_t *)lds, I, lane);
}

// Items as the device-side parser leaves them (mobi_dparse.hip): per clip, raster order, n_intra[clip] of them at a stride of
// n_mbs.  Workgroup = slot * ceil(n_clips / 4) + clip quad: the four rows of a wave are the same slot of four clips, neighbours in
// the dispatch order belong to different clips, so every clip advances along its own dependency chain at the same time, and what
// a row waits for (raster-earlier, same clip) always sits in an earlier slot, i.e. was dispatched before it.  Who has to poll and
// who has to publish is in the descriptor (w3 bits 1 and 2: the parsers set them where they list the dependencies).
extern "C" __global__ __launch_bounds__(64) void mobi_recon_intra_cl(MobiReconArgs A, const uint32_t *items, const uint32_t *n_intra, uint32_t n_intra_stride,
                                                                      uint32_t quads, uint32_t magic_quads) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[4 * IQ_WORDS];
  const int lane = threadIdx.x;
  uint32_t cq;
  const uint32_t slot = fastdiv(blockIdx.x, quads, magic_quads, cq);
  const uint32_t clip = 4 * cq + (uint32_t)(lane >> 4);
  const bool inb = clip < (uint32_t)A.n_clips;
  const uint32_t ni = inb ? n_intra[(size_t)clip * n_intra_stride] : 0u;
  const bool valid = slot < ni;
  if (__builtin_amdgcn_ballot_w64(valid) == 0) return;
  const uint32_t mb = valid ? items[(size_t)clip * A.n_mbs + slot] & 0x1FFFu : 0u;
  const MbDesc *desc = A.desc + (size_t)(valid ? clip : 0) * A.n_mbs + mb;
  const uint32_t w3 = valid ? desc->w3 : 0u; // [1] has intra dependencies: poll their tags; [2] has intra dependents: publish its own
  const QItem I{valid, valid ? clip : 0u, mb, valid ? desc->w1 : 0u, valid ? desc->payload_off : 0u, w3 & (0xFFFF0001u | MOBI_W3_WIDE),
                valid ? desc->w2 & 0x3FFu : 0u, (w3 & 2u) != 0, (w3 & 4u) != 0};
  recon_intra_quad(A, lds, I, lane);
}
// When the host does not know the longest list (a step submitted before its parse has run: mobi_batch_submit launches
// MOBI_ASYNC_INTRA_SLOTS slots, not one per macroblock of the picture), this launch follows: one workgroup per clip quad walks through
// whatever its four clips have beyond the slots already launched.  Those macroblocks depend on raster-earlier ones of the same clip
// only -- the launch before, or this workgroup's own earlier rounds -- so the waits cannot deadlock; they run one after the other,
// which a raster chain that long (an I-frame) mostly does anyway.  (r03 had the walk as a loop inside mobi_recon_intra_cl: with everything
// lane-derived kept alive round it the kernel took 185 registers -- two waves per SIMD -- for every workgroup, walking or not.)
extern "C" __global__ __launch_bounds__(64) void mobi_recon_intra_walk(MobiReconArgs A, const uint32_t *items, const uint32_t *n_intra, uint32_t n_intra_stride,
                                                                        uint32_t first_slot) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[4 * IQ_WORDS];
  for (uint32_t slot = first_slot;; slot++) {
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane)); // (opaque: nothing derived from the lane number is carried from round to round)
    const uint32_t clip = 4 * blockIdx.x + (uint32_t)(lane >> 4);
    const bool inb = clip < (uint32_t)A.n_clips;
    const uint32_t ni = inb ? n_intra[(size_t)clip * n_intra_stride] : 0u;
    const bool valid = slot < ni;
    if (__builtin_amdgcn_ballot_w64(valid) == 0) return;
    const uint32_t mb = valid ? items[(size_t)clip * A.n_mbs + slot] & 0x1FFFu : 0u;
    const MbDesc *desc = A.desc + (size_t)(valid ? clip : 0) * A.n_mbs + mb;
    const uint32_t w3 = valid ? desc->w3 : 0u;
    const QItem I{valid, valid ? clip : 0u, mb, valid ? desc->w1 : 0u, valid ? desc->payload_off : 0u, w3 & (0xFFFF0001u | MOBI_W3_WIDE),
                  valid ? desc->w2 & 0x3FFu : 0u, (w3 & 2u) != 0, (w3 & 4u) != 0};
    recon_intra_quad(A, lds, I, lane);
    wave_sync();
  }
}

// =====================================================================================================
// launch wrappers (called from mobi_abi.cpp)
// =====================================================================================================
// Untile one frame: lin = the reference's Y[stride * height] followed by UV[stride * height / 2] (MD.cs:107-108, 414-415).  One lane = 8
// bytes of a plane row (an 8-aligned run never leaves a quadrant row / one plane's half of a chroma tile row).
extern "C" __global__ __launch_bounds__(256) void mobi_untile(const uint8_t *slot, uint8_t *lin, uint32_t ysz, int lgS) {
  const uint32_t a = (blockIdx.x * 256u + threadIdx.x) * 8u; // linear offset inside Y (a < ysz) or ysz + offset inside UV
  if (a >= ysz + (ysz >> 1)) return;
  const uint32_t t = a < ysz ? mobi_ty(a, lgS) : ysz + mobi_tc(a - ysz, lgS);
  *(uint2 *)(lin + a) = *(const uint2 *)(slot + t);
}
extern "C" int mobi_launch_untile(const uint8_t *slot, uint8_t *lin_dev, int stride, int height, hipStream_t s) {
  const uint32_t ysz = (uint32_t)stride * (uint32_t)height;
  hipLaunchKernelGGL(mobi_untile, dim3((ysz + (ysz >> 1)) / 8 / 256 + 1), dim3(256), 0, s, slot, lin_dev, ysz, 31 - __builtin_clz((unsigned)stride));
  return (int)hipGetLastError();
}
// Profiling switches (extra LDS per workgroup to lower occupancy; ablations of the intra kernel that produce WRONG pictures on purpose)
// exist only in builds with -DMOBI_PROFILING (tools/): a drop-in decoder must not be one environment variable away from them.
#if defined(MOBI_PROFILING)
static int prof_env(const char *name) { const char *v = getenv(name); return v ? atoi(v) : 0; }
#else
static int prof_env(const char *) { return 0; }
#endif
extern "C" int mobi_launch_inter(const MobiReconArgs *a, hipStream_t s) {
  if (a->n_clips <= 0) return 0;
  if (a->slot_bytes >= (1u << 24)) return (int)hipErrorInvalidValue; // 24-bit multiply in the kernel
  MobiReconArgs b = *a;
  static const int lds_pad = prof_env("MOBI_LDS_PAD");
  static const int stop_stage = prof_env("MOBI_STOP_STAGE"); // (--profiling builds only: tools/exp_stages.sh)
  b.reserved21 = (uint32_t)stop_stage;
  b.qpr = ((uint32_t)b.mbw + 7) / 8;                    // octets per macroblock row
  b.qpc = b.qpr * (uint32_t)(b.n_mbs / b.mbw);          // ... per clip
  auto magic = [](uint32_t d) { uint64_t m = ((uint64_t)1 << 32) / d; return (uint32_t)(m > 0xFFFFFFFFull ? 0xFFFFFFFFull : m); };
  b.magic_qpr = magic(b.qpr);
  b.magic_qpc = magic(b.qpc);
  const unsigned g8 = (unsigned)(((long)b.qpc * b.n_clips + 7) / 8 * 8); // whole number of workgroups per XCD
  b.inter_per_xcd = g8 / 8;
#if defined(MOBI_PROFILING)
  if (b.prof) hipLaunchKernelGGL(mobi_recon_inter8_prof, dim3(g8), dim3(64), lds_pad, s, b);
  else
#endif
  hipLaunchKernelGGL(mobi_recon_inter8, dim3(g8), dim3(64), lds_pad, s, b);
  return (int)hipGetLastError();
}
extern "C" int mobi_launch_intra(const MobiReconArgs *a, const uint32_t *items_dev, int n_items, hipStream_t s) {
  if (n_items <= 0) return 0;
  if (n_items & 3) return (int)hipErrorInvalidValue; // levels are padded to whole waves of four macroblocks
  static const int dbg = prof_env("MOBI_INTRA_DBG"); // timing ablations only: 1 no transform, 2 no steps, 4 no scatter, 8 no stores, 16 no halo, 32 no levels
  static const int pad = prof_env("MOBI_INTRA_LDS_PAD");
  hipLaunchKernelGGL(mobi_recon_intra, dim3((unsigned)n_items / 4), dim3(64), pad, s, *a, (const uint4 *)items_dev, n_items, dbg);
  return (int)hipGetLastError();
}
extern "C" int mobi_launch_step(const MobiReconArgs *a, const uint32_t *items_dev, int n_items, hipStream_t s) {
  if (a->n_clips <= 0 || n_items <= 0 || (n_items & 3)) return (int)hipErrorInvalidValue;
  if (a->slot_bytes >= (1u << 24)) return (int)hipErrorInvalidValue;
  MobiReconArgs b = *a;
  b.reserved21 = 0;
  b.qpr = ((uint32_t)b.mbw + 7) / 8;
  b.qpc = b.qpr * (uint32_t)(b.n_mbs / b.mbw);
  auto magic = [](uint32_t d) { uint64_t m = ((uint64_t)1 << 32) / d; return (uint32_t)(m > 0xFFFFFFFFull ? 0xFFFFFFFFull : m); };
  b.magic_qpr = magic(b.qpr);
  b.magic_qpc = magic(b.qpc);
  const unsigned g8 = (unsigned)(((long)b.qpc * b.n_clips + 7) / 8 * 8);
  b.inter_per_xcd = g8 / 8;
  hipLaunchKernelGGL(mobi_recon_step, dim3(g8 + (unsigned)n_items / 4), dim3(64), 0, s, b, (const uint4 *)items_dev, (uint32_t)g8);
  return (int)hipGetLastError();
}
// K slots are launched; walk != 0: the lists may be longer than K (the host has not seen the counts): mobi_recon_intra_walk finishes them
extern "C" int mobi_launch_intra_cl(const MobiReconArgs *a, const uint32_t *items_dev, const uint32_t *n_intra_dev, int n_intra_stride_words, int K, int walk, hipStream_t s) {
  if (K <= 0 || a->n_clips <= 0) return 0;
  const uint32_t quads = ((uint32_t)a->n_clips + 3) / 4;
  const uint64_t m = ((uint64_t)1 << 32) / quads;
  hipLaunchKernelGGL(mobi_recon_intra_cl, dim3((unsigned)K * quads), dim3(64), 0, s, *a, items_dev, n_intra_dev, (uint32_t)n_intra_stride_words, quads,
                     (uint32_t)(m > 0xFFFFFFFFull ? 0xFFFFFFFFull : m));
  if (walk && K < a->n_mbs)
    hipLaunchKernelGGL(mobi_recon_intra_walk, dim3(quads), dim3(64), 0, s, *a, items_dev, n_intra_dev, (uint32_t)n_intra_stride_words, (uint32_t)K);
  return (int)hipGetLastError();
}
