// mobi_kernels.hip -- gfx950 (MI355X / CDNA4) reconstruction kernels for the command lists of mobi_cmd.h.
//
//   mobi_recon_inter : the inter macroblocks of a frame step, one wavefront per quad of four adjacent macroblocks:
//                      half-pel truncating motion compensation from the reference planes (CopyBlock, MD.cs:418-456),
//                      dequant + 8x8/4x4 integer inverse transforms + clamp-add of the residual
//                      (MD.cs:3424-3429, :3435-3798), whole-row stores of Y/U/V.
//   mobi_recon_intra : intra macroblocks (I-frames and codes 6/7 inside P-frames), one wavefront each: halo load
//                      with raster-order availability masking, predictors (MD.cs:1883-2774, :3017-3327) and
//                      residuals in decode order inside LDS; all dependency levels of a step in one launch, ordered by
//                      per-macroblock completion tags.
//   mobi_recon_step  : both of the above as ONE launch (alternative step mode).
//
// 8-bit pel work is HBM-bound by nature: no MFMA.  LDS use is per wave (no workgroup barriers): waves never share
// LDS data, so a wavefront-scope fence (a pure compiler barrier -- LDS executes a wave's instructions in order) is
// all that separates producer and consumer lanes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "mobi_cmd.h"
#include "mobi_kernels.h"
#include "mobi_recon_math.h"

namespace {

enum { TP = 32 };                     // intra tile pitch: interior col c at byte 4+c, halo col -1 at byte 3
enum { HALO_Y_RIGHT = 23, HALO_C_RIGHT = 15 }; // must match MOBI_HALO_* in mobi_parse.h
enum { INTER_WAVES = 1, IWAVES = 1 }; // waves per workgroup: one (a workgroup's LDS and wave slots are released only when its last wave
                                      // ends, and quads / intra macroblocks differ widely in how long they take)

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

// ---- reference fetch: a lane needs 5 consecutive bytes (4 pixels + the half-pel neighbour) of a row, and
// the same of the row below, at an arbitrary byte offset `o` of a 4-byte-aligned plane.  They are fetched as
// aligned dword pairs (global_load_dwordx2, never flat) and cut out with v_alignbyte.  All loads of a
// macroblock are issued before the first one is consumed: no control flow sits between them.
typedef uint2 __attribute__((aligned(4))) uint2_a4;
typedef uint4 __attribute__((aligned(4))) uint4_a4;
struct Win { uint2 r0, r1; uint32_t sh; }; // row, row below, byte shift 0..3
__device__ __forceinline__ Win fetch_win(const uint32_t *plane32, int o, int S) {
  Win w;
  const uint32_t *q = plane32 + (o >> 2);
  w.r0 = *(const uint2_a4 *)q;
  w.r1 = *(const uint2_a4 *)(q + (S >> 2));
  w.sh = (uint32_t)o & 3;
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
// ---- residual helpers (LDS: coef[6*64] ints, tmp[6*64] ints) --------------------------------------
__device__ __forceinline__ void zero_coefs(int *coef, int lane) {
#pragma unroll
  for (int i = 0; i < 6; i++) coef[lane + 64 * i] = 0;
}
__device__ __forceinline__ void scatter_one(const int32_t *sc, uint32_t e, uint32_t t8, int *coef) {
  const int t = e & 0x1FF, level = (int32_t)e >> 16, area = t >> 6, p = t & 63;
  const int scale = ((t8 >> area) & 1) ? sc[p] : sc[64 + (p & 15)];
  coef[t] = scale * level;
}
__device__ __forceinline__ void scatter_coefs(const int32_t *sc, const uint32_t *cw, int first, int n, uint32_t t8, int *coef, int lane) {
  for (int i = first + lane; i < n; i += 64) scatter_one(sc, cw[i], t8, coef);
}
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
// pass 2 of area b by lane r: adds the residual into the 8x8 pixel area at `px` (pitch in bytes)
__device__ __forceinline__ void idct_pass2(const int *t, bool is8, int r, uint8_t *px, int pitch, int sub_mask, int *fault) {
  int in[8], out[8];
  if (is8) {
#pragma unroll
    for (int m = 0; m < 8; m++) in[m] = t[8 * r + m];
    mobi_bfly8(in, out);
    uint8_t *row = px + r * pitch;
#pragma unroll
    for (int j = 0; j < 8; j++) row[j] = (uint8_t)mobi_add_clamp(row[j], out[j] >> 6, fault);
  } else {
    const int s = r >> 1;
    if (!((sub_mask >> s) & 1)) return;
#pragma unroll
    for (int g = 0; g < 2; g++) {
      const int i = (r & 1) * 2 + g;
#pragma unroll
      for (int m = 0; m < 4; m++) in[m] = t[16 * s + 4 * i + m];
      mobi_bfly4(in, out);
      uint8_t *row = px + ((s >> 1) * 4 + i) * pitch + (s & 1) * 4;
#pragma unroll
      for (int j = 0; j < 4; j++) row[j] = (uint8_t)mobi_add_clamp(row[j], out[j] >> 6, fault);
    }
  }
}

// pass 2 that keeps the residual instead of adding it: res[y * 8 + x] for the 8x8 pixel area (all four 4x4 blocks of a split area)
__device__ __forceinline__ void idct_pass2_res(const int *t, bool is8, int r, int *res) {
  int in[8], out[8];
  if (is8) {
#pragma unroll
    for (int m = 0; m < 8; m++) in[m] = t[8 * r + m];
    mobi_bfly8(in, out);
#pragma unroll
    for (int j = 0; j < 8; j++) res[8 * r + j] = out[j] >> 6;
  } else {
    const int s = r >> 1;
#pragma unroll
    for (int g = 0; g < 2; g++) {
      const int i = (r & 1) * 2 + g;
#pragma unroll
      for (int m = 0; m < 4; m++) in[m] = t[16 * s + 4 * i + m];
      mobi_bfly4(in, out);
      int *row = res + ((s >> 1) * 4 + i) * 8 + (s & 1) * 4;
#pragma unroll
      for (int j = 0; j < 4; j++) row[j] = out[j] >> 6;
    }
  }
}

} // namespace

// =====================================================================================================
// inter macroblocks
// =====================================================================================================
// q = x / d, r = x % d with magic = floor(2^32 / d): the estimate is at most one short
__device__ __forceinline__ uint32_t fastdiv(uint32_t x, uint32_t d, uint32_t magic, uint32_t &r) {
  uint32_t q = __umulhi(x, magic);
  r = x - q * d;
  if (r >= d) { q++; r -= d; }
  return q;
}

// ---- mobi_recon_inter: one wavefront per QUAD of four horizontally adjacent macroblocks (64 x 16 luma) ----
// Measured on MI355X (tools/ubench/valu.hip, salu.hip, MOBI_DEBUG=9 cycle records): a wave64 integer VALU op
// occupies a SIMD for ~3-4 cycles, a SCALAR op for ~4.3 (one scalar issue slot per SIMD visit) and the two
// overlap only partly; a memory round trip is ~3.4k cycles under load.  The previous version of this kernel
// ran ~700 VALU + ~650 SALU instructions per quad and was bound by instruction issue, most of the scalar ones
// being per-macroblock decode (bit fields, address arithmetic, exec masks, branches) repeated four times.
// Hence the shape of this one: NOTHING is done per macroblock in scalar code.
//   * lane = (g, j): g = lane >> 4 is the macroblock of the quad the lane works for, in every stage.
//     Each lane loads ITS macroblock's descriptor and decodes it with vector ops: one instruction stream
//     serves the four macroblocks at once.
//   * all global reads of the quad -- MC windows as whole 16-byte chunks, MV cell maps, residual levels,
//     dequant scales -- are six full-wave asynchronous global->LDS copies (global_load_lds_dwordx4: "LDS
//     staging of the macroblock + MC halo"), waited for once.  Lanes with nothing to fetch re-read a line
//     some other lane already touches instead of being masked off (exec-mask juggling is scalar work).
//   * motion compensation of all single-leaf macroblocks: 4 + 2 vector iterations (4 luma rows x 4 MBs,
//     one chroma plane x 4 MBs), the half-pel phase a per-lane select; multi-leaf macroblocks (MV cell map)
//     are redone one at a time by the whole wave;
//   * ONE batched inverse transform serves the coded areas of all four macroblocks, 8 areas x 8 rows = 64
//     lanes per pass; the residual levels of the four macroblocks are scattered together (16 lanes each);
//   * the quad leaves as whole 64-byte luma rows / 8-byte chroma rows.
// A wave lives for one quad (no loop-carried state, no workgroup barriers).  XCD-aware order: the dispatcher
// deals consecutive workgroups round-robin to the 8 XCDs, so XCD x gets one contiguous run of quads.
namespace {
enum {                           // per-wave LDS map.  A DMA round r puts lane i's 16 bytes at R_r + 16 * i, i = g*16 + j
  Q_R0 = 0,                      // luma window rows 0..7: j = row*2 + 16-byte half, so a row's 32 bytes are contiguous
  Q_R1 = 1024,                   // luma window rows 8..15
  Q_R2 = 2048,                   // U window rows 0..7
  Q_R3 = 3072,                   // V window rows 0..7
  Q_R4 = 4096,                   // j 0,1: luma row 16; 2,3: U row 8; 4,5: V row 8; j 6..15 of g 0,1: dequant scales (20 chunks)
  // after motion compensation the windows are dead and the same bytes are reused:
  Q_OUT_Y = 0,                   // out tile: luma 16 rows x 64 B
  Q_OUT_C = 1024,                //           chroma 2 planes x 8 rows x 32 B
  Q_COEF = 1536,                 // coefficient tile, 8 areas x 64 ints; transposed in place between the two passes
  Q_TAB = Q_R4,                  // coded-area table: entry -> g*8 + area (row 16 of MB 0)
  Q_META = 5120,                 // per quad: cbp6[4], t8mask[4], flags[4]
  Q_BYTES = 5136                 // 5.5 KB allocated: 29 waves per CU fit the 160 KB
};
typedef const void __attribute__((address_space(1))) *gptr_t;
typedef void __attribute__((address_space(3))) *lptr_t;
// lane i lands at dst + IMM + i*16; dst must be wave-uniform (it travels in M0); IMM = constant byte offset added to BOTH
// the source address and the LDS address
#define MOBI_DMA16(src, dst, IMM) __builtin_amdgcn_global_load_lds((gptr_t)(src), (lptr_t)(dst), 16, IMM, 0)

__device__ __forceinline__ uint32_t lds32(const uint8_t *L, int byte_off) { return *(const uint32_t *)(L + byte_off); }
// CopyBlock on four packed pixels (MD.cs:424-452): x0,x1 = aligned dwords holding the row, y0,y1 the row below,
// sh = byte shift 0..3, sh8 = 8*sh; the phase arrives as three lane masks.  (Unaligned LDS dword reads would make the
// byte-align arithmetic unnecessary -- they work on gfx950, but at a quarter of the aligned rate: 0.39 vs 0.25 ms.)
__device__ __forceinline__ uint32_t mc4_lane(uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1, uint32_t sh, uint32_t sh8,
                                             bool ph0, bool ph1, bool ph2) {
  const uint32_t M = 0x7F7F7F7Fu;
  const uint32_t a = __builtin_amdgcn_alignbyte(x1, x0, sh), c = __builtin_amdgcn_alignbyte(y1, y0, sh);
  const uint32_t b = __builtin_amdgcn_alignbyte(x1 >> sh8, a, 1), d = __builtin_amdgcn_alignbyte(y1 >> sh8, c, 1);
  const uint32_t ha = (a >> 1) & M, hb = (b >> 1) & M, hc = (c >> 1) & M, hd = (d >> 1) & M;
  const uint32_t p1 = ha + hb, p2 = ha + hc, p3 = ((p1 >> 1) & M) + (((hc + hd) >> 1) & M);
  return ph0 ? a : ph1 ? p1 : ph2 ? p2 : p3;
}
// pass 2 of one area by lane r, tracking the range of pred+residual instead of testing every sample.  Only the
// butterflies differ between one 8x8 transform (lane r = pixel row r) and four 4x4s (lane r = rows (r&1)*2, +1 of
// sub-block r>>1): both leave 8 residuals for two 4-pixel words, so the pixel update is one shared instruction stream
// (the 8 lanes of an area agree on the kind, the lanes of a wave do not).
__device__ __forceinline__ void idct_pass2_q(const int *t, bool is8, int r, uint8_t *px, int pitch, int &lo, int &hi) {
  int in[8], out[8];
  uint8_t *wa, *wb; // the two 4-pixel words (4-byte aligned)
  if (is8) {
#pragma unroll
    for (int m = 0; m < 8; m++) in[m] = t[8 * r + m];
    mobi_bfly8(in, out);
    wa = px + r * pitch;
    wb = wa + 4;
  } else {
    const int s = r >> 1, i0 = (r & 1) * 2;
#pragma unroll
    for (int m = 0; m < 8; m++) in[m] = t[16 * s + 4 * i0 + m]; // groups i0 and i0 + 1
    mobi_bfly4(in, out);
    mobi_bfly4(in + 4, out + 4);
    wa = px + ((s >> 1) * 4 + i0) * pitch + (s & 1) * 4;
    wb = wa + pitch;
  }
  const uint32_t pa = *(const uint32_t *)wa, pb = *(const uint32_t *)wb;
  int pix[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int v = (int)(((j < 4 ? pa : pb) >> (8 * (j & 3))) & 0xFF) + (out[j] >> 6);
    lo = v < lo ? v : lo; hi = v > hi ? v : hi;
    pix[j] = v < 0 ? 0 : (v > 255 ? 255 : v);
  }
  // byte stores, spelled out: LDS has issue slots to spare and the VALU has not, but left to itself the compiler packs the
  // eight results into two words with a dozen VALU instructions (and a volatile store through a generic pointer becomes a
  // flat_store with system scope)
#if defined(__HIP_DEVICE_COMPILE__)
  typedef uint8_t __attribute__((address_space(3))) *lds_u8p;
  const uint32_t aa = (uint32_t)(uintptr_t)(lds_u8p)wa, ab = (uint32_t)(uintptr_t)(lds_u8p)wb;
  asm volatile("ds_write_b8 %0, %1\n\tds_write_b8 %0, %2 offset:1\n\tds_write_b8 %0, %3 offset:2\n\tds_write_b8 %0, %4 offset:3"
               : : "v"(aa), "v"(pix[0]), "v"(pix[1]), "v"(pix[2]), "v"(pix[3]) : "memory");
  asm volatile("ds_write_b8 %0, %1\n\tds_write_b8 %0, %2 offset:1\n\tds_write_b8 %0, %3 offset:2\n\tds_write_b8 %0, %4 offset:3"
               : : "v"(ab), "v"(pix[4]), "v"(pix[5]), "v"(pix[6]), "v"(pix[7]) : "memory");
#else
  for (int j = 0; j < 4; j++) { wa[j] = (uint8_t)pix[j]; wb[j] = (uint8_t)pix[4 + j]; }
#endif
}
} // namespace

template <bool PROF>
__device__ __forceinline__ void recon_inter_quad(const MobiReconArgs &A, uint8_t *L, uint32_t qi, int lane) {
  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
  if (PROF) t0 = __builtin_readcyclecounter();
  uint32_t rem, qx;
  const uint32_t clip = fastdiv(qi, A.qpc, A.magic_qpc, rem);
  const uint32_t mby = fastdiv(rem, A.qpr, A.magic_qpr, qx);
  const uint32_t mbx0 = qx * 4, mbw = (uint32_t)A.mbw;
  const int nmb = (int)(mbw - mbx0 < 4 ? mbw - mbx0 : 4);
  const int S = A.stride, lgS = 31 - __builtin_clz((unsigned)S);
  const uint32_t ysz = (uint32_t)S * (uint32_t)A.height, slot_w = A.slot_bytes >> 2, ysz_w = ysz >> 2;
  uint8_t *clip_base = A.planes + (size_t)clip * A.clip_bytes;
  const int off0 = (int)(mby * 16 * (uint32_t)S + mbx0 * 16);
  const int g = lane >> 4, j = lane & 15;

  // ---- stage A: decode this lane's macroblock, then every global read of the quad, asynchronously into LDS ----
  const uint4 *dp = (const uint4 *)(A.desc + (clip * (uint32_t)A.n_mbs + mby * mbw + mbx0) + g); // the table has slack past the last quad
  const uint4 d = dp[0], d2 = dp[1];
  const bool valid = g < nmb && (d.y & 1) == MOBI_MB_INTER;
  const int nl = (d.y >> 1) & 0x7F, kind2 = (d.y >> 26) & 3;
  const bool single = valid && nl == 1, dual = valid && kind2 != 0, multi = valid && nl > 1 && kind2 == 0;
  const uint32_t cbp6 = valid ? (d.y >> 8) & 0x3F : 0, ncoef = cbp6 ? d.z & 0x3FF : 0;
  L[Q_META + g] = (uint8_t)cbp6;
  L[Q_META + 4 + g] = (uint8_t)((d.y >> 14) & 0x3F);
  L[Q_META + 8 + g] = (uint8_t)((valid ? 1 : 0) | (multi ? 2 : 0));
  const bool any_dual = __builtin_amdgcn_ballot_w64(dual) != 0;
  // Leaf records (mobi_cmd.h): the host already turned motion vectors into source positions and CopyBlock phases.
  // What a lane needs: one record for its luma rows 0..7 (iterations t = 0,1), one for rows 8..15 (t = 2,3), one for
  // its chroma samples.  Single-leaf: all three are leaf A.  DUAL (two halves): top/bottom switches between t = 1
  // and 2, left/right by the lane's column.
  auto slot_off = [&](uint32_t ref) {
    int sl = A.ring_base - (int)ref;
    sl = sl < 0 ? sl + 6 : sl;
    return __umul24((uint32_t)sl, A.slot_bytes); // slot_bytes < 2^24: checked by mobi_launch_inter
  };
  const uint32_t refA = slot_off((d.z >> 10) & 7);
  uint32_t ref01 = refA, ref23 = refA, refC = refA;
  int ypos = (int)d.w, ypos23 = (int)d.w, cpos = (int)d2.x;
  int yph01 = (d.z >> 16) & 3, yph23 = yph01, cph = (d.z >> 18) & 3;
  if (any_dual) {
    const uint32_t refB = slot_off((d.z >> 13) & 7);
    const bool lr = kind2 == MOBI_DUAL_LR, tb = kind2 == MOBI_DUAL_TB;
    const bool b01 = dual && lr && (j & 2), b23 = dual && (tb || (lr && (j & 2))), bc = dual && (lr ? (j & 1) != 0 : j >= 8);
    const int yphB = (d.z >> 20) & 3, cphB = (d.z >> 22) & 3;
    ref01 = b01 ? refB : refA; ypos = b01 ? (int)d2.y : ypos; yph01 = b01 ? yphB : yph01;
    ref23 = b23 ? refB : refA; ypos23 = b23 ? (int)d2.y : ypos23; yph23 = b23 ? yphB : yph23;
    refC = bc ? refB : refA; cpos = bc ? (int)d2.z : cpos; cph = bc ? cphB : cph;
  }
  // a lane without a window keeps re-reading the start of its clip.  (Row offsets by shifts: the pitch is a power of
  // two, and a 32-bit integer multiply costs four VALU slots.)
  const int hS = single ? S >> 1 : 0;
  auto rowoff = [&](int rows) { return single ? (uint32_t)rows << lgS : 0u; };
  const uint32_t ywin = single ? ref01 + (uint32_t)(ypos & ~15) : 0u;
  const uint32_t cwin = single ? refC + ysz + (uint32_t)(cpos & ~15) : 0u;
  const uint8_t *lbase = clip_base;
  {
    const uint8_t *p0 = lbase + (ywin + rowoff(j >> 1) + (uint32_t)(j & 1) * 16u);
    MOBI_DMA16(p0, L + Q_R0, 0);
    MOBI_DMA16(p0 + rowoff(8), L + Q_R1, 0);
    const uint8_t *p2 = lbase + (cwin + rowoff(j >> 1) + (uint32_t)(j & 1) * 16u);
    MOBI_DMA16(p2, L + Q_R2, 0);
    MOBI_DMA16(p2 + hS, L + Q_R3, 0);
    // leftovers: window rows 16 (luma) / 8 (U, V); the spare lanes of MBs 0 and 1 bring the dequant scales of the clip's quantizer
    const int h = j >> 1;
    const uint32_t o4 = (h == 0 ? ywin + rowoff(16) : h == 1 ? cwin + rowoff(8) : cwin + hS + rowoff(8)) + (uint32_t)(j & 1) * 16u;
    const int quant = __builtin_amdgcn_readfirstlane((int)((d.y >> 20) & 63)); // per clip (MD.cs:113-143): MB 0's copy
    const uint8_t *p4 = lbase + (j < 6 ? o4 : 0u);
    if (j >= 6 && g < 2) p4 = (const uint8_t *)(A.scale + quant * MOBI_SCALE_STRIDE) + (g * 10 + j - 6) * 16;
    MOBI_DMA16(p4, L + Q_R4, 0);
  }
  // residual level words: lane (g, j) scatters words j, j+16, j+32, ... of macroblock g, so they go straight into its registers
  const uint32_t *cw = A.payload + d.x + (multi ? MOBI_MV_CELLS : 0);
  uint32_t cwr[4] = {0, 0, 0, 0};
  if ((uint32_t)j < ncoef) cwr[0] = cw[j];
  if (__builtin_amdgcn_ballot_w64(ncoef > 16) != 0) {
#pragma unroll
    for (int k = 1; k < 4; k++)
      if ((uint32_t)(16 * k + j) < ncoef) cwr[k] = cw[16 * k + j];
  }
  // DUAL macroblocks: 8-wide / 8-high halves do not fit the window layout; their lanes fetch their own 2 x 8 bytes
  // per iteration straight into registers, in flight together with the DMA rounds.
  uint2 fx[6], fy[6]; // only read by DUAL lanes
  if (dual) {
    const int rr = j >> 2, q = j & 3;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const uint32_t o = (t < 2 ? ref01 : ref23) + (uint32_t)(((t < 2 ? ypos : ypos23) + ((4 * t + rr) << lgS) + 4 * q) & ~3);
      fx[t] = *(const uint2_a4 *)(clip_base + o);
      fy[t] = *(const uint2_a4 *)(clip_base + o + S);
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const uint32_t o = refC + ysz + (uint32_t)((cpos + u * (S >> 1) + ((j >> 1) << lgS) + 4 * (j & 1)) & ~3);
      fx[4 + u] = *(const uint2_a4 *)(clip_base + o);
      fy[4 + u] = *(const uint2_a4 *)(clip_base + o + S);
    }
  }
  if (PROF) t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wave_sync();
  if (PROF) t2 = __builtin_readcyclecounter();
  const uint32_t m32 = __builtin_amdgcn_readfirstlane(lds32(L, Q_META));      // coded 8x8 areas: bit g*8 + area
  const uint32_t t32 = __builtin_amdgcn_readfirstlane(lds32(L, Q_META + 4));  // ... that use one 8x8 transform
  const uint32_t f32 = __builtin_amdgcn_readfirstlane(lds32(L, Q_META + 8));  // byte g: bit 0 inter, bit 1 multi-leaf (cell map)
  if ((f32 & 0x01010101u) == 0) return;

  // ---- stage B: motion compensation into the quad's out tile ----
  // single-leaf macroblocks out of the LDS windows, DUAL ones out of their registers, all at once (the other lanes
  // compute garbage into tiles nobody stores, or that B2 overwrites)
  uint32_t mcv[6]; // 4 luma + 2 chroma words of this lane: the out tile takes the place of the windows, so nothing is
                   // written before every window read has been issued (LDS executes a wave's instructions in order)
  auto stage_b = [&](auto with_dual) {
    constexpr bool DUAL = decltype(with_dual)::value;
    {
      // lane (g, rr = j>>2, q = j&3) takes row 4t + rr, pixels 4q..4q+3 (+1 for the half-pel neighbour): two aligned dwords of
      // the row and of the row below.  Row y of the window: (y>>3)*1024 + g*256 + (y&7)*32, row 16 in Q_R4; rows 7 -> 8
      // and 15 -> 16 are the only non-contiguous steps (lanes rr == 3 at t = 1, 3).
      const int rr = j >> 2, q = j & 3, ys = ypos & 15, wq = (ys + 4 * q) & ~3;
      const int A0 = Q_R0 + g * 256 + rr * 32 + wq;
      const int c1 = rr == 3 ? Q_R1 + g * 256 + wq : A0 + 128 + 32;        // row below at t = 1 (row 8 for rr == 3)
      const int c3 = rr == 3 ? Q_R4 + g * 256 + wq : A0 + 1024 + 128 + 32; // row below at t = 3 (row 16 for rr == 3)
      const uint32_t sh01 = ypos & 3, sh23 = ypos23 & 3;
      const bool p0a = yph01 == 0, p1a = yph01 == 1, p2a = yph01 == 2, p0b = yph23 == 0, p1b = yph23 == 1, p2b = yph23 == 2;
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int at = A0 + (t >> 1) * 1024 + (t & 1) * 128, ct = t == 1 ? c1 : t == 3 ? c3 : at + 32;
        uint32_t x0 = lds32(L, at), x1 = lds32(L, at + 4), y0 = lds32(L, ct), y1 = lds32(L, ct + 4);
        if (DUAL) { x0 = dual ? fx[t].x : x0; x1 = dual ? fx[t].y : x1; y0 = dual ? fy[t].x : y0; y1 = dual ? fy[t].y : y1; }
        const bool second = DUAL && t >= 2;
        const uint32_t sh = second ? sh23 : sh01;
        mcv[t] = mc4_lane(x0, x1, y0, y1, sh, sh * 8, second ? p0b : p0a, second ? p1b : p1a, second ? p2b : p2a);
      }
    }
    {
      // chroma: plane u, lane (g, row = j>>1, q = j&1); row r of the window: Q_R2 + u*1024 + g*256 + r*32, row 8 in Q_R4
      const int row = j >> 1, q = j & 1, cs = cpos & 15, wq = (cs + 4 * q) & ~3;
      const int A0 = Q_R2 + g * 256 + row * 32 + wq;
      const int C0 = row == 7 ? Q_R4 + g * 256 + 32 + wq : A0 + 32, cstep = row == 7 ? 32 : 1024; // U row 8 at +32, V row 8 at +64
      const uint32_t sh = cs & 3, sh8 = sh * 8;
      const bool ph0 = cph == 0, ph1 = cph == 1, ph2 = cph == 2;
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int at = A0 + 1024 * u, ct = C0 + u * cstep;
        uint32_t x0 = lds32(L, at), x1 = lds32(L, at + 4), y0 = lds32(L, ct), y1 = lds32(L, ct + 4);
        if (DUAL) { x0 = dual ? fx[4 + u].x : x0; x1 = dual ? fx[4 + u].y : x1; y0 = dual ? fy[4 + u].x : y0; y1 = dual ? fy[4 + u].y : y1; }
        mcv[4 + u] = mc4_lane(x0, x1, y0, y1, sh, sh8, ph0, ph1, ph2);
      }
    }
  };
  if (any_dual) stage_b(std::true_type{});
  else stage_b(std::false_type{});
  wave_sync();
  {
    const int oy = Q_OUT_Y + (j >> 2) * 64 + g * 16 + (j & 3) * 4, oc = Q_OUT_C + (j >> 1) * 32 + g * 8 + (j & 1) * 4;
#pragma unroll
    for (int t = 0; t < 4; t++) *(uint32_t *)(L + oy + 256 * t) = mcv[t];
#pragma unroll
    for (int u = 0; u < 2; u++) *(uint32_t *)(L + oc + 256 * u) = mcv[4 + u];
  }
  // B2: multi-leaf macroblocks (deeper partition trees; rare), one at a time by the whole wave.  Every lane looks up the
  // MV cells under its own pixels, then all its fetches fly together.
  {
    const int yrow = lane >> 2, yc4 = (lane & 3) * 4;
    const int cv = (lane >> 4) & 1, crow = (lane & 15) >> 1, cc4 = (lane & 1) * 4;
    uint32_t mm = (f32 >> 1) & 0x01010101u;
    while (mm) {
      const int gm = (__builtin_ctz(mm)) >> 3;
      mm &= mm - 1;
      const uint32_t *cells = A.payload + __builtin_amdgcn_readlane(d.x, gm * 16); // the MV cell map opens the payload
      const uint32_t *clip32 = (const uint32_t *)clip_base;
      const int offm = off0 + gm * 16;
      const int ybase = offm + (yrow << lgS) + yc4, cbase = (offm >> 1) + cv * (S >> 1) + (crow << lgS) + cc4;
      const uint2 yc = *(const uint2_a4 *)(cells + (yrow >> 1) * 8 + (yc4 >> 1));
      const uint4_a4 c4v = *(const uint4_a4 *)(cells + crow * 8 + cc4);
      const uint32_t cell[4] = {c4v.x, c4v.y, c4v.z, c4v.w};
      auto slot_of = [&](uint32_t c) { int s2 = A.ring_base - mobi_cell_ref(c); return __umul24((uint32_t)(s2 < 0 ? s2 + 6 : s2), slot_w); };
      // a lane's 4 luma pixels sit under two cells, its 4 chroma samples under four; for the common splits
      // (leaves at least 8 wide) they are the same cell: one window instead of two / four
      const bool ysplit = __builtin_amdgcn_ballot_w64(yc.x != yc.y) != 0;
      const bool csplit = __builtin_amdgcn_ballot_w64(lane < 32 && (cell[0] != cell[1] || cell[0] != cell[2] || cell[0] != cell[3])) != 0;
      const int dxa = mobi_cell_dx(yc.x), dya = mobi_cell_dy(yc.x), dxb = mobi_cell_dx(yc.y), dyb = mobi_cell_dy(yc.y);
      // every fetch of the macroblock is issued before the first one is used
      const Win wa = fetch_win(clip32 + slot_of(yc.x), ybase + ((dya >> 1) << lgS) + (dxa >> 1), S);
      Win wb; // only read when ysplit (copying wa here would wait for its loads)
      if (ysplit) wb = fetch_win(clip32 + slot_of(yc.y), ybase + ((dyb >> 1) << lgS) + (dxb >> 1), S);
      int qx[4], qy[4];
#pragma unroll
      for (int k = 0; k < 4; k++) { qx[k] = mobi_cell_dx(cell[k]) >> 1; qy[k] = mobi_cell_dy(cell[k]) >> 1; }
      Win wq[4];
      wq[0] = fetch_win(clip32 + slot_of(cell[0]) + ysz_w, cbase + ((qy[0] >> 1) << lgS) + (qx[0] >> 1), S);
      if (csplit) {
#pragma unroll
        for (int k = 1; k < 4; k++) wq[k] = fetch_win(clip32 + slot_of(cell[k]) + ysz_w, cbase + ((qy[k] >> 1) << lgS) + (qx[k] >> 1), S);
      }
      asm volatile("" ::: "memory"); // keep the loads above the arithmetic
      const uint32_t va = mc4_select(wa, (dxa & 1) | ((dya & 1) << 1));
      const uint32_t vb = ysplit ? mc4_select(wb, (dxb & 1) | ((dyb & 1) << 1)) : va;
      uint32_t cpred = mc4_select(wq[0], (qx[0] & 1) | ((qy[0] & 1) << 1));
      if (csplit) {
        cpred &= 0xFFu;
#pragma unroll
        for (int k = 1; k < 4; k++) cpred |= mc4_select(wq[k], (qx[k] & 1) | ((qy[k] & 1) << 1)) & (0xFFu << (8 * k));
      }
      *(uint32_t *)(L + Q_OUT_Y + yrow * 64 + gm * 16 + yc4) = (va & 0x0000FFFFu) | (vb & 0xFFFF0000u);
      if (lane < 32) *(uint32_t *)(L + Q_OUT_C + cv * 256 + crow * 32 + gm * 8 + cc4) = cpred;
    }
  }
  if (PROF) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); t3 = __builtin_readcyclecounter(); }

  // ---- stage C: one batched inverse transform for the coded areas of the whole quad ----
  wave_sync(); // windows are dead from here on: the coefficient tile takes their place
  if (m32) {
    const int n_ent = __builtin_popcount(m32);
    int *coef = (int *)(L + Q_COEF), *tmp = coef;
    if (lane < 32 && ((m32 >> lane) & 1)) L[Q_TAB + __builtin_popcount(m32 & ((1u << lane) - 1))] = (uint8_t)lane;
    int lo = 0, hi = 0;
    for (int pass = 0; pass * 8 < n_ent; pass++) {
      {
        const uint4 z = uint4{0, 0, 0, 0};
        *(uint4 *)(L + Q_COEF + lane * 16) = z;
        *(uint4 *)(L + Q_COEF + 1024 + lane * 16) = z;
      }
      wave_sync();
      // residual levels of the four macroblocks together: lane (g, j) takes levels j, j+16, ... of macroblock g
      auto scatter = [&](uint32_t e) {
        const int t = e & 0x1FF, level = (int32_t)e >> 16, k = g * 8 + (t >> 6), p = t & 63;
        const int slot = __builtin_popcount(m32 & ((1u << k) - 1)) - pass * 8;
        const int si = ((t32 >> k) & 1) ? p : 64 + (p & 15);                       // scale8[p] / scale4[p & 15]
        const int scale = (int)lds32(L, Q_R4 + 96 + si * 4 + (si >= 40 ? 96 : 0)); // two runs of 10 chunks, see Q_R4
        if ((unsigned)slot < 8u) coef[slot * 64 + p] = __mul24(scale, level); // scale < 2^24, level 16 bits: exact, and full rate
      };
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bool mine = (uint32_t)(16 * k + j) < ncoef;
        if (k && __builtin_amdgcn_ballot_w64(mine) == 0) break;
        uint32_t e = cwr[k];
        asm volatile("" : "+v"(e)); // keeps the decode of words 16.. behind the branch (the compiler hoists it out of the pass loop otherwise)
        if (mine) scatter(e);
      }
      for (uint32_t i = 64u + (uint32_t)j; __builtin_amdgcn_ballot_w64(i < ncoef) != 0; i += 16)
        if (i < ncoef) scatter(cw[i]);
      wave_sync();
      const int e = lane >> 3, r = lane & 7, idx = pass * 8 + e;
      const bool act = idx < n_ent;
      const int k = act ? L[Q_TAB + idx] : 0;
      const int ge = k >> 3, a = k & 7;
      const bool is8 = (t32 >> k) & 1;
      if (act) idct_pass1(coef + 64 * e, tmp + 64 * e, is8, r);
      wave_sync();
      if (act) {
        uint8_t *px = a < 4 ? L + Q_OUT_Y + (a >> 1) * 8 * 64 + ge * 16 + (a & 1) * 8 : L + Q_OUT_C + (a - 4) * 256 + ge * 8;
        idct_pass2_q(tmp + 64 * e, is8, r, px, a < 4 ? 64 : 32, lo, hi);
      }
      wave_sync();
    }
    if (lo < -64 || hi > 319) atomicOr(&A.fault[clip], 1); // clamp table domain (MobiConst.cs:587)
  }
  if (PROF) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t4 = __builtin_readcyclecounter(); }

  // ---- stage D: the quad leaves as whole rows: 64 B of luma, 8 B per macroblock of chroma ----
  // One-launch steps (A.done): intra macroblocks of the same launch read these pixels from other CUs, so the stores are
  // write-through (sc1), drained, and then the four macroblocks' completion tags are published (see mobi_recon_step).
  uint8_t *y0 = clip_base + (uint32_t)A.ring_base * A.slot_bytes;
  {
    const int gq = lane & 3, yrow = lane >> 2;
    if ((f32 >> (8 * gq)) & 1) {
      const int pl = lane >> 5, row = (lane >> 2) & 7;
      uint8_t *py = y0 + (off0 + (yrow << lgS) + gq * 16), *pc = y0 + ysz + ((off0 >> 1) + pl * (S >> 1) + (row << lgS) + gq * 8);
      const uint4 vy = *(const uint4 *)(L + Q_OUT_Y + yrow * 64 + gq * 16);
      const uint2 vc = *(const uint2 *)(L + Q_OUT_C + pl * 256 + row * 32 + gq * 8);
      if (A.done) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x4 ay = {vy.x, vy.y, vy.z, vy.w};
        const u32x2 ac = {vc.x, vc.y};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx2 %2, %3, off sc1\n\ts_nop 1" : : "v"(py), "v"(ay), "v"(pc), "v"(ac) : "memory");
#endif
      } else {
        *(uint4 *)py = vy;
        *(uint2 *)pc = vc;
      }
    }
  }
  if (A.done) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the pixels have left this CU before the tags do
    if (lane < 4 && ((f32 >> (8 * lane)) & 1))
      __hip_atomic_store(A.done + (size_t)clip * A.n_mbs + (mby * mbw + mbx0) + lane, A.step_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (PROF && lane == 0) // MOBI_DEBUG=9: where does a wave's life go (shader clock): issue, DMA wait, MC, IDCT
    ((uint4 *)A.prof)[qi] = uint4{(uint32_t)(t1 - t0), (uint32_t)(t2 - t1), (uint32_t)(t3 - t2) | ((uint32_t)__builtin_popcount(m32) << 24), (uint32_t)(t4 - t3)};
}

template <bool PROF>
__device__ __forceinline__ void recon_inter_entry(const MobiReconArgs &A) {
  __shared__ __attribute__((aligned(16))) uint8_t lds_all[INTER_WAVES][Q_BYTES];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t qi = ((blockIdx.x & 7) * A.inter_per_xcd + (blockIdx.x >> 3)) * INTER_WAVES + wave;
  if (qi >= A.qpc * (uint32_t)A.n_clips) return;
  recon_inter_quad<PROF>(A, lds_all[wave], qi, lane);
}
extern "C" __global__ __launch_bounds__(64 * INTER_WAVES) void mobi_recon_inter(MobiReconArgs A) { recon_inter_entry<false>(A); }
extern "C" __global__ __launch_bounds__(64 * INTER_WAVES) void mobi_recon_inter_prof(MobiReconArgs A) { recon_inter_entry<true>(A); }


// =====================================================================================================
// mobi_recon_inter8: the same work, one wavefront per OCTET of eight horizontally adjacent macroblocks
// =====================================================================================================
// The quad kernel is bound by instruction issue of every kind, and a good part of its instructions do not scale with
// the pixels: the per-wave prologue, and the decode / address stage in which every lane works for "its" macroblock.
// With 8 lanes per macroblock instead of 16, one pass through that code serves eight macroblocks.  Lane (g, j):
// g = lane >> 3 the macroblock, j = lane & 7.  DMA rounds hold 8 chunks per macroblock: 4 rows x 2 halves.
namespace {
enum {
  O_L = 0,       // 4 rounds: luma window rows 4t..4t+3  (slot ((row&3)*2 + half)*8 + g: 256 B per row, the rounds are contiguous)
  O_U = 4096,    // 2 rounds: U rows 0..3, 4..7
  O_V = 6144,    // 2 rounds: V rows 0..3, 4..7
  O_BYTES = 8192, // exactly 8 KB per wave: 20 waves per CU fit in the 160 KB of LDS (5 per SIMD)
  // once the chroma windows have been consumed (the V rows lie under them):
  O_TAB = 7168,    // entry -> area*8 + g (<= 48 bytes)
  O_SC = 7232,     // dequant scales (320 B), fetched while the luma is interpolated
  // after motion compensation:
  O_OUT_Y = 0,     // 16 rows x 128 B
  O_OUT_C = 2048,  // 2 planes x 8 rows x 64 B
  O_COEF = 3072    // 16 areas x 64 ints, transposed in place (up to 7168)
};
} // namespace

#ifndef MOBI_OCT_CWR
#define MOBI_OCT_CWR 16 // 128 level words per macroblock in registers: no effect on the default mix, -15 % kernel time on dense streams (848x480 config) against 32; 127 VGPRs, still 4 waves per SIMD
#endif
enum { CWR = MOBI_OCT_CWR };
__device__ __forceinline__ void recon_inter_oct_r1(const MobiReconArgs &A, uint8_t *L, uint32_t oi, int lane) {
  uint32_t rem, ox;
  const uint32_t clip = fastdiv(oi, A.qpc, A.magic_qpc, rem); // qpr / qpc: OCTETS per row / per clip for this kernel
  const uint32_t mby = fastdiv(rem, A.qpr, A.magic_qpr, ox);
  const uint32_t mbx0 = ox * 8, mbw = (uint32_t)A.mbw;
  const int nmb = (int)(mbw - mbx0 < 8 ? mbw - mbx0 : 8);
  const int S = A.stride, lgS = 31 - __builtin_clz((unsigned)S);
  const uint32_t ysz = (uint32_t)S * (uint32_t)A.height, slot_w = A.slot_bytes >> 2, ysz_w = ysz >> 2;
  uint8_t *clip_base = A.planes + (size_t)clip * A.clip_bytes;
  const int off0 = (int)(mby * 16 * (uint32_t)S + mbx0 * 16);
  const int g = lane & 7, j = lane >> 3; // adjacent lanes = adjacent macroblocks: a DMA round lays chunk j of the 8 macroblocks side by
                                         // side (slot j*8 + g), so the eight macroblocks of an LDS access fall on different banks

  // ---- stage A ----
  const uint4 *dp = (const uint4 *)(A.desc + (clip * (uint32_t)A.n_mbs + mby * mbw + mbx0) + g); // the table has slack past the last octet
  const uint4 d = dp[0], d2 = dp[1];
  const bool valid = g < nmb && (d.y & 1) == MOBI_MB_INTER;
  const int nl = (d.y >> 1) & 0x7F, kind2 = (d.y >> 26) & 3;
  const bool single = valid && nl == 1, dual = valid && kind2 != 0, multi = valid && nl > 1 && kind2 == 0;
  const uint32_t cbp6 = valid ? (d.y >> 8) & 0x3F : 0, ncoef = cbp6 ? d.z & 0x3FF : 0;
  // What the whole wave needs to know about the eight macroblocks travels by ballot (lane index = j*8 + g): coded areas and
  // transform kinds as bit area*8 + g (areas 0..3 in the low word, 4..5 in the high one), inter / cell-map flags as bit g.
  const unsigned long long mb64 = __builtin_amdgcn_ballot_w64(j < 6 && ((cbp6 >> j) & 1));
  const unsigned long long tb64 = __builtin_amdgcn_ballot_w64(j < 6 && ((d.y >> (14 + j)) & 1));
  const uint32_t m_lo = (uint32_t)mb64, m_hi = (uint32_t)(mb64 >> 32), t_lo = (uint32_t)tb64, t_hi = (uint32_t)(tb64 >> 32);
  const uint32_t inter_mask = (uint32_t)__builtin_amdgcn_ballot_w64(valid) & 0xFFu, multi_mask = (uint32_t)__builtin_amdgcn_ballot_w64(multi) & 0xFFu;
  if (inter_mask == 0) return; // nothing but intra macroblocks here
  const bool any_dual = __builtin_amdgcn_ballot_w64(dual) != 0;
  auto slot_off = [&](uint32_t ref) {
    int sl = A.ring_base - (int)ref;
    sl = sl < 0 ? sl + 6 : sl;
    return __umul24((uint32_t)sl, A.slot_bytes);
  };
  // leaf records as this lane needs them: luma rows 0..7 (iterations t < 4) / rows 8..15 (t >= 4); chroma rows 0..3 (u even) /
  // rows 4..7 (u odd).  DUAL top/bottom switches with the iteration, left/right with the lane's column.
  const uint32_t refA = slot_off((d.z >> 10) & 7);
  uint32_t refY0 = refA, refY1 = refA, refC0 = refA, refC1 = refA;
  int ypos0 = (int)d.w, ypos1 = (int)d.w, cpos0 = (int)d2.x, cpos1 = (int)d2.x;
  int yph0 = (d.z >> 16) & 3, yph1 = yph0, cph0 = (d.z >> 18) & 3, cph1 = cph0;
  if (any_dual) {
    const uint32_t refB = slot_off((d.z >> 13) & 7);
    const bool lr = kind2 == MOBI_DUAL_LR, tb = kind2 == MOBI_DUAL_TB;
    const bool by0 = dual && lr && (j & 2), by1 = dual && (tb || (lr && (j & 2)));
    const bool bc0 = dual && lr && (j & 1), bc1 = dual && (tb || (lr && (j & 1)));
    const int yphB = (d.z >> 20) & 3, cphB = (d.z >> 22) & 3;
    refY0 = by0 ? refB : refA; ypos0 = by0 ? (int)d2.y : ypos0; yph0 = by0 ? yphB : yph0;
    refY1 = by1 ? refB : refA; ypos1 = by1 ? (int)d2.y : ypos1; yph1 = by1 ? yphB : yph1;
    refC0 = bc0 ? refB : refA; cpos0 = bc0 ? (int)d2.z : cpos0; cph0 = bc0 ? cphB : cph0;
    refC1 = bc1 ? refB : refA; cpos1 = bc1 ? (int)d2.z : cpos1; cph1 = bc1 ? cphB : cph1;
  }
  const int hS = single ? S >> 1 : 0;
  auto rowoff = [&](int rows) { return single ? (uint32_t)rows << lgS : 0u; };
  const uint32_t ywin = single ? refA + (uint32_t)((int)d.w & ~15) : 0u;
  const uint32_t cwin = single ? refA + ysz + (uint32_t)((int)d2.x & ~15) : 0u;
  {
    const uint8_t *p0 = clip_base + (ywin + rowoff(j >> 1) + (uint32_t)(j & 1) * 16u);
    MOBI_DMA16(p0, L + O_L, 0);
    MOBI_DMA16(p0 + rowoff(4), L + O_L + 1024, 0);
    MOBI_DMA16(p0 + rowoff(8), L + O_L + 2048, 0);
    MOBI_DMA16(p0 + rowoff(12), L + O_L + 3072, 0);
    const uint8_t *p2 = clip_base + (cwin + rowoff(j >> 1) + (uint32_t)(j & 1) * 16u);
    MOBI_DMA16(p2, L + O_U, 0);
    MOBI_DMA16(p2 + rowoff(4), L + O_U + 1024, 0);
    MOBI_DMA16(p2 + hS, L + O_V, 0);
    MOBI_DMA16(p2 + hS + rowoff(4), L + O_V + 1024, 0);
  }
  const int quant = __builtin_amdgcn_readfirstlane((int)((d.y >> 20) & 63));
  const uint32_t *cw = A.payload + d.x + (multi ? MOBI_MV_CELLS : 0);
  uint32_t cwr[CWR] = {}; // lane (g, j) scatters words j, j+8, j+16, ... of macroblock g; the first 8*CWR of them travel in registers
  if ((uint32_t)j < ncoef) cwr[0] = cw[j];
  if (__builtin_amdgcn_ballot_w64(ncoef > 8) != 0) {
#pragma unroll
    for (int k = 1; k < 4; k++)
      if ((uint32_t)(8 * k + j) < ncoef) cwr[k] = cw[8 * k + j];
    if (CWR > 4 && __builtin_amdgcn_ballot_w64(ncoef > 32) != 0) {
#pragma unroll
      for (int k = 4; k < CWR; k++)
        if ((uint32_t)(8 * k + j) < ncoef) cwr[k] = cw[8 * k + j];
    }
  }
  // The 17th luma row and the 9th chroma rows of the windows (needed by the last row's vertical half-pel only) go
  // straight into the registers of the lanes that use them: a tenth DMA round would cost 1 KB of LDS, i.e. two waves per CU
  uint2 r16, r8u, r8v;
  {
    const int q = j & 3, qc = j & 1;
    const uint32_t oy = single ? refA + (uint32_t)(((int)d.w + (16 << lgS) + 4 * q) & ~3) : 0u;
    const uint32_t oc = single ? refA + ysz + (uint32_t)(((int)d2.x + (8 << lgS) + 4 * qc) & ~3) : 0u;
    r16 = *(const uint2_a4 *)(clip_base + oy);
    r8u = *(const uint2_a4 *)(clip_base + oc);
    r8v = *(const uint2_a4 *)(clip_base + oc + (S >> 1));
  }
  uint2 fx[12], fy[12]; // DUAL lanes: their own 2 x 8 bytes per iteration, straight into registers
  if (dual) {
    const int rr = j >> 2, q = j & 3;
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const uint32_t o = (t < 4 ? refY0 : refY1) + (uint32_t)(((t < 4 ? ypos0 : ypos1) + ((2 * t + rr) << lgS) + 4 * q) & ~3);
      fx[t] = *(const uint2_a4 *)(clip_base + o);
      fy[t] = *(const uint2_a4 *)(clip_base + o + S);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int row = (u & 1) * 4 + (j >> 1);
      const uint32_t o = ((u & 1) ? refC1 : refC0) + ysz + (uint32_t)((((u & 1) ? cpos1 : cpos0) + (u >> 1) * (S >> 1) + (row << lgS) + 4 * (j & 1)) & ~3);
      fx[8 + u] = *(const uint2_a4 *)(clip_base + o);
      fy[8 + u] = *(const uint2_a4 *)(clip_base + o + S);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wave_sync();

  // ---- stage B ----
  uint32_t mcv[12];
  auto stage_b = [&](auto with_dual) {
    constexpr bool DUAL = decltype(with_dual)::value;
    {
      // chroma: iteration u: plane u>>1, rows (u&1)*4 + (j>>1), q = j&1.  Dword w of row r: O_U/O_V + r*256 + (w>>2)*128 + g*16 + (w&3)*4,
      // row 8 sits in registers
      const int r4 = j >> 1, q = j & 1, w0 = (((int)d2.x & 15) + 4 * q) >> 2, w1 = w0 + 1;
      const int c0 = g * 16 + (w0 >> 2) * 128 + (w0 & 3) * 4, c1 = g * 16 + (w1 >> 2) * 128 + (w1 & 3) * 4;
      const uint32_t sh0 = cpos0 & 3, sh1 = cpos1 & 3;
      const bool p0a = cph0 == 0, p1a = cph0 == 1, p2a = cph0 == 2, p0b = cph1 == 0, p1b = cph1 == 1, p2b = cph1 == 2;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int pl = u >> 1, half = u & 1, rb = (pl ? O_V : O_U) + (half * 4 + r4) * 256;
        const int at0 = rb + c0, at1 = rb + c1;
        const bool last = half == 1 && r4 == 3; // row 7 -> row 8 (in registers)
        uint32_t x0 = lds32(L, at0), x1 = lds32(L, at1), y0 = lds32(L, last ? at0 : at0 + 256), y1 = lds32(L, last ? at1 : at1 + 256);
        if (half) { y0 = last ? (pl ? r8v.x : r8u.x) : y0; y1 = last ? (pl ? r8v.y : r8u.y) : y1; }
        if (DUAL) { x0 = dual ? fx[8 + u].x : x0; x1 = dual ? fx[8 + u].y : x1; y0 = dual ? fy[8 + u].x : y0; y1 = dual ? fy[8 + u].y : y1; }
        const bool second = DUAL && half;
        const uint32_t sh = second ? sh1 : sh0;
        mcv[8 + u] = mc4_lane(x0, x1, y0, y1, sh, sh * 8, second ? p0b : p0a, second ? p1b : p1a, second ? p2b : p2a);
      }
    }
    // the chroma windows are consumed: the dequant scales of this frame's quantizer land on the V rows while the luma is interpolated
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < MOBI_SCALE_STRIDE / 4) MOBI_DMA16((const uint8_t *)(A.scale + quant * MOBI_SCALE_STRIDE) + lane * 16, L + O_SC, 0);
    {
      // lane (g, rr = j>>2, q = j&3): row 2t + rr.  Dword w (0..7) of window row y: O_L + y*256 + (w>>2)*128 + g*16 + (w&3)*4
      // (rows 0..15; the four rounds are contiguous); row 16 sits in the registers of the lanes that need it.
      const int rr = j >> 2, q = j & 3, w0 = (((int)d.w & 15) + 4 * q) >> 2, w1 = w0 + 1;
      const int c0 = g * 16 + (w0 >> 2) * 128 + (w0 & 3) * 4, c1 = g * 16 + (w1 >> 2) * 128 + (w1 & 3) * 4;
      const int A0 = O_L + rr * 256 + c0, A1 = O_L + rr * 256 + c1;
      const uint32_t sh0 = ypos0 & 3, sh1 = ypos1 & 3;
      const bool p0a = yph0 == 0, p1a = yph0 == 1, p2a = yph0 == 2, p0b = yph1 == 0, p1b = yph1 == 1, p2b = yph1 == 2;
#pragma unroll
      for (int t = 0; t < 8; t++) {
        const int at0 = A0 + 512 * t, at1 = A1 + 512 * t;
        uint32_t x0 = lds32(L, at0), x1 = lds32(L, at1), y0, y1;
        if (t < 7) { y0 = lds32(L, at0 + 256); y1 = lds32(L, at1 + 256); }
        else { y0 = lds32(L, rr ? at0 : at0 + 256); y1 = lds32(L, rr ? at1 : at1 + 256); y0 = rr ? r16.x : y0; y1 = rr ? r16.y : y1; } // row 15 -> row 16
        if (DUAL) { x0 = dual ? fx[t].x : x0; x1 = dual ? fx[t].y : x1; y0 = dual ? fy[t].x : y0; y1 = dual ? fy[t].y : y1; }
        const bool second = DUAL && t >= 4;
        const uint32_t sh = second ? sh1 : sh0;
        mcv[t] = mc4_lane(x0, x1, y0, y1, sh, sh * 8, second ? p0b : p0a, second ? p1b : p1a, second ? p2b : p2a);
      }
    }
  };
  if (any_dual) stage_b(std::true_type{});
  else stage_b(std::false_type{});
  wave_sync();
  {
    const int oy = O_OUT_Y + (j >> 2) * 128 + g * 16 + (j & 3) * 4, oc = O_OUT_C + (j >> 1) * 64 + g * 8 + (j & 1) * 4;
#pragma unroll
    for (int t = 0; t < 8; t++) *(uint32_t *)(L + oy + 256 * t) = mcv[t];
#pragma unroll
    for (int u = 0; u < 4; u++) *(uint32_t *)(L + oc + (u >> 1) * 512 + (u & 1) * 256) = mcv[8 + u];
  }
  // B2: macroblocks with deeper partition trees, one at a time by the whole wave (as in the quad kernel)
  {
    const int yrow = lane >> 2, yc4 = (lane & 3) * 4;
    const int cv = (lane >> 4) & 1, crow = (lane & 15) >> 1, cc4 = (lane & 1) * 4;
    {
      uint32_t mm = multi_mask;
      while (mm) {
        const int gm = __builtin_ctz(mm);
        mm &= mm - 1;
        const uint32_t *cells = A.payload + __builtin_amdgcn_readlane(d.x, gm);
        const uint32_t *clip32 = (const uint32_t *)clip_base;
        const int offm = off0 + gm * 16;
        const int ybase = offm + (yrow << lgS) + yc4, cbase = (offm >> 1) + cv * (S >> 1) + (crow << lgS) + cc4;
        const uint2 yc = *(const uint2_a4 *)(cells + (yrow >> 1) * 8 + (yc4 >> 1));
        const uint4_a4 c4v = *(const uint4_a4 *)(cells + crow * 8 + cc4);
        const uint32_t cell[4] = {c4v.x, c4v.y, c4v.z, c4v.w};
        auto slot_of = [&](uint32_t c) { int s2 = A.ring_base - mobi_cell_ref(c); return __umul24((uint32_t)(s2 < 0 ? s2 + 6 : s2), slot_w); };
        const bool ysplit = __builtin_amdgcn_ballot_w64(yc.x != yc.y) != 0;
        const bool csplit = __builtin_amdgcn_ballot_w64(lane < 32 && (cell[0] != cell[1] || cell[0] != cell[2] || cell[0] != cell[3])) != 0;
        const int dxa = mobi_cell_dx(yc.x), dya = mobi_cell_dy(yc.x), dxb = mobi_cell_dx(yc.y), dyb = mobi_cell_dy(yc.y);
        const Win wa = fetch_win(clip32 + slot_of(yc.x), ybase + ((dya >> 1) << lgS) + (dxa >> 1), S);
        Win wb;
        if (ysplit) wb = fetch_win(clip32 + slot_of(yc.y), ybase + ((dyb >> 1) << lgS) + (dxb >> 1), S);
        int qx[4], qy[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { qx[k] = mobi_cell_dx(cell[k]) >> 1; qy[k] = mobi_cell_dy(cell[k]) >> 1; }
        Win wq[4];
        wq[0] = fetch_win(clip32 + slot_of(cell[0]) + ysz_w, cbase + ((qy[0] >> 1) << lgS) + (qx[0] >> 1), S);
        if (csplit) {
#pragma unroll
          for (int k = 1; k < 4; k++) wq[k] = fetch_win(clip32 + slot_of(cell[k]) + ysz_w, cbase + ((qy[k] >> 1) << lgS) + (qx[k] >> 1), S);
        }
        asm volatile("" ::: "memory");
        const uint32_t va = mc4_select(wa, (dxa & 1) | ((dya & 1) << 1));
        const uint32_t vb = ysplit ? mc4_select(wb, (dxb & 1) | ((dyb & 1) << 1)) : va;
        uint32_t cpred = mc4_select(wq[0], (qx[0] & 1) | ((qy[0] & 1) << 1));
        if (csplit) {
          cpred &= 0xFFu;
#pragma unroll
          for (int k = 1; k < 4; k++) cpred |= mc4_select(wq[k], (qx[k] & 1) | ((qy[k] & 1) << 1)) & (0xFFu << (8 * k));
        }
        *(uint32_t *)(L + O_OUT_Y + yrow * 128 + gm * 16 + yc4) = (va & 0x0000FFFFu) | (vb & 0xFFFF0000u);
        if (lane < 32) *(uint32_t *)(L + O_OUT_C + cv * 512 + crow * 64 + gm * 8 + cc4) = cpred;
      }
    }
  }

  // ---- stage C ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the scales
  wave_sync();
  if (m_lo | m_hi) {
    const int n_lo = __builtin_popcount(m_lo), n_ent = n_lo + __builtin_popcount(m_hi);
    int *coef = (int *)(L + O_COEF);
    {
      const uint32_t w = lane < 32 ? m_lo : m_hi, sh = lane & 31;
      if ((w >> sh) & 1) L[O_TAB + (lane < 32 ? 0 : n_lo) + __builtin_popcount(w & ((1u << sh) - 1))] = (uint8_t)lane;
    }
    int lo = 0, hi = 0;
    // 16 areas per round (two tiles of 8): half as many LDS round trips between the stages as with 8
    for (int base = 0; base < n_ent; base += 16) {
      {
        const uint4 z = uint4{0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; k++) *(uint4 *)(L + O_COEF + k * 1024 + lane * 16) = z;
      }
      wave_sync();
      auto scatter = [&](uint32_t e) {
        const int t = e & 0x1FF, level = (int32_t)e >> 16, ar = t >> 6, kk = (ar & 3) * 8 + g, p = t & 63;
        const bool chroma = ar >= 4; // entries are ordered by area, then macroblock: luma areas in the low mask word
        const int slot = (chroma ? n_lo : 0) + __builtin_popcount((chroma ? m_hi : m_lo) & ((1u << kk) - 1)) - base;
        const int si = (((chroma ? t_hi : t_lo) >> kk) & 1) ? p : 64 + (p & 15);
        const int scale = (int)lds32(L, O_SC + si * 4);
        if ((unsigned)slot < 16u) coef[slot * 64 + p] = __mul24(scale, level);
      };
#pragma unroll
      for (int k = 0; k < CWR; k++) {
        const bool mine = (uint32_t)(8 * k + j) < ncoef;
        if (k && __builtin_amdgcn_ballot_w64(mine) == 0) break;
        uint32_t e = cwr[k];
        asm volatile("" : "+v"(e));
        if (mine) scatter(e);
      }
      for (uint32_t i = 8u * CWR + (uint32_t)j; __builtin_amdgcn_ballot_w64(i < ncoef) != 0; i += 8) // beyond the registers: one exposed round trip per 8 words
        if (i < ncoef) scatter(cw[i]);
      wave_sync();
      const int r = lane & 7;
      int kx[2];
      bool actx[2], is8x[2];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int idx = base + 8 * h + (lane >> 3);
        actx[h] = idx < n_ent;
        kx[h] = actx[h] ? L[O_TAB + idx] : 0;
        is8x[h] = ((kx[h] < 32 ? t_lo : t_hi) >> (kx[h] & 31)) & 1;
      }
#pragma unroll
      for (int h = 0; h < 2; h++) {
        int *tile = coef + 64 * (8 * h + (lane >> 3));
        if (actx[h]) idct_pass1(tile, tile, is8x[h], r);
      }
      wave_sync();
#pragma unroll
      for (int h = 0; h < 2; h++) {
        if (actx[h]) {
          const int ge = kx[h] & 7, a = kx[h] >> 3;
          uint8_t *px = a < 4 ? L + O_OUT_Y + (a >> 1) * 8 * 128 + ge * 16 + (a & 1) * 8 : L + O_OUT_C + (a - 4) * 512 + ge * 8;
          idct_pass2_q(coef + 64 * (8 * h + (lane >> 3)), is8x[h], r, px, a < 4 ? 128 : 64, lo, hi);
        }
      }
      wave_sync();
    }
    if (lo < -64 || hi > 319) atomicOr(&A.fault[clip], 1);
  }

  // ---- stage D: whole rows, 128 B of luma and 8 B per macroblock of chroma ----
  uint8_t *y0 = clip_base + (uint32_t)A.ring_base * A.slot_bytes;
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const int i = lane + 64 * it, gq = i & 7, yrow = i >> 3;
    if ((inter_mask >> gq) & 1) {
      *(uint4 *)(y0 + (off0 + (yrow << lgS) + gq * 16)) = *(const uint4 *)(L + O_OUT_Y + yrow * 128 + gq * 16);
      const int row = yrow & 7; // chroma: plane = it, row = (i >> 3) & 7
      *(uint2 *)(y0 + ysz + ((off0 >> 1) + it * (S >> 1) + (row << lgS) + gq * 8)) = *(const uint2 *)(L + O_OUT_C + it * 512 + row * 64 + gq * 8);
    }
  }
}
extern "C" __global__ __launch_bounds__(64) void mobi_recon_inter8_r1(MobiReconArgs A) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[O_BYTES];
  const uint32_t oi = (blockIdx.x & 7) * A.inter_per_xcd + (blockIdx.x >> 3);
  if (oi >= A.qpc * (uint32_t)A.n_clips) return;
  recon_inter_oct_r1(A, lds, oi, (int)threadIdx.x);
}

// =====================================================================================================
// mobi_recon_inter8 (r02): one wavefront per octet, second generation
// =====================================================================================================
// What r01's counters said about the kernel above (profiles/r01_pmc_summary.txt): the texture addresser is busy 80 % of
// the launch (44 vector-memory instructions per wave, 24 of them the 8-byte register fetches of the two-half macroblocks),
// every wave waits for two memory round trips in series (descriptor, then windows), and the integer VALU is not one
// machine: two-operand adds / shifts / logic ops issue in ~2.7 cycles per wave, everything with three operands, byte
// selects, compares, SDWA and 16-bit packed forms in ~4.3 (tools/ubench/oprate.hip).  Hence:
//   * ONE fetch path for every macroblock with whole leaves (16x16, two 16x8, two 8x16): the DMA rounds take a per-lane
//     source address, so the rows of a top/bottom pair and the 16-byte halves of a left/right pair simply come from the
//     other leaf's position.  Windows start at a 4-byte boundary (not 16): 20 bytes of a row always fit the two chunks,
//     9 + 3 chroma bytes fit ONE chunk, and the dword a lane reads no longer depends on the motion vector.
//     19 vector-memory instructions per wave instead of 44.
//   * a lane owns 8 consecutive luma rows x 4 pixels (4 chroma rows x 4): it lies inside one leaf whatever the split, the
//     row below of one row is the row of the next (masked bytes and the horizontal average are computed once per row),
//     and the extra row under a lane's rows (9th / 5th) comes straight into its registers.
//   * CopyBlock (MD.cs:424-452) by v_perm_b32 (byte window out of two dwords) and v_lerp_u8: (a>>1)+(b>>1) per byte is
//     the byte average of a & 0xFE and b & 0xFE; phase 0 uses the same instructions with mask 0xFF and b = a.
//   * macroblocks with deeper trees fetch their MV cells beside the DMA rounds and their pixels under the MC of the others.
//   * coefficient tiles at a pitch of 72 words (the transposing stores of the 8 lanes of 4 areas hit 32 different banks),
//     8x8 areas sorted in front of 4x4 ones so that a half round usually runs one kind of butterfly.
namespace {
enum {
  P_L = 0,        // luma windows: dword w (0..7) of row y (0..15) of macroblock g at y*256 + (w>>2)*128 + g*16 + (w&3)*4
  P_C = 4096,     // chroma windows, chunk 0: plane pl, row r at pl*1024 + r*128 + g*16
  P_C1 = 6144,    //   chunk 1: the right half of a LEFT/RIGHT pair (other macroblocks leave it unused)
  P_BYTES = 8192,
  // after motion compensation:
  P_OUT_Y = 0,    // 16 rows x 128 B
  P_OUT_C = 2048, // 2 planes x 8 rows x 64 B
  P_COEF = 3072,  // 16 tiles of P_TILE words
  P_TILE = 72,
  P_TAB = 7680,   // entry -> area*8 + g (<= 48 bytes)
  P_SC = 7744     // dequant scales (320 B): on top of the last V rows of chunk 1, once the chroma has been interpolated
};
// N output rows of 4 pixels from N + 1 window rows (x0[i], x1[i] = the two aligned dwords holding row i's 5 bytes)
template <int N>
__device__ __forceinline__ void mc_rows(const uint32_t (&x0)[N + 1], const uint32_t (&x1)[N + 1], uint32_t sh, int phase, uint32_t *out) {
  const uint32_t selA = 0x03020100u + sh * 0x01010101u;
  const bool ph0 = phase == 0, ph2 = phase == 2, ph3 = phase == 3;
  const uint32_t selB = ph0 ? selA : selA + 0x01010101u;
  const uint32_t Em = ph0 ? 0xFFFFFFFFu : 0xFEFEFEFEu, m2 = ph2 ? 0xFFFFFFFFu : 0u;
  uint32_t ae[N + 1], be[N + 1], s[N + 1];
#pragma unroll
  for (int i = 0; i <= N; i++) {
    ae[i] = __builtin_amdgcn_perm(x1[i], x0[i], selA) & Em;
    be[i] = __builtin_amdgcn_perm(x1[i], x0[i], selB) & Em;
  }
#pragma unroll
  for (int i = 0; i < N; i++) s[i] = __builtin_amdgcn_lerp(ae[i], (ae[i + 1] & m2) | (be[i] & ~m2), 0u); // v_bfi_b32 (a select between two
                                                                                   // array elements would become a select between their addresses)
  s[N] = __builtin_amdgcn_lerp(ae[N], be[N], 0u); // the horizontal average of the row below: phase 3 only
#pragma unroll
  for (int i = 0; i < N; i++) {
    const uint32_t p3 = __builtin_amdgcn_lerp(s[i] & 0xFEFEFEFEu, s[i + 1] & 0xFEFEFEFEu, 0u);
    out[i] = ph3 ? p3 : s[i];
  }
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
template <int PROF, int CWR>
__device__ __forceinline__ void recon_inter_oct(const MobiReconArgs &A, uint8_t *L, uint32_t oi, int lane) {
  unsigned long long pt[7] = {0, 0, 0, 0, 0, 0, 0};
  if (PROF) pt[0] = prof_stamp();
  uint32_t rem, ox;
  const uint32_t clip = fastdiv(oi, A.qpc, A.magic_qpc, rem); // qpr / qpc: OCTETS per row / per clip for this kernel
  const uint32_t mby = fastdiv(rem, A.qpr, A.magic_qpr, ox);
  const uint32_t mbx0 = ox * 8, mbw = (uint32_t)A.mbw;
  const int nmb = (int)(mbw - mbx0 < 8 ? mbw - mbx0 : 8);
  const int S = A.stride, lgS = 31 - __builtin_clz((unsigned)S);
  const uint32_t ysz = (uint32_t)S * (uint32_t)A.height, slot_w = A.slot_bytes >> 2, ysz_w = ysz >> 2;
  uint8_t *clip_base = A.planes + (size_t)clip * A.clip_bytes;
  const int off0 = (int)(mby * 16 * (uint32_t)S + mbx0 * 16);
  const int g = lane & 7, j = lane >> 3; // adjacent lanes = adjacent macroblocks: chunk j of the 8 macroblocks lies side by side in LDS
  unsigned long long pa = 0, pb = 0;
  if (PROF) { asm volatile("" : : "s"(off0), "s"(clip)); pa = prof_stamp(); }

  // ---- stage A: descriptor, then every global read of the octet ----
  const uint4 *dp = (const uint4 *)(A.desc + (clip * (uint32_t)A.n_mbs + mby * mbw + mbx0) + g); // the table has slack past the last octet
  const uint4 d = dp[0], d2 = dp[1];
  if (PROF) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pb = prof_stamp(); }
  const bool valid = g < nmb && (d.y & 1) == MOBI_MB_INTER;
  const int nl = (d.y >> 1) & 0x7F, kind2 = (d.y >> 26) & 3;
  const bool win = valid && (nl == 1 || kind2 != 0);         // whole leaves: fetched through the LDS windows
  const bool multi = valid && nl > 1 && kind2 == 0;          // deeper tree: MV cell map
  const bool tb = valid && kind2 == MOBI_DUAL_TB, lr = valid && kind2 == MOBI_DUAL_LR;
  const uint32_t cbp6 = valid ? (d.y >> 8) & 0x3F : 0, ncoef = cbp6 ? d.z & 0x3FF : 0;
  const unsigned long long mb64 = __builtin_amdgcn_ballot_w64(j < 6 && ((cbp6 >> j) & 1));      // bit area*8 + g
  const unsigned long long tb64 = __builtin_amdgcn_ballot_w64(j < 6 && ((d.y >> (14 + j)) & 1));
  const uint32_t m_lo = (uint32_t)mb64, m_hi = (uint32_t)(mb64 >> 32), t_lo = (uint32_t)tb64, t_hi = (uint32_t)(tb64 >> 32);
  const uint32_t inter_mask = (uint32_t)__builtin_amdgcn_ballot_w64(valid) & 0xFFu, multi_mask = (uint32_t)__builtin_amdgcn_ballot_w64(multi) & 0xFFu;
  if (inter_mask == 0) return; // nothing but intra macroblocks here
  const bool any_lr = __builtin_amdgcn_ballot_w64(lr) != 0;
  auto slot_off = [&](uint32_t ref) {
    int sl = A.ring_base - (int)ref;
    sl = sl < 0 ? sl + 6 : sl;
    return __umul24((uint32_t)sl, A.slot_bytes); // slot_bytes < 2^24: checked by mobi_launch_inter
  };
  const uint32_t refA = slot_off((d.z >> 10) & 7), refB = slot_off((d.z >> 13) & 7);
  const int posA = (int)d.w, cposA = (int)d2.x, posB = (int)d2.y, cposB = (int)d2.z;
  {
    // luma rounds: lane (g, j) brings chunk (row 4t + (j >> 1), half j & 1).  A row's two chunks start at the leaf's position
    // rounded down to 4 bytes; the right half of a LEFT/RIGHT pair is the first chunk of leaf B's columns 8..15
    const int h = j & 1, r4 = win ? j >> 1 : 0;
    const bool rB = lr && h;
    const int pT = win ? (rB ? posB + 8 : posA) : 0, xT = (win && !lr) ? h * 16 : 0;
    const uint32_t fT = win ? (rB ? refB : refA) : 0u;
    const int pU = tb ? posB : pT;
    const uint32_t fU = tb ? refB : fT;
    const uint8_t *sT = clip_base + fT + (uint32_t)(((pT + (r4 << lgS)) & ~3) + xT);
    const uint8_t *sU = clip_base + fU + (uint32_t)(((pU + ((8 + r4) << lgS)) & ~3) + xT);
    const int s4 = win ? 4 << lgS : 0;
    MOBI_DMA16(sT, L + P_L, 0);
    MOBI_DMA16(sT + s4, L + P_L + 1024, 0);
    MOBI_DMA16(sU, L + P_L + 2048, 0);
    MOBI_DMA16(sU + s4, L + P_L + 3072, 0);
    // chroma rounds: round = plane, lane j = row; one chunk per row
    const bool cB = tb && j >= 4;
    const int cp = win ? (cB ? cposB : cposA) : 0;
    const uint32_t cf = win ? (cB ? refB : refA) + ysz : 0u;
    const uint8_t *sC = clip_base + cf + (uint32_t)((cp + ((win ? j : 0) << lgS)) & ~3);
    MOBI_DMA16(sC, L + P_C, 0);
    MOBI_DMA16(sC + (win ? S >> 1 : 0), L + P_C + 1024, 0);
    if (any_lr) {
      if (lr) {
        const uint8_t *sD = clip_base + refB + ysz + (uint32_t)((cposB + 4 + (j << lgS)) & ~3);
        MOBI_DMA16(sD, L + P_C1, 0);
        MOBI_DMA16(sD + (S >> 1), L + P_C1 + 1024, 0);
      }
    }
  }
  // this lane's leaf for the luma rows it interpolates: lane (g, rr = j >> 2, q = j & 3) = rows 8rr..8rr+7, pixels 4q..4q+3
  const int rr = j >> 2, q = j & 3;
  const bool yB = (tb && rr) || (lr && q >= 2);
  const int ypos = yB ? posB : posA, yph = (d.z >> (yB ? 20 : 16)) & 3;
  // ... and for its chroma samples: lane (g, pl = j >> 2, ch = (j >> 1) & 1, qc = j & 1) = plane pl, rows 4ch..4ch+3, samples 4qc..4qc+3
  const int pl = j >> 2, ch = (j >> 1) & 1, qc = j & 1;
  const bool cBl = (tb && ch) || (lr && qc);
  const int cpos = cBl ? cposB : cposA, cph = (d.z >> (cBl ? 22 : 18)) & 3;
  // the row under the lane's rows (luma row 8rr + 8, chroma row 4ch + 4): straight into registers
  uint2 hy, hc;
  {
    // (only a vertical half-pel reads it: the other lanes all point at one line instead -- the kernel is bound by the number of
    // 64-byte requests a CU's L1 can send to the L2, ~0.15 per clock, and a row costs at least one)
    const uint32_t oy = (win && (yph & 2)) ? (yB ? refB : refA) + (uint32_t)((ypos + ((8 * rr + 8) << lgS) + 4 * q) & ~3) : 0u;
    const uint32_t oc = (win && (cph & 2)) ? (cBl ? refB : refA) + ysz + (uint32_t)((cpos + pl * (S >> 1) + ((4 * ch + 4) << lgS) + 4 * qc) & ~3) : 0u;
    hy = *(const uint2_a4 *)(clip_base + oy);
    hc = *(const uint2_a4 *)(clip_base + oc);
  }
  const int quant = __builtin_amdgcn_readfirstlane((int)((d.y >> 20) & 63));
  const uint32_t *cw = A.payload + d.x + (multi ? MOBI_MV_CELLS : 0);
  uint32_t cwr[CWR] = {}; // lane (g, j) scatters words j, j+8, j+16, ... of macroblock g; the first 8*CWR of them travel in registers
  if ((uint32_t)j < ncoef) cwr[0] = cw[j];
  if (__builtin_amdgcn_ballot_w64(ncoef > 8) != 0) {
#pragma unroll
    for (int k = 1; k < 4; k++)
      if ((uint32_t)(8 * k + j) < ncoef) cwr[k] = cw[8 * k + j];
    if (CWR > 4 && __builtin_amdgcn_ballot_w64(ncoef > 32) != 0) {
#pragma unroll
      for (int k = 4; k < CWR; k++)
        if ((uint32_t)(8 * k + j) < ncoef) cwr[k] = cw[8 * k + j];
    }
  }
  // deeper trees: the first such macroblock's MV cells travel with everything else (the whole wave works for it later:
  // lane = (row lane >> 2, pixels 4 * (lane & 3)) for luma, lanes 0..31 = (plane, row, 4 samples) for chroma)
  const int yrow = lane >> 2, yc4 = (lane & 3) * 4;
  const int cv = (lane >> 4) & 1, crow = (lane & 15) >> 1, cc4 = (lane & 1) * 4;
  uint2 yc0 = uint2{0, 0};
  uint4 c4v0 = uint4{0, 0, 0, 0};
  if (multi_mask) {
    const uint32_t *cells = A.payload + __builtin_amdgcn_readlane(d.x, __builtin_ctz(multi_mask));
    yc0 = *(const uint2_a4 *)(cells + (yrow >> 1) * 8 + (yc4 >> 1));
    c4v0 = *(const uint4_a4 *)(cells + crow * 8 + cc4);
  }
  if (PROF) pt[1] = prof_stamp();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wave_sync();
  if (PROF) pt[2] = prof_stamp();

  // deeper trees: issue the first one's pixel fetches now, consume them after the others' motion compensation
  const uint32_t *clip32 = (const uint32_t *)clip_base;
  auto slot_of = [&](uint32_t c) { int s2 = A.ring_base - mobi_cell_ref(c); return __umul24((uint32_t)(s2 < 0 ? s2 + 6 : s2), slot_w); };
  struct Deep { Win wa, wb, wq[4]; uint2 yc; uint32_t cell[4]; bool ysplit, csplit; };
  auto deep_fetch = [&](Deep &D, int gm, uint2 yc, uint4 c4v) {
    D.yc = yc;
    D.cell[0] = c4v.x; D.cell[1] = c4v.y; D.cell[2] = c4v.z; D.cell[3] = c4v.w;
    const int offm = off0 + gm * 16;
    const int ybase = offm + (yrow << lgS) + yc4, cbase = (offm >> 1) + cv * (S >> 1) + (crow << lgS) + cc4;
    // a lane's 4 luma pixels sit under two cells, its 4 chroma samples under four; for the common splits they are the same cell
    D.ysplit = __builtin_amdgcn_ballot_w64(yc.x != yc.y) != 0;
    D.csplit = __builtin_amdgcn_ballot_w64(lane < 32 && (D.cell[0] != D.cell[1] || D.cell[0] != D.cell[2] || D.cell[0] != D.cell[3])) != 0;
    const int dxa = mobi_cell_dx(yc.x), dya = mobi_cell_dy(yc.x), dxb = mobi_cell_dx(yc.y), dyb = mobi_cell_dy(yc.y);
    D.wa = fetch_win(clip32 + slot_of(yc.x), ybase + ((dya >> 1) << lgS) + (dxa >> 1), S);
    if (D.ysplit) D.wb = fetch_win(clip32 + slot_of(yc.y), ybase + ((dyb >> 1) << lgS) + (dxb >> 1), S);
    {
      const int qx = mobi_cell_dx(D.cell[0]) >> 1, qy = mobi_cell_dy(D.cell[0]) >> 1;
      D.wq[0] = fetch_win(clip32 + slot_of(D.cell[0]) + ysz_w, cbase + ((qy >> 1) << lgS) + (qx >> 1), S);
    }
    if (D.csplit) {
#pragma unroll
      for (int k = 1; k < 4; k++) {
        const int qx = mobi_cell_dx(D.cell[k]) >> 1, qy = mobi_cell_dy(D.cell[k]) >> 1;
        D.wq[k] = fetch_win(clip32 + slot_of(D.cell[k]) + ysz_w, cbase + ((qy >> 1) << lgS) + (qx >> 1), S);
      }
    }
  };
  auto deep_finish = [&](int gm, const Deep &D) {
    const int dxa = mobi_cell_dx(D.yc.x), dya = mobi_cell_dy(D.yc.x), dxb = mobi_cell_dx(D.yc.y), dyb = mobi_cell_dy(D.yc.y);
    const uint32_t va = mc4_select(D.wa, (dxa & 1) | ((dya & 1) << 1));
    const uint32_t vb = D.ysplit ? mc4_select(D.wb, (dxb & 1) | ((dyb & 1) << 1)) : va;
    auto cph_of = [&](uint32_t c) { const int qx = mobi_cell_dx(c) >> 1, qy = mobi_cell_dy(c) >> 1; return (qx & 1) | ((qy & 1) << 1); };
    uint32_t cpred = mc4_select(D.wq[0], cph_of(D.cell[0]));
    if (D.csplit) {
      cpred &= 0xFFu;
#pragma unroll
      for (int k = 1; k < 4; k++) cpred |= mc4_select(D.wq[k], cph_of(D.cell[k])) & (0xFFu << (8 * k));
    }
    *(uint32_t *)(L + P_OUT_Y + yrow * 128 + gm * 16 + yc4) = (va & 0x0000FFFFu) | (vb & 0xFFFF0000u);
    if (lane < 32) *(uint32_t *)(L + P_OUT_C + cv * 512 + crow * 64 + gm * 8 + cc4) = cpred;
  };
  Deep D0;
  if (multi_mask) deep_fetch(D0, __builtin_ctz(multi_mask), yc0, c4v0);

  // ---- stage B: motion compensation ----
  uint32_t mcv[12];
  {
    // chroma first: its windows make room for the dequant scales
    const bool c1 = lr && qc; // the right half of a LEFT/RIGHT pair has its own chunk; everybody else reads dwords qc, qc + 1 of chunk 0
    const int base = (c1 ? P_C1 : P_C + qc * 4) + pl * 1024 + ch * 512 + g * 16;
    uint32_t x0[5], x1[5];
#pragma unroll
    for (int k = 0; k < 4; k++) { x0[k] = lds32(L, base + k * 128); x1[k] = lds32(L, base + k * 128 + 4); }
    x0[4] = hc.x; x1[4] = hc.y;
    mc_rows<4>(x0, x1, (uint32_t)cpos & 3u, cph, mcv + 8);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane < MOBI_SCALE_STRIDE / 4) MOBI_DMA16((const uint8_t *)(A.scale + quant * MOBI_SCALE_STRIDE) + lane * 16, L + P_SC, 0);
  {
    const int w0 = q + ((lr && q >= 2) ? 2 : 0), w1 = w0 + 1; // columns 8..15 of a LEFT/RIGHT pair start their own chunk
    const int base = P_L + rr * 2048 + g * 16;
    const int a0 = base + (w0 >> 2) * 128 + (w0 & 3) * 4, a1 = base + (w1 >> 2) * 128 + (w1 & 3) * 4;
    uint32_t x0[9], x1[9];
#pragma unroll
    for (int k = 0; k < 8; k++) { x0[k] = lds32(L, a0 + k * 256); x1[k] = lds32(L, a1 + k * 256); }
    x0[8] = hy.x; x1[8] = hy.y;
    mc_rows<8>(x0, x1, (uint32_t)ypos & 3u, yph, mcv);
  }
  wave_sync();
  {
    const int oy = P_OUT_Y + rr * 1024 + g * 16 + q * 4, oc = P_OUT_C + pl * 512 + ch * 256 + g * 8 + qc * 4;
#pragma unroll
    for (int k = 0; k < 8; k++) *(uint32_t *)(L + oy + 128 * k) = mcv[k];
#pragma unroll
    for (int k = 0; k < 4; k++) *(uint32_t *)(L + oc + 64 * k) = mcv[8 + k];
  }
  if (PROF) pt[3] = prof_stamp();
  if (multi_mask) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the scales too)
    deep_finish(__builtin_ctz(multi_mask), D0);
    uint32_t mm = multi_mask & (multi_mask - 1);
    while (mm) { // a second, third ... macroblock with a deep tree in the same octet: one exposed round trip each (rare)
      const int gm = __builtin_ctz(mm);
      mm &= mm - 1;
      const uint32_t *cells = A.payload + __builtin_amdgcn_readlane(d.x, gm);
      const uint2 yc = *(const uint2_a4 *)(cells + (yrow >> 1) * 8 + (yc4 >> 1));
      const uint4 c4v = *(const uint4_a4 *)(cells + crow * 8 + cc4);
      Deep Dn;
      deep_fetch(Dn, gm, yc, c4v);
      asm volatile("" ::: "memory");
      deep_finish(gm, Dn);
    }
  }
  if (PROF) pt[4] = prof_stamp();

  // ---- stage C: residual ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the scales
  wave_sync();
  if (m_lo | m_hi) {
    // entries: 8x8 areas first, then the areas made of 4x4 blocks; inside a kind by area, then macroblock
    const uint32_t m8_lo = m_lo & t_lo, m8_hi = m_hi & t_hi, m4_lo = m_lo & ~t_lo, m4_hi = m_hi & ~t_hi;
    const int n8_lo = __builtin_popcount(m8_lo), n8 = n8_lo + __builtin_popcount(m8_hi);
    const int n4_lo = __builtin_popcount(m4_lo), n_ent = n8 + n4_lo + __builtin_popcount(m4_hi);
    auto slot_of_entry = [=](int kk, bool chroma, bool is8) { // kk = (area & 3) * 8 + g.  (By value, and masks by arithmetic: a select between
      const uint32_t flip = is8 ? 0u : 0xFFFFFFFFu;          // captured variables turns into a select between their addresses, i.e. scratch.)
      const uint32_t mask = chroma ? m_hi & (t_hi ^ flip) : m_lo & (t_lo ^ flip);
      const int first = (is8 ? 0 : n8) + (chroma ? (is8 ? n8_lo : n4_lo) : 0);
      return first + __builtin_popcount(mask & ((1u << kk) - 1u));
    };
    int *coef = (int *)(L + P_COEF);
    {
      const bool hi = lane >= 32;
      const int kk = lane & 31;
      if (((hi ? m_hi : m_lo) >> kk) & 1) L[P_TAB + slot_of_entry(kk, hi, ((hi ? t_hi : t_lo) >> kk) & 1)] = (uint8_t)lane;
    }
    int lo = 0, hi = 0;
    for (int base = 0; base < n_ent; base += 16) {
      {
        const uint4 z = uint4{0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; k++) *(uint4 *)(L + P_COEF + k * 1024 + lane * 16) = z;
        if (lane < (16 * P_TILE * 4 - 4096) / 16) *(uint4 *)(L + P_COEF + 4096 + lane * 16) = z;
      }
      wave_sync();
      auto scatter = [&](uint32_t e) {
        const int t = e & 0x1FF, level = (int32_t)e >> 16, ar = t >> 6, kk = (ar & 3) * 8 + g, p = t & 63;
        const bool chroma = ar >= 4, is8 = ((chroma ? t_hi : t_lo) >> kk) & 1;
        const int slot = slot_of_entry(kk, chroma, is8) - base;
        const int si = is8 ? p : 64 + (p & 15);
        const int scale = (int)lds32(L, P_SC + si * 4);
        if ((unsigned)slot < 16u) coef[slot * P_TILE + p] = __mul24(scale, level);
      };
#pragma unroll
      for (int k = 0; k < CWR; k++) {
        const bool mine = (uint32_t)(8 * k + j) < ncoef;
        if (k && __builtin_amdgcn_ballot_w64(mine) == 0) break;
        uint32_t e = cwr[k];
        asm volatile("" : "+v"(e));
        if (mine) scatter(e);
      }
      for (uint32_t i = 8u * CWR + (uint32_t)j; __builtin_amdgcn_ballot_w64(i < ncoef) != 0; i += 8) // beyond the registers: one exposed round trip per 8 words
        if (i < ncoef) scatter(cw[i]);
      wave_sync();
      const int r = lane & 7;
      int kx[2];
      bool actx[2], is8x[2];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int idx = base + 8 * h + (lane >> 3);
        actx[h] = idx < n_ent;
        kx[h] = actx[h] ? L[P_TAB + idx] : 0;
        is8x[h] = idx < n8;
      }
#pragma unroll
      for (int h = 0; h < 2; h++) {
        int *tile = coef + P_TILE * (8 * h + (lane >> 3));
        if (actx[h]) idct_pass1(tile, tile, is8x[h], r);
      }
      wave_sync();
#pragma unroll
      for (int h = 0; h < 2; h++) {
        if (actx[h]) {
          const int ge = kx[h] & 7, a = kx[h] >> 3;
          uint8_t *px = a < 4 ? L + P_OUT_Y + (a >> 1) * 8 * 128 + ge * 16 + (a & 1) * 8 : L + P_OUT_C + (a - 4) * 512 + ge * 8;
          idct_pass2_q(coef + P_TILE * (8 * h + (lane >> 3)), is8x[h], r, px, a < 4 ? 128 : 64, lo, hi);
        }
      }
      wave_sync();
    }
    if (lo < -64 || hi > 319) atomicOr(&A.fault[clip], 1); // clamp table domain (MobiConst.cs:587)
  }
  if (PROF) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); pt[5] = prof_stamp(); }

  // ---- stage D: whole rows, 128 B of luma and 8 B per macroblock of chroma ----
  uint8_t *y0 = clip_base + (uint32_t)A.ring_base * A.slot_bytes;
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const int i = lane + 64 * it, gq = i & 7, row16 = i >> 3;
    if ((inter_mask >> gq) & 1) {
      *(uint4 *)(y0 + (off0 + (row16 << lgS) + gq * 16)) = *(const uint4 *)(L + P_OUT_Y + row16 * 128 + gq * 16);
      const int row = row16 & 7; // chroma: plane = it, row = (i >> 3) & 7
      *(uint2 *)(y0 + ysz + ((off0 >> 1) + it * (S >> 1) + (row << lgS) + gq * 8)) = *(const uint2 *)(L + P_OUT_C + it * 512 + row * 64 + gq * 8);
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
#define MOBI_OCT_KERNEL(NAME, WAVES, PROF, NCWR)                                                      \
  extern "C" __global__ __launch_bounds__(64, WAVES) void NAME(MobiReconArgs A) {                      \
    __shared__ __attribute__((aligned(16))) uint8_t lds[P_BYTES];                                      \
    const uint32_t oi = (blockIdx.x & 7) * A.inter_per_xcd + (blockIdx.x >> 3);                        \
    if (oi >= A.qpc * (uint32_t)A.n_clips) return;                                                     \
    recon_inter_oct<PROF, NCWR>(A, lds, oi, (int)threadIdx.x);                                         \
  }
// 5 waves per SIMD (96 VGPRs; the 8 KB of LDS allow exactly 20 waves per CU) with 96 level words per macroblock in registers:
// 24576 clips 640x480: 7.59 ms per launch against 7.95 with 4 waves and 128 words (r2c/bench_variants.txt)
MOBI_OCT_KERNEL(mobi_recon_inter8, 5, 0, 12)
MOBI_OCT_KERNEL(mobi_recon_inter8_prof, 5, 1, 12)
MOBI_OCT_KERNEL(mobi_recon_inter8_w4, 4, 0, 16) // MOBI_OCT_VARIANT=1: A/B runs

// =====================================================================================================
// intra macroblocks of one dependency level
// =====================================================================================================
namespace {
// byte load that bypasses this CU's L1 (sc1): issued now, valid only after ld_wait6() -- the compiler puts a full wait
// behind every __hip_atomic_load, which serialises the halo into six round trips
__device__ __forceinline__ uint32_t ld_u8_sc1(const uint8_t *base, uint32_t off) { // base: wave-uniform
  uint32_t v;
#if defined(__HIP_DEVICE_COMPILE__)
  // s_nop 4: when the compiler has to make `base` scalar with v_readfirstlane, a VMEM instruction may not read that
  // SGPR for 5 wait states, and nothing inside an asm string is padded for us (cdna_hip_programming.md 5.7)
  asm volatile("s_nop 4\n\tglobal_load_ubyte %0, %1, %2 sc1" : "=v"(v) : "v"(off), "s"(base) : "memory");
#else
  v = base[off];
#endif
  return v;
}
__device__ __forceinline__ void ld_wait6(uint32_t (&v)[6]) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]) : : "memory");
#endif
}
struct TileNb {
  const uint8_t *t;
  int by, bx;
  __device__ __forceinline__ int operator()(int dy, int dx) const { return t[(by + dy + 1) * TP + 4 + bx + dx]; }
};
// predict one block on the tile (all lanes call; lanes >= n*n idle) and add its residual when coded.  The residuals of the whole
// macroblock were computed beforehand (they do not depend on the prediction): res = the area's 8x8 residual tile.
__device__ __forceinline__ void run_block(uint8_t *tile, int by, int bx, int n, int mode, int param, bool coded,
                                          const int *res, int sub, int block_off, bool is_uv,
                                          int S, int lane, int *fault) {
  asm volatile("" : "+v"(lane)); // per-lane predicates are recomputed here: hoisted out of the block loops they end up as ~50 spilled SGPR pairs
  TileNb nb{tile, by, bx};
  const int y = (n == 8) ? lane >> 3 : lane >> 2, x = (n == 8) ? lane & 7 : lane & 3; // this lane's pixel of an 8x8 / 4x4 block
  const int ri = (n == 8) ? lane : ((sub >> 1) * 4 + y) * 8 + (sub & 1) * 4 + x;        // ... and its place in the area's residual tile
  bool add_pending = coded;
  if (mode == 2) {
    const int wpr = n >> 2, nw = n * wpr; // words per row, words in block
    if (lane < nw) {
      const int yy = lane / wpr, x0 = (lane % wpr) * 4;
      const uint32_t w = mobi_plane_word(n, param, yy, x0, nb);
      *(uint32_t *)(tile + (by + yy + 1) * TP + 4 + bx + x0) = w;
    }
    wave_sync();
  } else if (mode != 9) {
    const int vfix = is_uv && (block_off & (S - 1)) >= (S >> 1);                                      // MD.cs:1886
    const int left_avail = ((block_off - (vfix ? (S >> 1) : 0)) & (S - 1)) != 0, top_avail = block_off >= S; // :1923-1924
    if (lane < n * n) {
      int v = mobi_pred_px(mode, n, y, x, top_avail, left_avail, nb);
      if (coded) v = mobi_add_clamp(v, res[ri], fault);
      tile[(by + y + 1) * TP + 4 + bx + x] = (uint8_t)v;
    }
    add_pending = false;
    wave_sync();
  }
  if (add_pending) { // the block was predicted by a plane (mode 2) or by an earlier plane pass (mode 9): add on top of what is there
    if (lane < n * n) {
      uint8_t *px = tile + (by + y + 1) * TP + 4 + bx + x;
      *px = (uint8_t)mobi_add_clamp(*px, res[ri], fault);
    }
    wave_sync();
  }
}
} // namespace

enum { INTRA_LDS_WORDS = 136 + 72 + 72 + 384 + 384 + MOBI_SCALE_STRIDE };
// one intra macroblock by one wave; L = INTRA_LDS_WORDS words of LDS private to the wave; `it` only labels the profiling record
__device__ __forceinline__ void recon_intra_item(const MobiReconArgs &A, uint32_t *L, int clip, int mb, int lane, int it, bool wait_inter) {
  const MbDesc *desc = A.desc + (long)clip * A.n_mbs + mb;
  const uint32_t w1 = desc->w1, w3 = desc->w3;
  const uint32_t *rec = A.payload + desc->payload_off;
  const int32_t *sc_g = A.scale + ((w1 >> 20) & 63) * MOBI_SCALE_STRIDE;
  const int t8 = (w1 >> 14) & 0x3F, ncoef = desc->w2 & 0x3FF;
  const int S = A.stride;
  const Geo g{A.width, A.height, S, A.mbw, 31 - __builtin_clz((unsigned)S)};
  uint8_t *y0 = A.planes + (size_t)clip * A.clip_bytes + (size_t)(A.ring_base % 6) * A.slot_bytes;
  uint8_t *uv0 = y0 + (size_t)S * A.height;
  const int off = (mb / A.mbw) * 16 * S + (mb % A.mbw) * 16; // < 2^20

  // block records and the first 64 residual level words: in flight while the wave waits for its dependencies (a record fetched
  // inside the block loop would be one exposed round trip per block)
  const uint32_t myrec = lane < MOBI_INTRA_RECORDS ? rec[lane] : 0u;
  const uint32_t mycw = lane < ncoef ? rec[MOBI_INTRA_RECORDS + lane] : 0u;
  auto rec_at = [&](int i) { return (uint32_t)__builtin_amdgcn_readlane((int)myrec, i); };
  const int32_t sc_lo = sc_g[lane], sc_hi = lane < MOBI_SCALE_STRIDE - 64 ? sc_g[64 + lane] : 0; // dequant scales -> LDS

  const unsigned long long pt0 = A.prof ? __builtin_readcyclecounter() : 0;
  // All dependency levels of a frame step run in ONE launch (A.done != null): items are sorted by level, workgroups are
  // dispatched in order, and a wave waits here until the intra macroblocks its halo reads (MbDesc.w4..w7) carry this
  // step's tag.  Hand-off across CUs: producer stores pixels write-through (sc1), drains them, then publishes its
  // tag (sc1); the consumer polls the tag with sc1 loads and reads the halo with sc1 loads, so neither a stale L1
  // line nor a dirty L2 line can sit in between (MI355X_MICROARCH.md, inter-workgroup visibility).
  if (A.done) {
    if (lane < MOBI_INTRA_DEPS) {
      const uint32_t wv = (&desc->w4)[lane >> 1];
      const uint32_t dep = (wv >> (16 * (lane & 1))) & 0xFFFFu;
      // inter macroblocks only count when they run in this same launch (mobi_recon_step); a separate inter launch is complete
      if (dep != MOBI_DEP_NONE && (wait_inter || !(dep & MOBI_DEP_INTER))) {
        const uint32_t *f = A.done + (size_t)clip * A.n_mbs + (dep & 0x1FFFu);
        int spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != A.step_tag) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > (1 << 21)) { atomicOr(&A.fault[clip], 2); break; } // a producer that never ran: report, do not hang
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned long long pt1 = A.prof ? __builtin_readcyclecounter() : 0;

  uint8_t *ty = (uint8_t *)L;                 // 17 rows x TP
  uint8_t *tcu = (uint8_t *)(L + 136);        // 9 rows x TP
  uint8_t *tcv = (uint8_t *)(L + 136 + 72);
  int *coef = (int *)(L + 136 + 144), *tmp = coef + 384;
  int32_t *sc = (int32_t *)(tmp + 384);
  { // tiles (280 words) and coefficients (384 words): 664 words = 2656 B, as 16-byte stores
    const uint4 z = uint4{0, 0, 0, 0};
    uint4 *L4 = (uint4 *)L;
    L4[lane] = z;
    L4[64 + lane] = z;
    if (lane < 166 - 128) L4[128 + lane] = z;
  }
  sc[lane] = sc_lo;
  if (lane < MOBI_SCALE_STRIDE - 64) sc[64 + lane] = sc_hi;
  wave_sync();

  // ---- halo: real pixels only from raster-earlier macroblocks; the rest is the reference's fresh 0 ----
  const int mbx = mb % A.mbw, mby = mb / A.mbw;
  if (mbx >= 1 && mbx + 1 < A.mbw && mby >= 1) {
    // away from the picture's left, right and top edges ownership is known without arithmetic: the row above (left,
    // above, above-right macroblocks) and the column to the left are raster-earlier, everything to the right in the
    // macroblock's own rows is raster-later (reads the fresh plane's 0).  Two loads per lane instead of six.
    int p0 = -1, p1 = -1;
    uint32_t v[6] = {0, 0, 0, 0, 0, 0};
    int a0 = off, a1 = off / 2;
    if (lane < 25) { a0 = off - S + lane - 1; p0 = 4 + lane - 1; }                          // luma row -1, columns -1..23
    else if (lane < 41) { a0 = off + ((lane - 25) << g.lg) - 1; p0 = (lane - 25 + 1) * TP + 3; } // luma column -1
    {
      const int vv = lane >> 5, jl = lane & 31; // lanes 0..24 U, 32..56 V
      const int cb = off / 2 + vv * (S >> 1), tb = (136 + vv * 72) * 4;
      if (jl < 17) { a1 = cb - S + jl - 1; p1 = tb + 4 + jl - 1; }
      else if (jl < 25) { a1 = cb + ((jl - 17) << g.lg) - 1; p1 = tb + (jl - 17 + 1) * TP + 3; }
    }
    v[0] = ld_u8_sc1(y0, (uint32_t)a0);
    v[1] = ld_u8_sc1(uv0, (uint32_t)a1);
    ld_wait6(v);
    if (p0 >= 0) ty[p0] = (uint8_t)v[0];
    if (p1 >= 0) ty[p1] = (uint8_t)v[1];
  } else {
  // (all loads are issued before the first one is consumed: six dependent round trips otherwise)
  int hpos[6];
  uint32_t hval[6];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int i = lane + 64 * k;
    int r, c;
    if (i < 25) { r = -1; c = i - 1; }
    else if (i < 41) { r = i - 25; c = -1; }
    else { r = (i - 41) >> 3; c = 16 + ((i - 41) & 7); }
    const int a = off + (r << g.lg) + c;
    const int o = g.owner_luma(a);
    const bool take = i < 25 + 16 + 128 && o >= 0 && o < mb;
    hpos[k] = take ? (r + 1) * TP + 4 + c : -1;
    hval[k] = ld_u8_sc1(y0, (uint32_t)(take ? a : off)); // not ours to read: load our own first pixel instead, and drop it
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int i = lane + 64 * k;
    const int v = i >= 89, j = v ? i - 89 : i;
    int r, c;
    if (j < 17) { r = -1; c = j - 1; }
    else if (j < 25) { r = j - 17; c = -1; }
    else { r = (j - 25) >> 3; c = 8 + ((j - 25) & 7); }
    const int a = off / 2 + v * (S >> 1) + (r << g.lg) + c;
    const int o = g.owner_chroma(a);
    const bool take = i < 2 * (17 + 8 + 64) && o >= 0 && o < mb;
    hpos[3 + k] = take ? (136 + v * 72) * 4 + (r + 1) * TP + 4 + c : -1; // tcu / tcv follow the luma tile
    hval[3 + k] = ld_u8_sc1(uv0, (uint32_t)(take ? a : off / 2));
  }
  ld_wait6(hval);
#pragma unroll
  for (int k = 0; k < 6; k++)
    if (hpos[k] >= 0) ty[hpos[k]] = (uint8_t)hval[k];
  }
  if (lane < ncoef) scatter_one(sc, mycw, t8, coef);
  scatter_coefs(sc, rec + MOBI_INTRA_RECORDS, 64, ncoef, t8, coef, lane);
  wave_sync();
  // residuals of all coded areas at once, eight lanes per area (the block loop below only predicts and adds): coef -> tmp -> coef
  {
    const int a = lane >> 3, r = lane & 7;
    const bool act = a < 6 && ((w1 >> (8 + a)) & 1), is8a = (t8 >> a) & 1;
    if (act) idct_pass1(coef + 64 * a, tmp + 64 * a, is8a, r);
    wave_sync();
    if (act) idct_pass2_res(tmp + 64 * a, is8a, r, coef + 64 * a);
    wave_sync();
  }

  unsigned long long pt2 = 0;
  if (A.prof) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); pt2 = __builtin_readcyclecounter(); }
  // ---- block records, in decode order ----
  int fault = 0;
  if (w3 & 1) run_block(ty, 0, 0, 16, 2, (int16_t)(w3 >> 16), false, coef, 0, off, false, S, lane, &fault);
  for (int a = 0; a < 6; a++) {
    uint8_t *tile = a < 4 ? ty : (a == 4 ? tcu : tcv);
    const int ay = a < 4 ? (a >> 1) * 8 : 0, ax = a < 4 ? (a & 1) * 8 : 0;
    const int aoff = a < 4 ? off + ay * S + ax : off / 2 + (a - 4) * (S >> 1);
    const uint32_t r0 = rec_at(a * 4);
    const bool pre = (r0 >> 6) & 1;
    if (pre) run_block(tile, ay, ax, 8, 2, (int16_t)(r0 >> 16), false, coef, 0, aoff, a >= 4, S, lane, &fault);
    if (!((r0 >> 5) & 1)) {
      run_block(tile, ay, ax, 8, r0 & 15, pre ? 0 : (int16_t)(r0 >> 16), (r0 >> 4) & 1, coef + 64 * a, 0, aoff, a >= 4, S, lane, &fault);
    } else {
      for (int s = 0; s < 4; s++) {
        const uint32_t rr = rec_at(a * 4 + s);
        const int sy = (s >> 1) * 4, sx = (s & 1) * 4;
        const int param = (s == 0 && pre) ? 0 : (int16_t)(rr >> 16);
        run_block(tile, ay + sy, ax + sx, 4, rr & 15, param, (rr >> 4) & 1, coef + 64 * a, s, aoff + sy * S + sx, a >= 4, S, lane, &fault);
      }
    }
  }
  if (fault) atomicOr(&A.fault[clip], 1);

  unsigned long long pt3 = 0;
  if (A.prof) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); pt3 = __builtin_readcyclecounter(); }
  // ---- store interiors ----
  {
    const int row = lane >> 2, c4 = (lane & 3) * 4;
    __hip_atomic_store((uint32_t *)(y0 + off + row * S + c4), *(const uint32_t *)(ty + (row + 1) * TP + 4 + c4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane < 32) {
      const int v = lane >> 4, crow = (lane & 15) >> 1, cc4 = (lane & 1) * 4;
      __hip_atomic_store((uint32_t *)(uv0 + off / 2 + v * (S >> 1) + crow * S + cc4), *(const uint32_t *)((v ? tcv : tcu) + (crow + 1) * TP + 4 + cc4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (A.done) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the pixels have left this CU before the tag does
    if (lane == 0) __hip_atomic_store(A.done + (size_t)clip * A.n_mbs + mb, A.step_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (A.prof && lane == 0 && it < A.n_clips * A.n_mbs / 2) { // MOBI_DEBUG=9: dependency wait, loads, blocks, store+publish (shader clock)
    const unsigned long long pt4 = __builtin_readcyclecounter();
    ((uint4 *)A.prof)[(size_t)A.n_clips * A.n_mbs / 2 + it] = uint4{(uint32_t)(pt1 - pt0), (uint32_t)(pt2 - pt1), (uint32_t)(pt3 - pt2), (uint32_t)(pt4 - pt3)};
  }
}

extern "C" __global__ __launch_bounds__(64 * IWAVES) void mobi_recon_intra(MobiReconArgs A, const uint32_t *items, int n_items) {
  __shared__ uint32_t lds[IWAVES][INTRA_LDS_WORDS];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int it = blockIdx.x * IWAVES + wave;
  if (it >= n_items) return;
  const uint32_t item = items[it];
  recon_intra_item(A, lds[wave], (int)(item >> 13), (int)(item & 0x1FFF), lane, it, false);
}

// Items as the device-side parser leaves them (mobi_dparse.hip): per clip, raster order, n_intra[clip] of them at a stride
// of n_mbs.  Workgroup it = slot * n_clips + clip: neighbours in the dispatch order belong to different clips, so every
// clip advances along its own dependency chain at the same time, and what a wave waits for (raster-earlier, same clip)
// always sits in an earlier slot, i.e. was dispatched before it.
extern "C" __global__ __launch_bounds__(64) void mobi_recon_intra_cl(MobiReconArgs A, const uint32_t *items, const uint32_t *n_intra, uint32_t n_intra_stride,
                                                                      uint32_t magic_n_clips) {
  __shared__ uint32_t lds[INTRA_LDS_WORDS];
  const int lane = threadIdx.x;
  uint32_t clip;
  const uint32_t slot = fastdiv(blockIdx.x, (uint32_t)A.n_clips, magic_n_clips, clip);
  if (slot >= n_intra[(size_t)clip * n_intra_stride]) return;
  const uint32_t item = items[(size_t)clip * A.n_mbs + slot];
  recon_intra_item(A, lds, (int)clip, (int)(item & 0x1FFF), lane, (int)blockIdx.x, false);
}

// =====================================================================================================
// mobi_recon_step: a whole frame step -- every inter quad and every intra macroblock of every clip -- in ONE launch
// =====================================================================================================
// One wave per workgroup.  Workgroup b runs on XCD b & 7 (observed dispatch order, used for speed and for the
// order of arrival only -- correctness rests on the completion tags).  Each XCD owns a contiguous range of clips
// and walks it in segments: segment s = the quads of its clip s, then the intra macroblocks (sorted by dependency
// level, padded to K per clip) of its clip s - STEP_LAG, whose inter neighbours have been dispatched a while ago.
// An intra wave waits until every raster-earlier macroblock its halo reads carries this step's tag (done[]); quads
// publish theirs after write-through stores.  Workgroups are dispatched in index order, so whatever a wave waits for
// was dispatched before it: waiting cannot deadlock (and a bounded spin reports instead of hanging if that ever fails).
enum { STEP_LAG = 2 };
struct MobiStepArgs {
  const uint32_t *items; // [clip][K] macroblock index or 0xFFFFFFFF
  uint32_t K, seg, magic_seg, clips_per_xcd;
};
template <bool PROF>
__device__ __forceinline__ void recon_step_entry(const MobiReconArgs &A, const MobiStepArgs &T) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[Q_BYTES > INTRA_LDS_WORDS * 4 ? Q_BYTES : INTRA_LDS_WORDS * 4];
  const int lane = threadIdx.x;
  const uint32_t x = blockIdx.x & 7, v = blockIdx.x >> 3;
  uint32_t pos;
  const uint32_t sgm = fastdiv(v, T.seg, T.magic_seg, pos);
  if (pos < A.qpc) {
    if (sgm >= T.clips_per_xcd) return;
    const uint32_t clip = x * T.clips_per_xcd + sgm;
    if (clip >= (uint32_t)A.n_clips) return;
    recon_inter_quad<PROF>(A, lds, clip * A.qpc + pos, lane);
  } else {
    if (sgm < STEP_LAG || sgm - STEP_LAG >= T.clips_per_xcd) return;
    const uint32_t clip = x * T.clips_per_xcd + (sgm - STEP_LAG);
    if (clip >= (uint32_t)A.n_clips) return;
    const uint32_t slot = pos - A.qpc;
    const uint32_t mb = T.items[(size_t)clip * T.K + slot];
    if (mb == 0xFFFFFFFFu) return;
    __builtin_amdgcn_s_setprio(3); // few, long and latency-bound: let them through ahead of the VALU-bound quads
    recon_intra_item(A, (uint32_t *)lds, (int)clip, (int)mb, lane, (int)(clip * T.K + slot), true);
  }
}
extern "C" __global__ __launch_bounds__(64) void mobi_recon_step(MobiReconArgs A, MobiStepArgs T) { recon_step_entry<false>(A, T); }
extern "C" __global__ __launch_bounds__(64) void mobi_recon_step_prof(MobiReconArgs A, MobiStepArgs T) { recon_step_entry<true>(A, T); }

// =====================================================================================================
// launch wrappers (called from mobi_abi.cpp)
// =====================================================================================================
extern "C" int mobi_launch_inter(const MobiReconArgs *a, int oct, hipStream_t s) {
  const long quads = (long)a->qpc * a->n_clips;
  if (quads <= 0) return 0;
  if (a->slot_bytes >= (1u << 24)) return (int)hipErrorInvalidValue; // 24-bit multiply in the kernel
  const unsigned grid = (unsigned)(((quads + INTER_WAVES - 1) / INTER_WAVES + 7) / 8 * 8); // whole number of workgroups per XCD
  MobiReconArgs b = *a;
  b.inter_per_xcd = grid / 8;
  static const int lds_pad = getenv("MOBI_LDS_PAD") ? atoi(getenv("MOBI_LDS_PAD")) : 0; // experiment: extra LDS per workgroup lowers occupancy
  if (oct && !b.done) { // eight macroblocks per wave: the q* fields count octets for this kernel
    b.qpr = ((uint32_t)b.mbw + 7) / 8;
    b.qpc = b.qpr * (uint32_t)(b.n_mbs / b.mbw);
    auto magic = [](uint32_t d) { uint64_t m = ((uint64_t)1 << 32) / d; return (uint32_t)(m > 0xFFFFFFFFull ? 0xFFFFFFFFull : m); };
    b.magic_qpr = magic(b.qpr);
    b.magic_qpc = magic(b.qpc);
    const unsigned g8 = (unsigned)(((long)b.qpc * b.n_clips + 7) / 8 * 8);
    b.inter_per_xcd = g8 / 8;
    if (oct == 2) hipLaunchKernelGGL(mobi_recon_inter8_r1, dim3(g8), dim3(64), lds_pad, s, b); // r01's octet kernel, kept for A/B runs
    else if (b.prof) hipLaunchKernelGGL(mobi_recon_inter8_prof, dim3(g8), dim3(64), lds_pad, s, b);
    else {
      static const int variant = getenv("MOBI_OCT_VARIANT") ? atoi(getenv("MOBI_OCT_VARIANT")) : 0;
      if (variant == 1) hipLaunchKernelGGL(mobi_recon_inter8_w4, dim3(g8), dim3(64), lds_pad, s, b);
      else hipLaunchKernelGGL(mobi_recon_inter8, dim3(g8), dim3(64), lds_pad, s, b);
    }
    return (int)hipGetLastError();
  }
  if (b.prof) hipLaunchKernelGGL(mobi_recon_inter_prof, dim3(grid), dim3(64 * INTER_WAVES), lds_pad, s, b);
  else hipLaunchKernelGGL(mobi_recon_inter, dim3(grid), dim3(64 * INTER_WAVES), lds_pad, s, b);
  return (int)hipGetLastError();
}
extern "C" int mobi_launch_intra(const MobiReconArgs *a, const uint32_t *items_dev, int n_items, hipStream_t s) {
  if (n_items <= 0) return 0;
  const unsigned grid = (unsigned)((n_items + IWAVES - 1) / IWAVES);
  hipLaunchKernelGGL(mobi_recon_intra, dim3(grid), dim3(64 * IWAVES), 0, s, *a, items_dev, n_items);
  return (int)hipGetLastError();
}
extern "C" int mobi_launch_intra_cl(const MobiReconArgs *a, const uint32_t *items_dev, const uint32_t *n_intra_dev, int n_intra_stride_words, int K, hipStream_t s) {
  if (K <= 0 || a->n_clips <= 0) return 0;
  const uint64_t m = ((uint64_t)1 << 32) / (uint32_t)a->n_clips;
  hipLaunchKernelGGL(mobi_recon_intra_cl, dim3((unsigned)K * (unsigned)a->n_clips), dim3(64), 0, s, *a, items_dev, n_intra_dev, (uint32_t)n_intra_stride_words,
                     (uint32_t)(m > 0xFFFFFFFFull ? 0xFFFFFFFFull : m));
  return (int)hipGetLastError();
}
extern "C" int mobi_launch_step(const MobiReconArgs *a, const uint32_t *items_dev, int K, hipStream_t s) {
  if (a->slot_bytes >= (1u << 24) || !a->done || a->n_clips <= 0) return (int)hipErrorInvalidValue;
  MobiStepArgs t;
  t.items = items_dev;
  t.K = (uint32_t)K;
  t.seg = a->qpc + (uint32_t)K;
  const uint64_t m = ((uint64_t)1 << 32) / t.seg;
  t.magic_seg = (uint32_t)(m > 0xFFFFFFFFull ? 0xFFFFFFFFull : m);
  t.clips_per_xcd = ((uint32_t)a->n_clips + 7) / 8;
  const unsigned grid = 8u * (t.clips_per_xcd + STEP_LAG) * t.seg;
  if (a->prof) hipLaunchKernelGGL(mobi_recon_step_prof, dim3(grid), dim3(64), 0, s, *a, t);
  else hipLaunchKernelGGL(mobi_recon_step, dim3(grid), dim3(64), 0, s, *a, t);
  return (int)hipGetLastError();
}
