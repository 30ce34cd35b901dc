// mobi_kernels.hip -- gfx950 (MI355X / CDNA4) reconstruction kernels: one wavefront per macroblock.
//
//   mobi_recon_inter : P-frame inter macroblocks.  Per MB: half-pel truncating motion compensation
//                      of every partition leaf from the reference planes (CopyBlock, MD.cs:418-456),
//                      dequant + 8x8/4x4 integer inverse transforms + clamp-add of the residual
//                      (MD.cs:3424-3429, :3435-3798), coalesced dword stores of Y/U/V.
//   mobi_recon_intra : intra macroblocks of one dependency level (I-frames and codes 6/7 inside
//                      P-frames): halo load with raster-order availability masking, predictors
//                      (MD.cs:1883-2774, :3017-3327) and residuals in decode order inside LDS.
//
// 8-bit pel work is HBM-bound: no MFMA.  64-wide wavefronts: a 16x16 luma block is 64 lanes x 4 px
// (one dword per lane per row segment); an 8x8 block is 64 lanes x 1 px for prediction and 8 lanes x
// 8-point butterflies for the transform, with the transpose staged through LDS.
// LDS use is per wave (no workgroup barriers): waves never share data, so a wavefront-scope fence
// (a pure compiler barrier -- LDS is FIFO per wave) is all that separates producer and consumer lanes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mobi_cmd.h"
#include "mobi_kernels.h"
#include "mobi_recon_math.h"

namespace {

enum { TP = 32 };                     // intra tile pitch: interior col c at byte 4+c, halo col -1 at byte 3
enum { HALO_Y_RIGHT = 23, HALO_C_RIGHT = 15 }; // must match MOBI_HALO_* in mobi_parse.h
enum { WAVES = 4 };

__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

struct Geo { // stride is 256/512/1024 (MD.cs:50-52): divide/modulo by shifts
  int width, height, stride, mbw, lg;
  __device__ __forceinline__ int owner_luma(long a) const {
    if (a < 0) return -1;
    const int row = (int)(a >> lg), col = (int)(a & (stride - 1));
    if (col >= width || row >= height) return -1;
    return (row >> 4) * mbw + (col >> 4);
  }
  __device__ __forceinline__ int owner_chroma(long a) const {
    if (a < 0) return -1;
    const int row = (int)(a >> lg), col = (int)(a & (stride - 1));
    const int x = col >= (stride >> 1) ? col - (stride >> 1) : col;
    if (x >= (width >> 1) || row >= (height >> 1)) return -1;
    return (row >> 3) * mbw + (x >> 3);
  }
};

// four reference pixels (+1 for the half-pel neighbour) starting at an arbitrary byte address,
// fetched as two aligned dwords and funnel-shifted
__device__ __forceinline__ void load5(const uint8_t *p, uint32_t &a, uint32_t &b) {
  const uintptr_t ad = (uintptr_t)p;
  const uint32_t *q = (const uint32_t *)(ad & ~(uintptr_t)3);
  const uint32_t sh = (uint32_t)(ad & 3) * 8;
  const uint64_t v = (((uint64_t)q[1] << 32) | q[0]) >> sh;
  a = (uint32_t)v;
  b = (uint32_t)(v >> 8);
}
__device__ __forceinline__ uint32_t mc_word(const uint8_t *p, int stride, int phase) {
  uint32_t a, b, c = 0, d = 0;
  load5(p, a, b);
  if (phase & 2) load5(p + stride, c, d);
  return mobi_mc4(a, b, c, d, phase);
}
// byte mask of the pixels [c4, c4+4) that fall inside [lo, lo+len)
__device__ __forceinline__ uint32_t seg_mask(int c4, int lo, int len) {
  uint32_t m = 0;
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (c4 + k >= lo && c4 + k < lo + len) m |= 0xFFu << (8 * k);
  return m;
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
// pass 1 of area b by lane r (0..7): 8x8 -> coefficient group r; 4x4 -> sub-block r>>1, groups (r&1)*2+{0,1}
__device__ __forceinline__ void idct_pass1(const int *c, int *t, bool is8, int r) {
  int in[8], out[8];
  if (is8) {
#pragma unroll
    for (int m = 0; m < 8; m++) in[m] = c[8 * r + m];
    if (r == 0) in[0] += 32;
    mobi_bfly8(in, out);
#pragma unroll
    for (int m = 0; m < 8; m++) t[8 * m + r] = out[m];
  } else {
    const int s = r >> 1;
#pragma unroll
    for (int g = 0; g < 2; g++) {
      const int k = (r & 1) * 2 + g;
#pragma unroll
      for (int m = 0; m < 4; m++) in[m] = c[16 * s + 4 * k + m];
      if (k == 0) in[0] += 32;
      mobi_bfly4(in, out);
#pragma unroll
      for (int m = 0; m < 4; m++) t[16 * s + 4 * m + k] = out[m];
    }
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

// One wavefront per macroblock, four macroblocks (64 x 16 luma pixels) per workgroup, no workgroup barriers:
// a wave lives for one macroblock, so the hardware scheduler does the load balancing and latency hiding.
// XCD-aware order: the dispatcher deals consecutive workgroups round-robin to the 8 XCDs, so XCD x gets one
// contiguous run of macroblocks (whole clips) -- neighbours that share output lines and overlapping MC
// windows then meet in ONE L2 while they are in flight together.
extern "C" __global__ __launch_bounds__(256) void mobi_recon_inter(MobiReconArgs A) {
  __shared__ uint32_t lds[WAVES][96 + 384 + 384]; // per wave: pred tiles (384 B), coef, tmp
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t total = (uint32_t)A.n_clips * (uint32_t)A.n_mbs;
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t gm = ((blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)) * WAVES + wave;
  if (gm >= total) return;
  const uint4 d = *(const uint4 *)(A.desc + gm);
  const uint32_t w1 = d.y;
  if ((w1 & 1) != MOBI_MB_INTER) return;
  if (A.debug == 1) return; // profiling ablation (env MOBI_DEBUG): launch + descriptor only
  uint32_t mb, mbx;
  const uint32_t clip = fastdiv(gm, (uint32_t)A.n_mbs, A.magic_n_mbs, mb);
  const uint32_t mby = fastdiv(mb, (uint32_t)A.mbw, A.magic_mbw, mbx);
  const int nl = (w1 >> 1) & 0x7F, cbp6 = (w1 >> 8) & 0x3F, t8 = (w1 >> 14) & 0x3F, ncoef = d.z & 0x3FF;
  const uint32_t *pl = A.payload + d.x;
  const uint32_t *cw = pl + (nl > 1 ? MOBI_MV_CELLS : 0); // residual levels follow the MV cell map
  const int S = A.stride;
  const uint32_t ysz = (uint32_t)S * (uint32_t)A.height;
  uint8_t *clip_base = A.planes + (size_t)clip * A.clip_bytes;
  const int off = (int)(mby * 16 * (uint32_t)S + mbx * 16);
  const int yrow = lane >> 2, yc4 = (lane & 3) * 4;
  const int cv = (lane >> 4) & 1, crow = (lane & 15) >> 1, cc4 = (lane & 1) * 4;
  // first 64 residual levels travel together with the pixel fetches
  const uint32_t c_first = (cbp6 && lane < ncoef) ? cw[lane] : 0;

  // ---- motion compensation: lane -> luma row lane>>2, px (lane&3)*4 ; lanes 0..31 -> chroma ----
  uint32_t ypred = 0, cpred = 0;
  if (nl == 1) { // one 16x16 leaf, inline in the descriptor
    const int ref = (d.z >> 22) & 7;
    const int dx = (int16_t)(d.w & 0xFFFF), dy = (int16_t)(d.w >> 16);
    const uint8_t *ry = clip_base + (uint32_t)((A.ring_base + 6 - ref) % 6) * A.slot_bytes;
    ypred = mc_word(ry + (off + (yrow + (dy >> 1)) * S + yc4 + (dx >> 1)), S, (dx & 1) | ((dy & 1) << 1));
    const int cdx = dx >> 1, cdy = dy >> 1;
    if (lane < 32)
      cpred = mc_word(ry + ysz + ((off >> 1) + cv * (S >> 1) + (crow + (cdy >> 1)) * S + cc4 + (cdx >> 1)), S, (cdx & 1) | ((cdy & 1) << 1));
  } else { // MV cell map: every lane looks up the cells under its own pixels, then all fetches fly together
    const uint2 yc = *(const uint2 *)(pl + (yrow >> 1) * 8 + (yc4 >> 1));
    uint4 cc = uint4{0, 0, 0, 0};
    if (lane < 32) cc = *(const uint4 *)(pl + crow * 8 + cc4);
    const int dxa = mobi_cell_dx(yc.x), dya = mobi_cell_dy(yc.x), dxb = mobi_cell_dx(yc.y), dyb = mobi_cell_dy(yc.y);
    const uint8_t *ra = clip_base + (uint32_t)((A.ring_base + 6 - mobi_cell_ref(yc.x)) % 6) * A.slot_bytes;
    const uint8_t *rb = clip_base + (uint32_t)((A.ring_base + 6 - mobi_cell_ref(yc.y)) % 6) * A.slot_bytes;
    const uint32_t va = mc_word(ra + (off + (yrow + (dya >> 1)) * S + yc4 + (dxa >> 1)), S, (dxa & 1) | ((dya & 1) << 1));
    const uint32_t vb = mc_word(rb + (off + (yrow + (dyb >> 1)) * S + yc4 + (dxb >> 1)), S, (dxb & 1) | ((dyb & 1) << 1));
    ypred = (va & 0x0000FFFFu) | (vb & 0xFFFF0000u);
    if (lane < 32) {
      const int cbase = (off >> 1) + cv * (S >> 1) + crow * S + cc4;
      const uint32_t cell[4] = {cc.x, cc.y, cc.z, cc.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int cdx = mobi_cell_dx(cell[k]) >> 1, cdy = mobi_cell_dy(cell[k]) >> 1;
        const uint8_t *rc = clip_base + (uint32_t)((A.ring_base + 6 - mobi_cell_ref(cell[k])) % 6) * A.slot_bytes + ysz;
        const uint32_t v = mc_word(rc + (cbase + (cdy >> 1) * S + (cdx >> 1)), S, (cdx & 1) | ((cdy & 1) << 1));
        cpred |= v & (0xFFu << (8 * k));
      }
    }
  }

  // ---- residual ----
  if (cbp6) {
    uint32_t *L = lds[wave];
    uint8_t *ty = (uint8_t *)L;          // 16 x 16
    uint8_t *tc = ty + 256;              // U 8x8 then V 8x8
    int *coef = (int *)(L + 96), *tmp = coef + 384;
    const int32_t *sc = A.scale + ((w1 >> 20) & 63) * MOBI_SCALE_STRIDE;
    ((uint32_t *)ty)[yrow * 4 + (lane & 3)] = ypred;
    if (lane < 32) ((uint32_t *)tc)[cv * 16 + crow * 2 + (lane & 1)] = cpred;
    zero_coefs(coef, lane);
    wave_sync();
    if (lane < ncoef) scatter_one(sc, c_first, t8, coef);
    scatter_coefs(sc, cw, 64, ncoef, t8, coef, lane);
    wave_sync();
    const int b = lane >> 3, r = lane & 7;
    const bool act = lane < 48 && ((cbp6 >> b) & 1);
    if (act) idct_pass1(coef + 64 * b, tmp + 64 * b, (t8 >> b) & 1, r);
    wave_sync();
    int fault = 0;
    if (act) {
      uint8_t *px = b < 4 ? ty + (b >> 1) * 8 * 16 + (b & 1) * 8 : tc + (b - 4) * 64;
      idct_pass2(tmp + 64 * b, (t8 >> b) & 1, r, px, b < 4 ? 16 : 8, 0xF, &fault);
    }
    wave_sync();
    if (fault) atomicOr(&A.fault[clip], 1);
    ypred = ((uint32_t *)ty)[yrow * 4 + (lane & 3)];
    if (lane < 32) cpred = ((uint32_t *)tc)[cv * 16 + crow * 2 + (lane & 1)];
  }

  // ---- store: 16 B per row per MB (4 adjacent MBs per workgroup -> 64 B runs) ----
  if (A.debug == 2) return; // profiling ablation: no stores
  uint8_t *y0 = clip_base + (uint32_t)(A.ring_base % 6) * A.slot_bytes;
  *(uint32_t *)(y0 + (off + yrow * S + yc4)) = ypred;
  if (lane < 32) *(uint32_t *)(y0 + ysz + ((off >> 1) + cv * (S >> 1) + crow * S + cc4)) = cpred;
}

// =====================================================================================================
// intra macroblocks of one dependency level
// =====================================================================================================
namespace {
struct TileNb {
  const uint8_t *t;
  int by, bx;
  __device__ __forceinline__ int operator()(int dy, int dx) const { return t[(by + dy + 1) * TP + 4 + bx + dx]; }
};
// predict one block on the tile (all lanes call; lanes >= n*n idle), then its residual when coded
__device__ __forceinline__ void run_block(uint8_t *tile, int by, int bx, int n, int mode, int param, bool coded,
                                          const int *coef, int *tmp, bool is8, int sub, long block_off, bool is_uv,
                                          int S, int lane, int *fault) {
  TileNb nb{tile, by, bx};
  if (mode == 2) {
    const int wpr = n >> 2, nw = n * wpr; // words per row, words in block
    if (lane < nw) {
      const int y = lane / wpr, x0 = (lane % wpr) * 4;
      const uint32_t w = mobi_plane_word(n, param, y, x0, nb);
      *(uint32_t *)(tile + (by + y + 1) * TP + 4 + bx + x0) = w;
    }
    wave_sync();
  } else if (mode != 9) {
    const int vfix = is_uv && (block_off & (S - 1)) >= (S >> 1);                                      // MD.cs:1886
    const int left_avail = ((block_off - (vfix ? (S >> 1) : 0)) & (S - 1)) != 0, top_avail = block_off >= S; // :1923-1924
    if (lane < n * n) {
      const int y = (n == 8) ? lane >> 3 : lane >> 2, x = (n == 8) ? lane & 7 : lane & 3;
      const int v = mobi_pred_px(mode, n, y, x, top_avail, left_avail, nb);
      tile[(by + y + 1) * TP + 4 + bx + x] = (uint8_t)v;
    }
    wave_sync();
  }
  if (coded) {
    uint8_t *area_px = tile + (by + 1) * TP + 4 + bx - (is8 ? 0 : ((sub >> 1) * 4 * TP + (sub & 1) * 4));
    const int r0 = is8 ? 0 : sub * 2, r1 = is8 ? 8 : sub * 2 + 2; // lanes (as pass rows r) that take part
    if (lane >= r0 && lane < r1) idct_pass1(coef, tmp, is8, lane);
    wave_sync();
    if (lane >= r0 && lane < r1) idct_pass2(tmp, is8, lane, area_px, TP, 1 << sub, fault);
    wave_sync();
  }
}
} // namespace

extern "C" __global__ __launch_bounds__(256) void mobi_recon_intra(MobiReconArgs A, const uint32_t *items, int n_items) {
  __shared__ uint32_t lds[WAVES][136 + 72 + 72 + 384 + 384];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int it = blockIdx.x * WAVES + wave;
  if (it >= n_items) return;
  const uint32_t item = items[it];
  const int clip = (int)(item >> 13), mb = (int)(item & 0x1FFF);
  const MbDesc *desc = A.desc + (long)clip * A.n_mbs + mb;
  const uint32_t w1 = desc->w1, w3 = desc->w3;
  const uint32_t *rec = A.payload + desc->payload_off;
  const int32_t *sc = A.scale + ((w1 >> 20) & 63) * MOBI_SCALE_STRIDE;
  const int t8 = (w1 >> 14) & 0x3F, ncoef = desc->w2 & 0x3FF;
  const int S = A.stride;
  const Geo g{A.width, A.height, S, A.mbw, 31 - __builtin_clz((unsigned)S)};
  uint8_t *y0 = A.planes + (size_t)clip * A.clip_bytes + (size_t)(A.ring_base % 6) * A.slot_bytes;
  uint8_t *uv0 = y0 + (size_t)S * A.height;
  const long off = (long)(mb / A.mbw) * 16 * S + (mb % A.mbw) * 16;

  uint32_t *L = lds[wave];
  uint8_t *ty = (uint8_t *)L;                 // 17 rows x TP
  uint8_t *tcu = (uint8_t *)(L + 136);        // 9 rows x TP
  uint8_t *tcv = (uint8_t *)(L + 136 + 72);
  int *coef = (int *)(L + 136 + 144), *tmp = coef + 384;
  for (int i = lane; i < 136 + 144; i += 64) L[i] = 0;
  zero_coefs(coef, lane);
  wave_sync();

  // ---- halo: real pixels only from raster-earlier macroblocks; the rest is the reference's fresh 0 ----
  for (int i = lane; i < 25 + 16 + 128; i += 64) {
    int r, c;
    if (i < 25) { r = -1; c = i - 1; }
    else if (i < 41) { r = i - 25; c = -1; }
    else { r = (i - 41) >> 3; c = 16 + ((i - 41) & 7); }
    const long a = off + (long)r * S + c;
    const int o = g.owner_luma(a);
    if (o >= 0 && o < mb) ty[(r + 1) * TP + 4 + c] = y0[a];
  }
  for (int i = lane; i < 2 * (17 + 8 + 64); i += 64) {
    const int v = i >= 89, j = v ? i - 89 : i;
    int r, c;
    if (j < 17) { r = -1; c = j - 1; }
    else if (j < 25) { r = j - 17; c = -1; }
    else { r = (j - 25) >> 3; c = 8 + ((j - 25) & 7); }
    const long a = off / 2 + v * (S >> 1) + (long)r * S + c;
    const int o = g.owner_chroma(a);
    if (o >= 0 && o < mb) (v ? tcv : tcu)[(r + 1) * TP + 4 + c] = uv0[a];
  }
  scatter_coefs(sc, rec + MOBI_INTRA_RECORDS, 0, ncoef, t8, coef, lane);
  wave_sync();

  // ---- block records, in decode order ----
  int fault = 0;
  if (w3 & 1) run_block(ty, 0, 0, 16, 2, (int16_t)(w3 >> 16), false, coef, tmp, false, 0, off, false, S, lane, &fault);
  for (int a = 0; a < 6; a++) {
    uint8_t *tile = a < 4 ? ty : (a == 4 ? tcu : tcv);
    const int ay = a < 4 ? (a >> 1) * 8 : 0, ax = a < 4 ? (a & 1) * 8 : 0;
    const long aoff = a < 4 ? off + (long)ay * S + ax : off / 2 + (a - 4) * (S >> 1);
    const uint32_t r0 = rec[a * 4];
    const bool pre = (r0 >> 6) & 1;
    if (pre) run_block(tile, ay, ax, 8, 2, (int16_t)(r0 >> 16), false, coef, tmp, false, 0, aoff, a >= 4, S, lane, &fault);
    if (!((r0 >> 5) & 1)) {
      run_block(tile, ay, ax, 8, r0 & 15, pre ? 0 : (int16_t)(r0 >> 16), (r0 >> 4) & 1, coef + 64 * a, tmp, true, 0, aoff, a >= 4, S, lane, &fault);
    } else {
      for (int s = 0; s < 4; s++) {
        const uint32_t rr = rec[a * 4 + s];
        const int sy = (s >> 1) * 4, sx = (s & 1) * 4;
        const int param = (s == 0 && pre) ? 0 : (int16_t)(rr >> 16);
        run_block(tile, ay + sy, ax + sx, 4, rr & 15, param, (rr >> 4) & 1, coef + 64 * a, tmp, false, s, aoff + (long)sy * S + sx, a >= 4, S, lane, &fault);
      }
    }
  }
  if (fault) atomicOr(&A.fault[clip], 1);

  // ---- store interiors ----
  {
    const int row = lane >> 2, c4 = (lane & 3) * 4;
    *(uint32_t *)(y0 + off + (long)row * S + c4) = *(const uint32_t *)(ty + (row + 1) * TP + 4 + c4);
    if (lane < 32) {
      const int v = lane >> 4, crow = (lane & 15) >> 1, cc4 = (lane & 1) * 4;
      *(uint32_t *)(uv0 + off / 2 + v * (S >> 1) + (long)crow * S + cc4) = *(const uint32_t *)((v ? tcv : tcu) + (crow + 1) * TP + 4 + cc4);
    }
  }
}

// =====================================================================================================
// launch wrappers (called from mobi_abi.cpp)
// =====================================================================================================
extern "C" int mobi_launch_inter(const MobiReconArgs *a, hipStream_t s) {
  const long waves = (long)a->n_clips * a->n_mbs;
  if (waves <= 0) return 0;
  const unsigned grid = (unsigned)(((waves + WAVES - 1) / WAVES + 7) / 8 * 8); // whole number of workgroups per XCD
  hipLaunchKernelGGL(mobi_recon_inter, dim3(grid), dim3(64 * WAVES), 0, s, *a);
  return (int)hipGetLastError();
}
extern "C" int mobi_launch_intra(const MobiReconArgs *a, const uint32_t *items_dev, int n_items, hipStream_t s) {
  if (n_items <= 0) return 0;
  const unsigned grid = (unsigned)((n_items + WAVES - 1) / WAVES);
  hipLaunchKernelGGL(mobi_recon_intra, dim3(grid), dim3(64 * WAVES), 0, s, *a, items_dev, n_items);
  return (int)hipGetLastError();
}
