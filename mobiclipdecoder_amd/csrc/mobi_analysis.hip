// mobi_analysis.hip -- encoder-side motion analysis on gfx950 (the search half).
//
// Analyzer.InterPredict2x2 (Analyzer.cs:608-681) for all 64 2x2 luma blocks of a macroblock at once
// (SolveInterPredictionPuzzle, :683-693): a three-step search (6, 3, 1 full pels; nine candidates per step, ties to the
// shorter vector) in each of up to five past frames, then the best frame (ties again to the shorter vector).  The
// encoder's PastFramesY[i] (MobiEncoder.cs:138-144) are the reconstruction ring this library already keeps in HBM, so the
// search runs where the frames are.  One wavefront per macroblock, one lane per 2x2 block; per past frame the 36 x 40 byte
// search window of the macroblock is staged in LDS once and all 27 candidates of all 64 blocks read it from there.
#include <hip/hip_runtime.h>

#include "mobi_kernels.h"
#include "mobi_tile.h"

namespace {
enum { WIN_ROWS = 36, WIN_DW = 10, REACH = 10 }; // 6 + 3 + 1 pels either way around a 16 x 16 macroblock; 40 bytes per row from column BX - 12
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
} // namespace

extern "C" __global__ __launch_bounds__(64) void mobi_motion_search_2x2(MobiReconArgs A, const uint8_t *src, uint32_t *out, int n_past) {
  __shared__ uint32_t win[WIN_ROWS * WIN_DW];
  const int lane = threadIdx.x, X = lane & 7, Y = lane >> 3;
  uint32_t mb;
  const uint32_t clip = blockIdx.x / (uint32_t)A.n_mbs;
  mb = blockIdx.x - clip * (uint32_t)A.n_mbs;
  const int mby = (int)(mb / (uint32_t)A.mbw), mbx = (int)mb - mby * A.mbw;
  const int W = A.width, H = A.height, S = A.stride, BX = mbx * 16, BY = mby * 16, lgS = 31 - __builtin_clz((unsigned)S);
  const int bx = BX + 2 * X, by = BY + 2 * Y; // this lane's block
  const uint8_t *c0 = src + (size_t)clip * W * H + (size_t)by * W + bx;
  const uint32_t c01 = *(const uint16_t *)c0, c23 = *(const uint16_t *)(c0 + W); // cmp[0..3], Encoder/MacroBlock.cs:80-83
  const int cmp0 = c01 & 0xFF, cmp1 = c01 >> 8, cmp2 = c23 & 0xFF, cmp3 = c23 >> 8;
  const uint8_t *wb = (const uint8_t *)win;
  int rdx = 0, rdy = 0, rframe = 0, resultscore = 0x7FFFFFFF;
  for (int i = 0; i < n_past; i++) { // Analyzer.cs:616
    int sl = A.ring_base - i;
    sl = sl < 0 ? sl + 6 : sl;
    const uint8_t *plane = A.planes + (size_t)clip * A.clip_bytes + (size_t)sl * A.slot_bytes;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the previous frame's reads are done
    for (int idx = lane; idx < WIN_ROWS * WIN_DW; idx += 64) {
      const int r = idx / WIN_DW, dw = idx - r * WIN_DW, row = BY - REACH + r, col = BX - 12 + 4 * dw;
      uint32_t v = 0;
      if (row >= 0 && row < H && col >= 0 && col < S) v = *(const uint32_t *)(plane + mobi_ty((uint32_t)(row * S + col), lgS)); // tiled planes (mobi_tile.h)
      win[idx] = v; // bytes outside the picture are never used: the bounds tests below skip those candidates (:630, :633)
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    int centerx = 0, centery = 0, centerscore = 0;
#pragma unroll 1
    for (int St = 6; St >= 1; St >>= 1) { // 6, 3, 1 (:619, :661)
      int bestscore = 0x7FFFFFFF, bestx = 0, besty = 0;
#pragma unroll
      for (int yi = -1; yi <= 1; yi++) {
        const int ny = yi * St + centery;
        const bool yok = !(by + ny < 0 || by + 2 + ny > H); // :630
#pragma unroll
        for (int xi = -1; xi <= 1; xi++) {
          const int nx = xi * St + centerx;
          const bool ok = yok && !(bx + nx < 0 || bx + 2 + nx > W); // :633
          const int o = (by + ny - (BY - REACH)) * (WIN_DW * 4) + (bx + nx - (BX - 12));
          const int oo = ok ? o : 0;
          // v_sad_u8 on single bytes: |a - b| + accumulator in one instruction
          const int score = (int)__builtin_amdgcn_sad_u8((uint32_t)cmp0, wb[oo], __builtin_amdgcn_sad_u8((uint32_t)cmp1, wb[oo + 1],
                            __builtin_amdgcn_sad_u8((uint32_t)cmp2, wb[oo + WIN_DW * 4], __builtin_amdgcn_sad_u8((uint32_t)cmp3, wb[oo + WIN_DW * 4 + 1], 0u))));
          if (ok && (score < bestscore || (score == bestscore && iabs(nx) + iabs(ny) < iabs(bestx) + iabs(besty)))) { // :646-651
            bestx = nx;
            besty = ny;
            bestscore = score;
          }
        }
      }
      centerx = bestx;
      centery = besty;
      centerscore = bestscore;
    }
    const int cx2 = centerx * 2, cy2 = centery * 2;
    if (centerscore < resultscore || (centerscore == resultscore && iabs(cx2) + iabs(cy2) < iabs(rdx) + iabs(rdy))) { // :666-671
      rdx = cx2;
      rdy = cy2;
      rframe = i;
      resultscore = centerscore;
    }
  }
  const uint32_t sc = resultscore == 0x7FFFFFFF ? 0xFFFu : (uint32_t)resultscore;
  out[(size_t)blockIdx.x * 64 + lane] = ((uint32_t)rdx & 0xFFu) | (((uint32_t)rdy & 0xFFu) << 8) | ((uint32_t)rframe << 16) | (sc << 20);
}

extern "C" int mobi_launch_motion_search(const MobiReconArgs *a, const uint8_t *src_dev, uint32_t *out_dev, int n_past, hipStream_t s) {
  if (a->n_clips <= 0) return 0;
  hipLaunchKernelGGL(mobi_motion_search_2x2, dim3((unsigned)a->n_clips * (unsigned)a->n_mbs), dim3(64), 0, s, *a, src_dev, out_dev, n_past);
  return (int)hipGetLastError();
}
