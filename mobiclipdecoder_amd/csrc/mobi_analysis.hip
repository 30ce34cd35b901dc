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

// =====================================================================================================
// Forward transforms of residual blocks: MobiEncoder.DCT64 (Encoder/MobiEncoder.cs:962-1010) and DCT16 (:1146-1178)
// =====================================================================================================
// The other compute loop of the reference's encoder (Encoder/MacroBlock.cs:584-590, 612-616: residual = Block - CompVals, DCT, quantise).
// Pure integer: x64, a fixed integer matrix per pass, a C# integer division (truncating) -- rows, then columns, results of the second
// pass stored transposed, as the reference does.  One lane = one row (then one column) of a block: 8 lanes per 8x8 block, 4 per 4x4;
// the transpose between the passes goes through LDS.  512 B (128 B) of traffic per block: HBM-bound.
// (The quantiser behind it is float division + Math.Round over a float table (MacroBlock.cs:591-595): not reproducible bit for bit
// outside the CLR that ran it, and not built.)
namespace {
__device__ __forceinline__ void dct8_pass(const int (&x)[8], int (&o)[8]) { // x = samples 0..7 of the row / column
  const int p = x[0], q = x[7], r = x[2], s = x[5], t = x[3], u = x[4], v = x[1], w = x[6];
  o[0] = (w + v + u + t + s + r + q + p) / 8;
  o[1] = (-40 * w + 40 * v - 12 * u + 12 * t - 24 * s + 24 * r - 48 * q + 48 * p) / 289;
  o[2] = (w + v - 2 * u - 2 * t - s - r + 2 * q + 2 * p) / 10;
  o[3] = (12 * w - 12 * v + 24 * u - 24 * t + 48 * s - 48 * r - 40 * q + 40 * p) / 289;
  o[4] = (-w - v + u + t - s - r + q + p) / 8;
  o[5] = (48 * w - 48 * v - 40 * u + 40 * t - 12 * s + 12 * r - 24 * q + 24 * p) / 289;
  o[6] = (-2 * w - 2 * v - u - t + 2 * s + 2 * r + q + p) / 10;
  o[7] = (24 * w - 24 * v + 48 * u - 48 * t - 40 * s + 40 * r - 12 * q + 12 * p) / 289;
}
__device__ __forceinline__ void dct4_pass(const int (&x)[4], int (&o)[4]) {
  const int q = x[0], r = x[1], s = x[2], t = x[3];
  o[0] = (t + s + r + q) / 4;
  o[1] = (-2 * t - s + r + 2 * q) / 5;
  o[2] = (t - s - r + q) / 4;
  o[3] = (-t + 2 * s - 2 * r + q) / 5;
}
} // namespace

extern "C" __global__ __launch_bounds__(256) void mobi_fwd_dct8(const int32_t *in, int32_t *out, uint32_t n_blocks) {
  __shared__ int tile[32][8][9]; // 32 blocks per workgroup; pitch 9: the column reads of eight lanes hit eight banks
  const uint32_t blk = blockIdx.x * 32u + (threadIdx.x >> 3);
  const int lb = threadIdx.x >> 3, r = threadIdx.x & 7;
  const bool on = blk < n_blocks;
  int x[8], o[8];
  if (on) {
    const int4 a = *(const int4 *)(in + (size_t)blk * 64 + r * 8), b = *(const int4 *)(in + (size_t)blk * 64 + r * 8 + 4);
    x[0] = a.x * 64; x[1] = a.y * 64; x[2] = a.z * 64; x[3] = a.w * 64; x[4] = b.x * 64; x[5] = b.y * 64; x[6] = b.z * 64; x[7] = b.w * 64;
    dct8_pass(x, o);
#pragma unroll
    for (int k = 0; k < 8; k++) tile[lb][r][k] = o[k];
  }
  __syncthreads();
  if (on) {
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = tile[lb][j][r]; // column r
    dct8_pass(x, o);
    *(int4 *)(out + (size_t)blk * 64 + r * 8) = int4{o[0], o[1], o[2], o[3]};
    *(int4 *)(out + (size_t)blk * 64 + r * 8 + 4) = int4{o[4], o[5], o[6], o[7]};
  }
}
extern "C" __global__ __launch_bounds__(256) void mobi_fwd_dct4(const int32_t *in, int32_t *out, uint32_t n_blocks) {
  __shared__ int tile[64][4][5];
  const uint32_t blk = blockIdx.x * 64u + (threadIdx.x >> 2);
  const int lb = threadIdx.x >> 2, r = threadIdx.x & 3;
  const bool on = blk < n_blocks;
  int x[4], o[4];
  if (on) {
    const int4 a = *(const int4 *)(in + (size_t)blk * 16 + r * 4);
    x[0] = a.x * 64; x[1] = a.y * 64; x[2] = a.z * 64; x[3] = a.w * 64;
    dct4_pass(x, o);
#pragma unroll
    for (int k = 0; k < 4; k++) tile[lb][r][k] = o[k];
  }
  __syncthreads();
  if (on) {
#pragma unroll
    for (int j = 0; j < 4; j++) x[j] = tile[lb][j][r];
    dct4_pass(x, o);
    *(int4 *)(out + (size_t)blk * 16 + r * 4) = int4{o[0], o[1], o[2], o[3]};
  }
}
extern "C" int mobi_launch_fwd_dct(int n, const int32_t *in_dev, int32_t *out_dev, uint32_t n_blocks, hipStream_t s) {
  if (n_blocks == 0) return 0;
  if (n == 8) hipLaunchKernelGGL(mobi_fwd_dct8, dim3((n_blocks + 31) / 32), dim3(256), 0, s, in_dev, out_dev, n_blocks);
  else if (n == 4) hipLaunchKernelGGL(mobi_fwd_dct4, dim3((n_blocks + 63) / 64), dim3(256), 0, s, in_dev, out_dev, n_blocks);
  else return (int)hipErrorInvalidValue;
  return (int)hipGetLastError();
}

// ---- "a checksum of checksums" for batches made of copies (bench.py, soak runs): ring slot 0 of every clip against the same slot of clip
// (clip mod modulus), byte for byte in the private tiled layout (padding included: it is zero everywhere).  One workgroup = 16 KB of one clip;
// out[clip] += 16-byte words that differ. ----
extern "C" __global__ __launch_bounds__(256) void mobi_compare_clips(const uint8_t *planes, uint64_t clip_bytes, uint32_t slot_bytes, int slot, uint32_t modulus,
                                                                       uint32_t chunks, uint32_t *out) {
  const uint32_t clip = blockIdx.x / chunks, chunk = blockIdx.x - clip * chunks, ref = clip % modulus;
  if (ref == clip) return;
  const uint8_t *a = planes + (size_t)clip * clip_bytes + (size_t)slot * slot_bytes, *r = planes + (size_t)ref * clip_bytes + (size_t)slot * slot_bytes;
  uint32_t bad = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t o = chunk * 16384u + (uint32_t)(k * 256 + threadIdx.x) * 16u;
    if (o < slot_bytes) {
      const uint4 x = *(const uint4 *)(a + o), y = *(const uint4 *)(r + o);
      bad += (x.x != y.x) | (x.y != y.y) | (x.z != y.z) | (x.w != y.w);
    }
  }
  if (bad) atomicAdd(out + clip, bad);
}
extern "C" int mobi_launch_compare_clips(const MobiReconArgs *a, int modulus, uint32_t *out_dev, hipStream_t s) {
  if (a->n_clips <= 0 || modulus <= 0) return (int)hipErrorInvalidValue;
  const uint32_t chunks = (a->slot_bytes + 16383u) / 16384u;
  hipLaunchKernelGGL(mobi_compare_clips, dim3((unsigned)a->n_clips * chunks), dim3(256), 0, s, (const uint8_t *)a->planes, (uint64_t)a->clip_bytes, a->slot_bytes,
                     a->ring_base, (uint32_t)modulus, chunks, out_dev);
  return (int)hipGetLastError();
}
