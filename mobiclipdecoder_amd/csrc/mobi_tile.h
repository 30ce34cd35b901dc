// mobi_tile.h -- the PRIVATE plane layout of libmobiclip_hip (r03): macroblock tiles.
//
// The reference keeps a frame as two row-major byte arrays, Y[Stride * Height] and UV[Stride * Height / 2] with U in columns
// [0, Stride/2) and V in [Stride/2, Stride) of each row (MD.cs:107-108, 414-415), and everything it does to them is a LINEAR
// byte offset (MD.cs:212-217, 418-456): no edge clamping, rows wrap when Stride == Width, reads of the zero padding.  The
// command list keeps those linear offsets.  What changed is where the byte at linear offset `a` lives in HBM: the planes are
// private to the library (mobi_get_planes copies out in the reference layout), so the map below -- a bijection of
// [0, Stride * Height) resp. [0, Stride * Height / 2), a pure bit permutation of (row, column) -- keeps every offset's meaning
// and makes a macroblock's samples contiguous:
//
//   luma    tile = one macroblock = 256 B at index (row >> 4) * (Stride >> 4) + (col >> 4); inside it four 8x8 quadrants
//           TL, TR, BL, BR of 64 B, row-major: a 16-byte chunk = two rows of one quadrant, a 64-byte piece = one quadrant.
//   chroma  tile = one macroblock = 128 B at the same index; 8 rows of [U 8 B | V 8 B]: a 16-byte chunk = one row of both planes.
//
// Why (profiles/r03_ubench_tilepat.txt, tools/ubench/tilepat.hip): the window fetch of the inter kernel without any arithmetic
// runs at 4.75 instead of 3.44 TB/s of algorithmic bytes (vectors of +-16 pels; 2.6 x fewer L1->L2 requests: a 17 x 17 window
// is 3 x 3 quadrants instead of 17 pieces of 17 different lines), and an intra macroblock is 3 whole 128-byte lines instead of
// 32 partial ones (10 against 0.72 macroblocks per ns: HBM turns every store below 64 B into a read-modify-write).
// Height is a multiple of 16 (mobi_batch_create), Stride a power of two >= 256 (MD.cs:50-52): the tile grid covers the plane.
#ifndef MOBI_TILE_H
#define MOBI_TILE_H
#include <stdint.h>

#if defined(__HIPCC__)
#define MOBI_TILE_FN static __host__ __device__ __forceinline__
#else
#define MOBI_TILE_FN static inline
#endif

// row and column parts of a tiled luma address (they add)
MOBI_TILE_FN uint32_t mobi_ty_row(uint32_t row, int lgS) { return ((row >> 4) << (lgS + 4)) | ((row & 8u) << 4) | ((row & 7u) << 3); }
MOBI_TILE_FN uint32_t mobi_ty_col(uint32_t col) { return ((col >> 4) << 8) | ((col & 8u) << 3) | (col & 7u); }
// linear offset inside the reference's Y plane -> byte offset inside the tiled Y plane
MOBI_TILE_FN uint32_t mobi_ty(uint32_t a, int lgS) { return mobi_ty_row(a >> lgS, lgS) + mobi_ty_col(a & ((1u << lgS) - 1u)); }
// chroma: row part; column part of a column inside the U half (x < Stride / 2); V = + 8
MOBI_TILE_FN uint32_t mobi_tc_row(uint32_t row, int lgS) { return ((row >> 3) << (lgS + 3)) | ((row & 7u) << 4); }
MOBI_TILE_FN uint32_t mobi_tc_x(uint32_t x) { return ((x >> 3) << 7) | (x & 7u); }
// linear offset inside the reference's UV plane (either half) -> byte offset inside the tiled UV plane
MOBI_TILE_FN uint32_t mobi_tc(uint32_t a, int lgS) {
  const uint32_t col = a & ((1u << lgS) - 1u), v = col >> (lgS - 1), x = col & ((1u << (lgS - 1)) - 1u);
  return mobi_tc_row(a >> lgS, lgS) + mobi_tc_x(x) + (v << 3);
}
// first byte of macroblock (mbx, mby)'s luma tile / chroma tile
MOBI_TILE_FN uint32_t mobi_tile_y(uint32_t mbx, uint32_t mby, int lgS) { return ((mby << (lgS - 4)) + mbx) << 8; }
MOBI_TILE_FN uint32_t mobi_tile_c(uint32_t mbx, uint32_t mby, int lgS) { return ((mby << (lgS - 4)) + mbx) << 7; }

#endif
