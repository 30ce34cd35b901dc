// mobi_cmd.h -- the host->GPU command list: what the serial bitstream parser (mobi_parse.cpp)
// emits per frame and what the reconstruction kernels (mobi_kernels.hip) consume.
//
// One frame of one clip =
//   MbDesc   desc[n_mbs]           32 B per macroblock, raster order (leaves of 1- and 2-leaf inter MBs are inline)
//   uint32_t payload[...]          variable: MV cell maps, intra block records, residual levels
//   uint32_t intra_items[...]      MB indices of intra MBs grouped by dependency level (host side only;
//                                  merged across clips into per-level launch lists)
//
// Positions are NOT stored as MB coordinates: like the reference, everything is a linear byte
// offset into the strided plane (MD.cs:212-217, SURVEY hard part 3); desc index -> offset is
// (mb / mbw) * 16 * stride + (mb % mbw) * 16 because widths are multiples of 16 here.
#ifndef MOBI_CMD_H
#define MOBI_CMD_H
#include <stdint.h>

#if defined(__HIPCC__)
#define MOBI_CMD_FN static __host__ __device__ __forceinline__
#else
#define MOBI_CMD_FN static inline
#endif

enum { MOBI_MB_INTER = 0, MOBI_MB_INTRA = 1 };

// ---- MbDesc: 32 B per macroblock; on the device one flat table per frame step, index = clip*n_mbs + mb,
//      so a wave's first load already tells it everything it needs to start fetching pixels ------------
// w0  payload word offset (host: inside the clip's payload; device: inside the frame step's payload arena)
// w1  [0]      type (MOBI_MB_*)
//     [7:1]    n_leaves   (inter: 1 = the single 16x16 leaf; 2 with a DUAL kind = both halves; both kinds ride in the
//                          descriptor as LEAF RECORDS, see below; otherwise the payload starts with the 64-word MV cell map)
//     [13:8]   cbp6       coded 8x8 areas: bits 0-3 luma TL,TR,BL,BR; 4 U; 5 V   (MD.cs:1820-1832)
//     [19:14]  t8mask     coded area uses ONE 8x8 transform (else four 4x4s)      (MD.cs:2911)
//     [25:20]  quantizer  of the frame (selects the dequant scale table, MD.cs:3884-3912)
//     [28]     reserved (r02: "right neighbour is intra", for the edge side buffer the tiled planes made unnecessary)
//     [27:26]  MOBI_DUAL_*: the macroblock is exactly two halves (partition codes 8 / 9 at the 16x16 level with two
//              plain leaves, MD.cs:585-600 -- by far the most common split), leaf A = top / left, B = bottom / right
// w2  [9:0]    n_coefs (<= 384)
//     inter leaf records, decoded by the host so that the kernel does no motion-vector arithmetic (MD.cs:400-416):
//     [12:10] ref slot 1..5 of leaf A   [15:13] of leaf B
//     [17:16] luma CopyBlock phase (dx&1)|((dy&1)<<1) of A   [19:18] chroma phase of A   [21:20], [23:22] the same for B
// w3  inter: luma source position of leaf A = MB offset + (dy>>1)*Stride + (dx>>1), linear inside the reference's Y plane
//            (signed: a bottom/right half may point above/left of its macroblock's origin; its own rows/columns add back)
//     intra: [0] plane16 present, [1] some dependency (w4..w7) is an intra macroblock, [2] some intra macroblock depends on this one,
//            [4] the plane16 parameter does not fit [31:16]: it is wide parameter 24 (below), [31:16] plane16 parameter
// w4  inter: chroma source position of leaf A = MB offset/2 + ((dy>>1)>>1)*Stride + ((dx>>1)>>1), inside the UV plane (U half)
// w5, w6  inter DUAL: the same two positions for leaf B
// w7  reserved (0) for inter macroblocks
// intra: w4..w7 hold up to 8 uint16 macroblock indices (MOBI_DEP_NONE = unused): the raster-earlier macroblocks of the
//        same frame, inter or intra, whose pixels this one's prediction halo reads.  When a frame step runs as one
//        launch the macroblock waits for exactly these (mobi_recon_step in mobi_kernels.hip).
struct MbDesc {
  uint32_t payload_off;
  uint32_t w1;
  uint32_t w2;
  uint32_t w3;
  uint32_t w4;
  uint32_t w5;
  uint32_t w6;
  uint32_t w7;
};
#define MOBI_DEP_NONE 0xFFFFu
#define MOBI_DEP_INTER 0x8000u /* flag on a dependency index: that macroblock is an inter one (index = low 13 bits) */
#define MOBI_INTRA_DEPS 8
enum { MOBI_DUAL_NONE = 0, MOBI_DUAL_TB = 1, MOBI_DUAL_LR = 2 }; // two 16x8 (top, bottom) / two 8x16 (left, right)

// ---- MC leaf as the parser records it while walking the partition tree (host only) ----------------
//  w0: [3:0] x/2  [7:4] y/2  [9:8] log2(16/w)  [11:10] log2(16/h)  [14:12] ref slot 1..5
//  w1: [15:0] dx (int16, half-pel, absolute)  [31:16] dy                         (MD.cs:400-416)
MOBI_CMD_FN uint32_t mobi_leaf_w0(int x, int y, int wi, int hi, int ref) {
  return (uint32_t)((x >> 1) | ((y >> 1) << 4) | (wi << 8) | (hi << 10) | (ref << 12));
}
MOBI_CMD_FN uint32_t mobi_leaf_w1(int dx, int dy) { return ((uint32_t)dx & 0xFFFFu) | ((uint32_t)dy << 16); }

// ---- MV cell map (macroblocks with more than one leaf that are not DUAL): 64 words, first thing in the payload ---
// The partition tree bottoms out at 2x2 luma (MD.cs:1683-1746), so an 8x8 grid of 2x2-pixel cells
// (= one chroma sample each) says for every pixel which leaf moved it.  cell[(y/2)*8 + x/2] =
//  [13:0] dx (signed 14)  [27:14] dy (signed 14)  [30:28] ref slot 1..5
// Every lane fetches the cells under its own pixels, so all reference reads of a macroblock are in
// flight together, however deep the tree was.
#define MOBI_MV_CELLS 64
#define MOBI_MV_LIMIT 8191
MOBI_CMD_FN uint32_t mobi_cell(int dx, int dy, int ref) {
  return ((uint32_t)dx & 0x3FFFu) | (((uint32_t)dy & 0x3FFFu) << 14) | ((uint32_t)ref << 28);
}
MOBI_CMD_FN int mobi_cell_dx(uint32_t c) { return (int)(c << 18) >> 18; }
MOBI_CMD_FN int mobi_cell_dy(uint32_t c) { return (int)(c << 4) >> 18; }
MOBI_CMD_FN int mobi_cell_ref(uint32_t c) { return (int)(c >> 28) & 7; }

// ---- residual level: one word ---------------------------------------------------------------
//  [8:0]   tile position = area*64 + p, area = 0..5 (Y0..Y3,U,V)
//            8x8 transform:  p = natural-order coefficient index (MD.cs:3426 zigzag target)
//            4x4 transforms: p = sub*16 + natural index inside that 4x4 (sub = 0..3: TL,TR,BL,BR)
//  [31:16] level (int16); the GPU multiplies by the dequant scale (MD.cs:3427-3429)
MOBI_CMD_FN uint32_t mobi_coef(int area, int p, int level) {
  return (uint32_t)(area * 64 + p) | ((uint32_t)level << 16);
}

// ---- intra MB payload: 24 block records (6 areas x 4) then the levels ------------------------
// record for area a, slot s (s = 0 only when the area is predicted as one 8x8):
//  [3:0]  mode 0..9 (8x8 numbering; 4x4 blocks use the same numbering, MD.cs mode-10)
//  [4]    residual coded for this block
//  [5]    split: the area is four 4x4 blocks (slots 0..3 all valid)
//  [7]    the plane parameter does not fit int16 (a code of 33 bits and more): it is WIDE PARAMETER r, r = this record's index
//  [31:16] plane parameter (int16) when mode == 2                           (MD.cs:3019,3170,3255)
// Wide parameters (r05): MOBI_WIDE_PARAMS words behind the macroblock's level words -- payload word MOBI_INTRA_RECORDS + n_coefs + r, r = the
// record's index, 24 = the 16x16 plane's -- present only when some record or MbDesc.w3 says so.  The plane predictors compute in int32 and
// OR their samples into words (MD.cs:3055-3062): every bit of the parameter reaches the picture.
// MbDesc.w3: [0] luma plane16 present, [1] chroma plane8 pair present,
//            [31:16] plane16 param; chroma plane params live in the U/V slot-0 records with mode 9:
//            record bit [6] = "run plane8 with param before this area" (keeps decode order).
#define MOBI_INTRA_RECORDS 24
#define MOBI_WIDE_PARAMS 25
#define MOBI_REC_WIDE 0x80u
#define MOBI_W3_WIDE 0x10u
MOBI_CMD_FN uint32_t mobi_intra_rec(int mode, int coded, int split, int pre_plane, int param) {
  return (uint32_t)(mode | (coded << 4) | (split << 5) | (pre_plane << 6)) | ((uint32_t)param << 16);
}

// ---- per-frame info kept on the host (launch planning, accounting); the kernels never read it ----------
struct FrameHdr {
  uint32_t frame_type;   // 0 = P, 1 = I
  uint32_t n_mbs;
  uint32_t n_intra;      // number of intra MBs
  uint32_t n_levels;     // highest intra dependency level (0 when no intra MBs)
  uint32_t payload_words;
  uint32_t quantizer;
  uint32_t cmd_bytes;    // bytes of this frame's command list the kernels read (desc + payload)
  uint32_t reserved;
};

// Dequant scale tables by NATURAL coefficient index for one quantizer: scale = dequant word >> 8
// (MD.cs:3897-3911; 4x4: tbl<<(q/6), 8x8: tbl<<(q/6-2)).  Valid for q in [12,53] (no zigzag-byte leak).
// Layout of one entry of the device table: int[80] = scale8[64] then scale4[16].
#define MOBI_SCALE_STRIDE 80
#define MOBI_SCALE_QMAX 54
// Intra tap table (mobi_build_intra_taps): behind the 64 rows of scale tables in the same device buffer.  Entry = 4 x int16 tile
// offsets; 8x8 blocks: entry mi * 64 + y * 8 + x, 4x4 blocks: MOBI_TAP_4X4 + mi * 16 + y * 4 + x, mi = mode < 2 ? mode : mode - 2
#define MOBI_TAP_MODES 7
#define MOBI_TAP_4X4 (MOBI_TAP_MODES * 64)
#define MOBI_TAP_ENTRIES (MOBI_TAP_MODES * 80)
#define MOBI_TAP_PITCH 32 /* the intra kernel's tile pitch */
#define MOBI_SCALE_ROWS 64
#define MOBI_SCALE_LITERAL 63 /* the row of ones: MbDesc.w1's quantiser field of a frame whose residual words carry coefficient VALUES
                                 (the host parser's Internal[] walk, mobi_parse.cpp); no real quantiser reaches 54 (MD.cs:3864-3880) */
#define MOBI_TQ_NONE 62      /* a row of zeros: MbDesc.w1's field while SetupQuantizationTables has never run -- every dequant word is 0
                                 (MD.cs:28: a fresh Internal[]), so every coefficient is.  The field is the quantiser the TABLES were built for,
                                 which is not Quantizer after a SetupQuantizationTables that threw (MD.cs:3886-3890; ModsDS, q >= 54) */

#endif
