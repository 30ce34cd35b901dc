/*
 * mobi_oracle.h -- CPU oracle for the Mobiclip frame decoder (TEST INFRASTRUCTURE ONLY).
 *
 * This is a plain-C restatement of the reference decoder
 *   /root/reference/LibMobiclip/Codec/Mobiclip/MobiclipDecoder.cs  (DecodeVXS2 and callees,
 *   :97-259 and :400-3937; the Bitmap/RGB block :260-323 is restated separately in mobi_oracle_argb)
 * used as the checker for the HIP path.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product library (libmobiclip_hip.so)
 * never links, loads or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or sample media, and the
 * C# reference cannot run in this image (no .NET).  See DESIGN.md "Oracle pinning" for the
 * cross-checks that substitute (self-consistency identities, generator round trips, and a
 * dev-time differential run against a mechanical transliteration of the C# source).
 */
#ifndef MOBI_ORACLE_H
#define MOBI_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* MobiclipDecoder.MobiclipVersion, MD.cs:32-37 */
enum { MOBI_VER_VXDS = 0, MOBI_VER_MODSDS = 1, MOBI_VER_MOFLEX3DS = 2 };

/* "what the C# code would have thrown" -- the reference swallows all of these (MD.cs:325) */
enum {
  ORA_OK = 0,
  ORA_E_INDEX = -1,     /* IndexOutOfRangeException / ArgumentException (array bounds) */
  ORA_E_NULLREF = -2,   /* NullReferenceException: reference frame slot never decoded */
  ORA_E_PARTCODE = -3,  /* explicit `throw new Exception()` on illegal partition code */
  ORA_E_VERSION = -4,   /* VxDS stub (NotImplementedException) */
};

typedef struct mobi_oracle mobi_oracle;

mobi_oracle *mobi_oracle_create(uint32_t width, uint32_t height, int version);
void mobi_oracle_destroy(mobi_oracle *d);

/* d.Data = data; d.Offset = *offset; d.DecodeFrame(); *offset = d.Offset.
 * Returns ORA_OK when the reference would have reached the Bitmap stage, else the
 * exception class.  The ring is rotated and the (possibly partial) frame kept in either case,
 * exactly as MD.cs:102-108 + :325 leave it. */
int mobi_oracle_decode(mobi_oracle *d, const uint8_t *data, size_t len, int32_t *offset);

/* DecodeFrame() over a whole clip (frame f = data[frame_off[f] .. frame_off[f+1])), optionally with the Bitmap of every frame:
 * one call per clip, for timing.  Returns n_frames or the first error. */
int mobi_oracle_decode_clip(mobi_oracle *d, const uint8_t *data, const uint32_t *frame_off, int n_frames, uint32_t *argb_or_null);
int mobi_oracle_stride(const mobi_oracle *d);
uint32_t mobi_oracle_quantizer(const mobi_oracle *d);
uint32_t mobi_oracle_yuvformat(const mobi_oracle *d);
/* ring slot idx 0..5; NULL when that slot has never been produced.  Y: Stride*H bytes,
 * UV: Stride*H/2 bytes (U in columns [0,Stride/2), V in [Stride/2,Stride)). */
const uint8_t *mobi_oracle_y(const mobi_oracle *d, int idx);
const uint8_t *mobi_oracle_uv(const mobi_oracle *d, int idx);
/* The Bitmap DecodeFrame() returns (MD.cs:260-323), as width*height 0xAARRGGBB words, row pitch = width:
 * chroma averaged from up to four neighbours by pixel parity (not on the last column / last row), then float
 * BT.601-like conversion with 16..255 range stretch (Moflex3DS) or the integer Y+U-V / Y+V / Y-U-V form (ModsDS).
 * Float arithmetic: IEEE single, one rounding per C# operator in source order, no fused multiply-add (what the
 * x64 CLR's scalar SSE code does); casts truncate toward zero.  Returns ORA_E_NULLREF before the first frame. */
int mobi_oracle_argb(const mobi_oracle *d, uint32_t *out);
/* Encoder-side analysis, Analyzer.InterPredict2x2 (Analyzer.cs:608-681) for all 64 2x2 luma blocks of every macroblock
 * (SolveInterPredictionPuzzle's loop, :683-693): three-step search (steps 6, 3, 1 full pels) over up to five past frames,
 * ties broken towards the shorter vector; the encoder's PastFramesY[i] (MobiEncoder.cs:138-144) are this decoder's ring
 * slots Y[i].  src: the picture being analysed, width*height luma bytes, pitch = width (MacroBlock.YData2x2,
 * Encoder/MacroBlock.cs:76-85).  out[(mb * 64) + Y*8 + X] = (Delta.X & 0xFF) | (Delta.Y & 0xFF) << 8 | Frame << 16 |
 * score << 20, Delta in half pels as the reference stores it; score = 0xFFF when there is no past frame at all. */
void mobi_oracle_motion_search(const mobi_oracle *d, const uint8_t *src, uint32_t *out);
/* Syntax coverage of everything this process has decoded through the oracle so far (tests/test_coverage.py): counters, copied to
 * out[MOBI_COV_WORDS]; reset != 0 clears them afterwards.
 *   MOBI_COV_PART + (ver * 16 + shape) * 10 + code   partition codes 0..9 per shape (wi * 4 + hi, 16 of them) and table version
 *                                                     (0 = Moflex3DS, 1 = ModsDS), MD.cs:469-1746
 *   MOBI_COV_INTRA + mode                             PredictIntra modes 0..19 (10..19: the 4x4 twins), MD.cs:1883-2774
 *   MOBI_COV_PLANE + {0,1,2}                          plane predictors 16x16 / 8x8 / 4x4, MD.cs:3017-3327
 *   MOBI_COV_ESCAPE + {0,1,2}                         ReadDCTMatrix escapes: level offset, run offset, raw (MD.cs:3342-3406)
 *   MOBI_COV_VLCTAB + {0,1}                           residual VLC table 0 / 1 (Internal[218])
 *   MOBI_COV_REF + {0..4}                             motion-compensated leaves from ring slot 1..5
 *   MOBI_COV_PHASE + {0..3}                           luma CopyBlock phase (dx & 1) | (dy & 1) << 1
 *   MOBI_COV_IDCT + {0..5}                            IDCT1Px8, 3Px8, 16Px8, 64Px8, 1Px4, 16Px4 (chosen by the last scan index) */
enum { MOBI_COV_PART = 0, MOBI_COV_INTRA = 320, MOBI_COV_PLANE = 340, MOBI_COV_ESCAPE = 343, MOBI_COV_VLCTAB = 346, MOBI_COV_REF = 348,
       MOBI_COV_PHASE = 353, MOBI_COV_IDCT = 357, MOBI_COV_WORDS = 363 };
void mobi_oracle_coverage(uint64_t *out, int reset);
/* testing hooks: direct access to the Internal[392] word array (MD.cs:28) */
uint32_t *mobi_oracle_internal(mobi_oracle *d);

/* ---- unit-level entry points (operate on caller buffers; used by the identity tests) ---- */
/* full / reduced inverse transforms, MD.cs:3435-3798. coef: natural-order block (64 or 16 i32).
 * variant: 64,16,3,1 for 8x8 ; 16,1 for 4x4.  Returns 0 or ORA_E_INDEX (clamp-table domain). */
int mobi_oracle_idct8(const int32_t *coef, int variant, uint8_t *dst, int dst_len, int offset, int stride);
int mobi_oracle_idct4(const int32_t *coef, int variant, uint8_t *dst, int dst_len, int offset, int stride);
/* CopyBlock, MD.cs:418-456 */
int mobi_oracle_copyblock(const uint8_t *src, int src_len, int dx, int dy, uint32_t w, uint32_t h,
                          uint8_t *dst, int dst_len, int offset, int stride);
/* PredictIntra for modes that do not read the bitstream (0,1,3..9,10,11,13..19), MD.cs:1883.
 * is_uv selects the `Dst == UV[0]` V-plane fix-up (:1886).  plane modes: param given explicitly. */
int mobi_oracle_predict(int mode, uint8_t *dst, int dst_len, int offset, int stride, int is_uv);
int mobi_oracle_plane(int size /*16,8,4*/, int param, uint8_t *dst, int dst_len, int offset, int stride);
/* encoder-side forward transforms of a residual block: MobiEncoder.DCT64 (Encoder/MobiEncoder.cs:962) and
 * DCT16 (:1146): in = 64 / 16 residuals (Block - CompVals), out = 64 / 16 coefficients as the reference returns them */
void mobi_oracle_dct8(const int32_t *in, int32_t *out);
void mobi_oracle_dct4(const int32_t *in, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif
