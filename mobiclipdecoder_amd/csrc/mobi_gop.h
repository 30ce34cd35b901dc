// mobi_gop.h -- frame-parallel device parse (r06): K consecutive frames of every clip parsed SIDE BY SIDE.
//
// The parse of a frame is serial inside the frame, and until r05 a device-parsed step was one frame of every clip: lanes (lock-step parser)
// or waves (mobi_parse_frames) were clips, and the number of clips is capped by the rings in HBM.  But what a frame's PARSE needs from the
// frame before it is tiny and, measured (tools/exp_framedep.py, profiles/r06_framedep.txt: 0 of 9 100 frames, damaged ones included),
// entirely determined by the frame HEADERS:
//   * Quantizer -- P-frame: += a signed Elias-gamma delta read right behind the frame-type bit (MD.cs:113-143); I-frame: a 6-bit field
//     (MD.cs:224-236) -- and with it the dequant tables and the intra-mode cache's BORDER bytes, which SetupQuantizationTables sets to 9
//     (MD.cs:3913-3924) and nothing else ever writes;
//   * YuvFormat (I-frame header), the number of frames in the ring (+1 per frame, whatever happens: MD.cs:102-108).
// The 16 INTERIOR bytes of the mode cache (Internal bytes 9..12, 17..20, 25..28, 33..36) are only ever read after a block of the SAME
// macroblock wrote them (MD.cs:1840-1859, 2785-2843: a block at byte r5 reads r5 - 8 and r5 - 1 and writes r5, r5 + 1, r5 + 8, r5 + 9; the
// blocks of a macroblock go 9, 11, 25, 27 and their 4x4 parts r5, r5 + 1, r5 + 8, r5 + 9: every byte read is a border byte or was written
// earlier in the macroblock), the MV predictor Internal[219], [220] is set per macroblock before it is read (MD.cs:207-208), Internal[218]
// and the MV row cache are reset per frame (MD.cs:144-154).  They have to be CARRIED (the state a later hand-over to the host parser
// starts from must be the sequential one) but they never steer a parse.
//
// So: a virtual clip v = k * n + c is frame k of clip c.  mobi_gop_prepare walks the K headers of every clip (a few bytes each) and writes
// every virtual clip's start state -- interior bytes 0xFF = "not written yet"; the parse kernels run over n * K virtual clips as they would
// over clips; mobi_gop_chain then goes through every clip's frames in order: it VERIFIES each start state against what the frame before
// really left (any mismatch, like any frame a device parser did not finish, hands the clip's remaining frames to the host parser: the
// prediction can only cost time, never the result), merges the carried bytes, and rebuilds the tail (mobi_state.h) frame by frame.
// The reconstruction stays one step per frame, K steps from K consecutive command lists.
#ifndef MOBI_GOP_H
#define MOBI_GOP_H
#include <stdint.h>

#include "mobi_state.h"

#define MOBI_GOP_MAX 6 /* the ring holds six pictures (MD.cs:19-20): every frame of a group is still there when the call returns */
#define MOBI_GOP_PARSE_MAX 128 /* mobi_batch_gop_begin: frames parsed side by side (one turn of full waves of the lock-step parser = 131 072: 4096 clips x 32, 1024 x 128); mobi_batch_gop_finish hands them out six at a time */
#define MOBI_GOP_RC_CHAIN (-99) /* MobiDevResult.rc of a frame whose start state was predicted wrong (device-private: the host parser takes over) */
#define MOBI_MC_UNWRITTEN 0xFFu

// the 16 interior bytes of the mode cache as a bit mask over its 40 bytes
#define MOBI_MC_INTERIOR_MASK ((0xFull << 9) | (0xFull << 17) | (0xFull << 25) | (0xFull << 33))
MOBI_ST_FN bool mobi_mc_interior(int i) { return (MOBI_MC_INTERIOR_MASK >> i) & 1; }

// What a device parser that FINISHES the frame makes of its header (mobi_lsparse.h ls_begin_frame / ls_setup_quant, MD.cs:113-143, 224-236,
// 3884-3925): quant, yuvfmt, tables_set and the border bytes.  p = the frame's first bytes (at least 4 readable, zero beyond len).
// Returns false when no device parser finishes a frame with this header (st is then left alone): nothing behind it will be used.
MOBI_ST_FN bool mobi_gop_header(int version_moflex, const uint8_t *p, uint32_t len, MobiDevState &st) {
  if (len < 2) return false;
  const uint32_t w0 = (uint32_t)p[0] | ((uint32_t)p[1] << 8), w1 = len >= 4 ? (uint32_t)p[2] | ((uint32_t)p[3] << 8) : 0u;
  uint32_t win = (w0 << 16) | w1;
  const bool iframe = win >> 31;
  win <<= 1;
  uint32_t q = st.quant;
  bool setup = false;
  if (iframe) {
    st.yuvfmt = win >> 31;
    const uint32_t nq = (win << 2) >> 26;
    if (q != nq) { q = nq; setup = true; }
  } else {
    int z = 0;
    while (z < 8 && !((win << z) >> 31)) z++;
    if (z >= 8) return false; // (a code of more than 15 bits: the device parsers hand the frame over)
    const uint32_t u = (z ? ((win << (z + 1)) >> (32 - z)) : 0u) + (1u << z);
    int dq = (int)u;
    if (dq & 1) dq = (int)(1u - u);
    dq >>= 1;
    if (version_moflex && q == 0) setup = true;
    else if (dq != 0) { q += (uint32_t)dq; setup = true; }
  }
  if (setup) {
    if (version_moflex) q = q < 12 ? 12 : q > 52 ? 52 : q;
    if (q >= 54) return false; // SetupQuantizationTables throws after assigning Quantizer (MD.cs:3886-3890): the host parser's frame
    st.quant = q;
    st.tables_set = 1;
    st.mcache[1] = st.mcache[2] = st.mcache[3] = st.mcache[4] = 9;
    st.mcache[8] = st.mcache[16] = st.mcache[24] = st.mcache[32] = 9;
  }
  return true;
}
// the start state of frame k + 1 as predicted from the (predicted) start state of frame k and frame k's header
MOBI_ST_FN void mobi_gop_next_guess(int version_moflex, const uint8_t *p, uint32_t len, MobiDevState &st) {
  (void)mobi_gop_header(version_moflex, p, len, st);
  st.frames_started++;
  for (int i = 0; i < 40; i++)
    if (mobi_mc_interior(i)) st.mcache[i] = MOBI_MC_UNWRITTEN;
}
// does a predicted start state agree with the true one in everything a parse can depend on?
MOBI_ST_FN bool mobi_gop_guess_ok(const MobiDevState &guess, const MobiDevState &truth) {
  bool ok = guess.quant == truth.quant && guess.yuvfmt == truth.yuvfmt && guess.frames_started == truth.frames_started && guess.tables_set == truth.tables_set;
  for (int i = 0; i < 40; i++)
    if (!mobi_mc_interior(i)) ok = ok && guess.mcache[i] == truth.mcache[i];
  return ok;
}
// what a frame parsed from a predicted start state left, made into what the sequential decoder would hold: bytes the frame never wrote come
// from the true state before it; an I-frame does not touch Internal[219], [220] (MD.cs:224-249)
MOBI_ST_FN void mobi_gop_merge(const MobiDevState &before, bool iframe, MobiDevState &after) {
  for (int i = 0; i < 40; i++)
    if (mobi_mc_interior(i) && after.mcache[i] == MOBI_MC_UNWRITTEN) after.mcache[i] = before.mcache[i];
  if (iframe) { after.predx = before.predx; after.predy = before.predy; }
}

#if defined(__HIPCC__) || defined(MOBI_GOP_DEVICE_DECLS)
#include "mobi_dparse.h"
// P: the parse launch's arguments over n * K virtual clips (state_in / state_out / tail_out / res / desc / payload / bit_off / bit_len by
// virtual clip; clip_mod = n).  ring_*: the batch's state ring entries the group reads and writes, by clip.
struct MobiGopArgs {
  MobiDevParseArgs P;
  const MobiDevState *ring_in;
  MobiDevState *ring_out;
  const MobiDevTail *rtail_in;
  MobiDevTail *rtail_out;
  int n, K;
};
// The intra macroblocks of the group's frames as launch items in WAVEFRONT order (mobi_recon_intra's own format: mobi_kernels.h), built on
// the device from what the parsers left -- per virtual clip raster-order lists and the descriptors -- behind the host parser's overrides.
// The raster-order launch (mobi_recon_intra_cl: a wave = the same list slot of four clips, every row polling the rows it depends on) takes
// 1.5 - 1.7 ms per P-frame step of 24576 clips and 25 ms per I-frame step; the host-built dependency-level order 1.0 and 20.  Levels proper
// are a chain through every clip's list (a lane per clip walking it: 28 ms per group of 5 x 24576, tried); the wavefront mbx + 2 * mby is
// a valid order as well -- whatever a halo reads (left, above left, above, above right, and at the picture's right edge the first macroblock
// of the same row: ls_intra_deps, mobi_lsparse.h) lies on an earlier one -- and needs the macroblock's index only: with thousands of clips
// every wavefront holds many rows, which is all the order is for.  NOT valid where Width == Stride (256, 512, 1024 wide): there the halo of a
// row's first macroblock wraps to the last one of the row above (a later wavefront); the caller keeps the raster-order launch for those.
//   hist / cursor / start  [K][MOBI_SORT_LEVELS] uint32 (hist and cursor zeroed by the caller)
//   sorted  frame k's items at sorted + sorted_off[k] * 4 words, sorted_cap[k] items of room, the frames back to back (every wavefront
//           starts on a wave of four items; the padding, and what is left behind the last one, are MOBI_ITEM_NONE rows that do nothing)
#define MOBI_SORT_LEVELS 1024 /* mbw + 2 * mbh of the largest picture (1024 x 2048: 64 + 256) and room */
struct MobiGopSortArgs {
  const MbDesc *desc;          // [v][n_mbs]
  const uint32_t *items;       // [v][n_mbs] raster-order MOBI_ITEM(clip, mb)
  const MobiDevResult *res;    // [v] n_intra
  uint32_t *hist, *cursor, *start;
  uint32_t *sorted;
  uint64_t sorted_off[MOBI_GOP_PARSE_MAX];
  uint32_t sorted_cap[MOBI_GOP_PARSE_MAX];
  int n, K, n_mbs, mbw;
};
extern "C" int mobi_launch_gop_sort(const MobiGopSortArgs *a, hipStream_t s);
extern "C" int mobi_launch_gop_prepare(const MobiGopArgs *a, hipStream_t s); // start states of the n * K virtual clips (before the parse kernels)
extern "C" int mobi_launch_gop_chain(const MobiGopArgs *a, hipStream_t s);   // verify, merge, tails (behind them)
#endif
#endif
