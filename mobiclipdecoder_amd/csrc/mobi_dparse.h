// mobi_dparse.h -- interface of the device-side bitstream parser (mobi_dparse.hip), SURVEY.md 8(f) row 3.
//
// The serial VLC / Elias-gamma parse of one frame cannot be split inside a clip (every code's position depends on the
// previous one), but clips share nothing (MD.cs:15-39).  mobi_parse_frames runs the same parser as mobi_parse.cpp with one
// wave per clip -- thousands of clips in flight on one GPU -- and writes the same command list (mobi_cmd.h) straight into
// HBM, so the reconstruction kernels start without a host parse or a command upload in between.
#ifndef MOBI_DPARSE_H
#define MOBI_DPARSE_H
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "mobi_cmd.h"

#include "mobi_dparse_tables.h"
#include "mobi_state.h" // MobiDevState, MobiDevTail: the decoder state that survives from frame to frame

struct MobiDevResult { // per clip, read back by the host after the parse launch
  int32_t rc;          // MOBI_OK / MOBI_E_*
  int32_t consumed;    // bytes the bit reader advanced from the start offset (Offset out = Offset in + consumed)
  uint32_t n_intra;    // intra macroblocks = items written
  uint32_t payload_words;
  uint32_t quant, yuvfmt;
  uint32_t frame_type; // 1 = I
  uint32_t pad;
};
struct MobiDevParseArgs {
  const uint8_t *bits;      // frame bytes of every clip; clip c starts at bits + bit_off[c] (8-byte aligned, >= 32 zero bytes follow)
  const uint64_t *bit_off;
  const uint32_t *bit_len;  // Data.Length - Offset of clip c (0: nothing readable; MOBI_DP_SKIP: the clip is the host parser's -- no kernel touches
                            // anything of it, its command list is uploaded into its rows beside the parse)
  const uint8_t *tables;    // MOBI_DT_BYTES blob
  // [clip] decoder state: read from *_in, written to *_out -- two entries of a ring of three (mobi_abi.cpp), so that the state a frame
  // STARTED from is still there when its parse turns out not to be the device's to finish and the host parser takes the frame over
  const MobiDevState *state_in;
  MobiDevState *state_out;
  const MobiDevTail *tail_in;
  MobiDevTail *tail_out;    // .mvc: written by the parsers at the end of a P-frame; the rest by mobi_parse_tail
  const int32_t *scale;     // [quantizer][MOBI_SCALE_STRIDE] dequant scales by natural index (MobiReconArgs.scale): mobi_parse_tail
  MbDesc *desc;             // [clip][n_mbs]  out
  uint32_t *payload;        // out: clip c owns words [c * pay_cap, (c + 1) * pay_cap)
  uint32_t *items;          // [clip][n_mbs] out: MOBI_ITEM(clip, mb) of the intra macroblocks in raster order, rest untouched
  MobiDevResult *res;       // [clip] out
  uint32_t pay_cap;
  int n_clips, version, width, height, stride, lg, mbw, mbh;
  // lockstep != 0: mobi_parse_frames_ls (mobi_lsparse.hip: 32 clips per wave) runs first; what it finishes carries MobiDevResult.pad ==
  // LS_MAGIC and its new decoder state in state_ls[clip]; mobi_parse_frames then only moves that state into place and parses the others
  MobiDevState *state_ls;
  int lockstep;
  int ls_clips; // clips per wave of mobi_parse_frames_ls (filled in by mobi_launch_parse_ls)
  int ls_mv_lds; // ... and whether its lanes keep the MV row cache in LDS (room to spare) or in the clips' tails in HBM
  // pay_local != 0: MbDesc.payload_off is written relative to the clip's own part of the arena (MobiReconArgs.pay_clip_words = pay_cap), so
  // that n_clips * pay_cap may exceed 2^32 words; 0: relative to the arena (the hybrid mode, whose host-parsed clips sit behind the others)
  int pay_local;
  // r06, frame-parallel parse (mobi_gop.h): the n_clips entries are VIRTUAL clips v = k * clip_mod + c (frame k of clip c); the intra items
  // name the real clip (v mod clip_mod), which is what the reconstruction of frame k indexes its tables with.  0: clips are clips.
  int clip_mod;
  int skip_tail; // mobi_launch_parse leaves mobi_parse_tail out: mobi_gop_chain rebuilds the tails frame by frame
};
#define MOBI_DP_SKIP 0xFFFFFFFFu
extern "C" int mobi_launch_parse(const MobiDevParseArgs *a, hipStream_t s); // the parse kernels, then mobi_parse_tail
extern "C" int mobi_launch_parse_ls(const MobiDevParseArgs *a, hipStream_t s); // the two kernels in front (called by mobi_launch_parse)
#endif
