/*
 * mobi_streamgen.h -- seeded synthetic Mobiclip bitstream generator (C ABI).
 *
 * The reference repository ships no sample media, so every test and
 * benchmark stream is produced here.  The writer follows the reference's bit packing
 * (LibMobiclip/Codec/Mobiclip/BitWriter.cs:16-65: MSB-first into 16-bit little-endian
 * words, Elias-gamma ue/se) and emits the syntax that MobiclipDecoder.cs parses
 * (SURVEY.md appendix A).  All VLC codes are obtained by inverting the decoder's own LUTs
 * (mobi_tables.h), so "generator -> parser" round trips are a self-consistency check on the
 * tables.  This is an input source, not part of the decode path.
 */
#ifndef MOBI_STREAMGEN_H
#define MOBI_STREAMGEN_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mobi_gen_params {
  uint32_t width, height; /* multiples of 16 */
  int32_t version;        /* 1 = ModsDS, 2 = Moflex3DS (MobiclipVersion, MD.cs:32-37) */
  uint64_t seed;
  int32_t n_frames;   /* frame 0 is always an I-frame */
  int32_t quantizer;  /* I-frame quantizer (12..52) */
  int32_t iframe_interval; /* 0: only frame 0; k>0: every k-th frame is an I-frame */
  /* P-frame macroblock mix, per mille; remainder = 16x16 with MV delta */
  int32_t pm_skip;    /* code 0: predicted MV, ref 1 */
  int32_t pm_split1;  /* exactly one split level (8 or 9), independent leaf MVs */
  int32_t pm_deep;    /* recursive splits down to 2x2 */
  int32_t pm_intra;   /* codes 6/7 */
  int32_t pm_multiref; /* per leaf: use ref 2..5 when that many frames exist */
  int32_t mv_range;   /* absolute MV components are drawn within +-mv_range half-pels of the predictor */
  int32_t cbp_prob;   /* per 8x8 block "coded" probability, per mille */
  int32_t t8_prob;    /* coded block uses one 8x8 transform (else 4x4 CBP), per mille */
  int32_t dense_prob; /* coded 8x8-transform block carries all 64 positions, per mille */
  int32_t max_coefs;  /* otherwise 1..max_coefs levels within the first `scan_span` scan positions */
  int32_t scan_span;
  int32_t intra_sub_prob; /* intra MB uses per-block predicted modes (DecIntraSubBlockPMode) */
  int32_t plane_prob;     /* intra: use the plane predictors where legal, per mille */
  int32_t intra_dc_only;  /* 1: I-frames use DC prediction only (SURVEY 8d frame-0 recipe) */
  int32_t edge_mode;      /* 0: MC windows stay inside the picture; 1: may touch stride padding / wrap rows */
  int32_t escape_prob;    /* per coefficient: force one of the three escape forms, per mille */
  int32_t qdelta_prob;    /* per P-frame: non-zero quantizer delta, per mille */
  int32_t table1_prob;    /* per I-frame: select residual VLC table 1, per mille */
  int32_t lowfreq_prob;   /* per coded block: 1..2 levels within the first 3 scan positions (what the reference's IDCT1P / IDCT3P classes take,
                             MD.cs:2939-2940) instead of the draw above, per mille; 0 = never (and no random number is drawn for it) */
} mobi_gen_params;

/* Fill `p` with the SURVEY.md 8(d) distribution for config 'A' (256x192 ModsDS),
 * 'B' (640x480 Moflex3DS) or 'C' (848x480 Moflex3DS, large MVs + dense 8x8). */
void mobi_gen_default_params(mobi_gen_params *p, int config, uint64_t seed);

/* Generate a clip.  out: bitstream bytes of all frames back to back (each frame a whole
 * number of 16-bit words); frame_off[0..n_frames] byte offsets.  Returns total bytes,
 * or -(needed bytes) when cap is too small, or -1 on bad parameters. */
int64_t mobi_gen_clip(const mobi_gen_params *p, uint8_t *out, size_t cap, uint32_t *frame_off);

#ifdef __cplusplus
}
#endif
#endif
