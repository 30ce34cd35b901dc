// mobi_state.h -- the decoder state that survives a frame, in the form both parse sides exchange (r05).
//
// A clip's frames may be parsed on the GPU (mobi_dparse.hip, mobi_lsparse.hip) or by the host parser (mobi_parse.cpp), and the answer must
// not depend on which: the device parsers handle frames that decode WITHOUT INCIDENT, and every other frame -- every condition under which the
// reference throws, every walk through Internal[] (MD.cs:3424-3429), every value the command list has to escape -- is parsed again by the host
// parser inside the same call, from the decoder state the clip had when that frame started.  That state is these two records: MobiDevState
// (what the device parsers themselves read and write) and MobiDevTail (the words of Internal[] only a walk can read: the coefficient block
// Internal[90..153] and the transforms' scratch Internal[154..217] as the last blocks of the last frame left them, and the MV row cache
// Internal[221..]).  The tail is not tracked inside the parsers' token loops: a clean frame's command list says what its last blocks were,
// and mobi_tail_* below rebuild the words from it after the parse (mobi_parse_tail in mobi_dparse.hip; the same functions on the CPU in
// tests/test_parse_fallback.py, against the host parser's own bookkeeping and the oracle's Internal[]).
#ifndef MOBI_STATE_H
#define MOBI_STATE_H
#include <stdint.h>

#include "mobi_cmd.h"
#include "mobi_recon_math.h"

struct MobiDevState {
  uint32_t quant;         // Quantizer (MD.cs:26)
  uint32_t yuvfmt;        // YuvFormat (MD.cs:27)
  int32_t frames_started; // how many ring slots hold a frame
  uint32_t tables_set;    // SetupQuantTables ran at least once (MD.cs:3884): the zigzag bytes of Internal[10..89] are valid
  uint8_t mcache[40];     // bytes of Internal[0..9]: intra-mode neighbour cache (MD.cs:1840-1859)
  int32_t predx, predy;   // Internal[219], [220]: the MV predictor of the last macroblock of the last P-frame (MD.cs:207-208)
};
#define MOBI_TAIL_MVC 132 /* 2 * (64 + 2): the MV row cache of the widest picture (Internal[221..], MD.cs:163-208) */
struct MobiDevTail {
  uint32_t ib[64];              // Internal[90..153]
  uint32_t scratch[64];         // Internal[154..217]
  int32_t mvc[MOBI_TAIL_MVC];   // Internal[221..]
  uint32_t pad[4];
};
#if defined(__cplusplus)
static_assert(sizeof(MobiDevState) == 64 && sizeof(MobiDevTail) == 1056, "decoder state records");
#endif

#if defined(__HIPCC__)
#define MOBI_ST_FN static __host__ __device__ __forceinline__
#else
#define MOBI_ST_FN static inline
#endif

// ---- what the transforms leave behind (MD.cs:3435-3798) -------------------------------------------------------------------------------
// IDCT64Px8 writes its first pass, transposed, to Internal[154 + 8m + k] (MD.cs:3452-3500); IDCT16Px8 the first pass of the top-left 4x4
// coefficients to Internal[154 + 4m + k], k < 4 (:3577-3612: its butterfly is the 8-point one on a zero-extended group); IDCT3Px8 keeps
// its first pass in Internal[90..97] (:3661-3707); IDCT16Px4 writes its first pass to Internal[106 + 4m + k] (:3728-3760); IDCT1Px8 /
// IDCT1Px4 write nothing.  c = the coefficient block as the transform finds it.
MOBI_ST_FN void mobi_scratch_from64(const uint32_t c[64], uint32_t scratch[64]) {
  for (int k = 0; k < 8; k++) {
    int in[8], out[8];
    for (int m = 0; m < 8; m++) in[m] = (int)c[8 * k + m];
    if (k == 0) in[0] += 0x20;
    mobi_bfly8(in, out);
    for (int m = 0; m < 8; m++) scratch[8 * m + k] = (uint32_t)out[m];
  }
}
MOBI_ST_FN void mobi_scratch_from16(const uint32_t c[64], uint32_t scratch[64]) { // (words 32..63 keep what they held)
  for (int k = 0; k < 4; k++) {
    int in[8] = {(int)c[8 * k], (int)c[8 * k + 1], (int)c[8 * k + 2], (int)c[8 * k + 3], 0, 0, 0, 0}, out[8];
    if (k == 0) in[0] += 0x20;
    mobi_bfly8(in, out);
    for (int m = 0; m < 8; m++) scratch[4 * m + k] = (uint32_t)out[m];
  }
}
MOBI_ST_FN void mobi_ib_after3(uint32_t ib[64]) { // IDCT3Px8 (MD.cs:3661-3707)
  const int r8 = (int)ib[0] + 32, r9 = (int)ib[1];
  const int r7 = r9 + (r9 >> 1), r11 = r7 >> 2, r3 = r9 + ((-r9) >> 2), r5 = r9 + (r9 >> 2);
  ib[0] = (uint32_t)(r8 + r7); ib[7] = (uint32_t)(r8 - r7);
  ib[1] = (uint32_t)(r8 + r5); ib[6] = (uint32_t)(r8 - r5);
  ib[2] = (uint32_t)(r8 + r3); ib[5] = (uint32_t)(r8 - r3);
  ib[3] = (uint32_t)(r8 + r11); ib[4] = (uint32_t)(r8 - r11);
}
MOBI_ST_FN void mobi_ib_after16x4(uint32_t ib[64]) { // IDCT16Px4's first pass (MD.cs:3728-3760)
  for (int k = 0; k < 4; k++) {
    int in[4] = {(int)ib[4 * k] + (k == 0 ? 0x20 : 0), (int)ib[4 * k + 1], (int)ib[4 * k + 2], (int)ib[4 * k + 3]}, out[4];
    mobi_bfly4(in, out);
    for (int m = 0; m < 4; m++) ib[16 + 4 * m + k] = (uint32_t)out[m];
  }
}
// the transform variant the reference dispatches by the final scan index (MD.cs:2939-2942, 2954-2955, 2966-2967); p = tokens' final
// position inside the block (r12 - 10 resp. r12 - 74)
enum { MOBI_V1 = 0, MOBI_V3 = 1, MOBI_V16 = 2, MOBI_VALL = 3 };
MOBI_ST_FN int mobi_variant(bool is8, int p) { return is8 ? (p <= 1 ? MOBI_V1 : p <= 3 ? MOBI_V3 : p <= 10 ? MOBI_V16 : MOBI_VALL) : (p <= 1 ? MOBI_V1 : MOBI_VALL); }

// ---- the tail of a CLEAN frame from its command list ------------------------------------------------------------------------------------
// In a frame that decoded without incident every token of a residual block is a level word in the payload (the device parsers hand
// frames with a zero-valued token to the host parser), in decode order, and its position field names the zigzag target: the block's
// final scan index is the inverse zigzag of its last word + 1.  What the frame leaves in Internal[90..217] follows from its last blocks
// only -- the last block of all ([90..105]), the last 8x8 block ([106..153]), the last 4x4 block with a full transform behind that
// ([106..121]), the last 8x8 block with a full transform ([154..217]) and the last one with the 16-coefficient transform behind that
// ([154..185]) -- so the scan runs backwards over the macroblocks and stops at the first full 8x8 transform it meets.
struct MobiTailScan {
  // block records: [21:0] word offset of the block's first level word inside the clip's payload (< 8191 macroblocks x 448 words), [28:22] words, [31] valid
  uint32_t last, last8, last4all, last8all, last8v16;
  uint32_t last_is8, last_variant; // of `last`
  bool done;                       // a full 8x8 transform was met: nothing earlier matters
};
MOBI_ST_FN void mobi_tail_scan_init(MobiTailScan &s) { s.last = s.last8 = s.last4all = s.last8all = s.last8v16 = 0; s.last_is8 = s.last_variant = 0; s.done = false; }
// One macroblock, called for macroblocks in REVERSE raster order.  w = its level words (n of them, behind the records / the cell map),
// woff = their word offset inside the clip's payload, t8 = MbDesc.w1's 8x8 mask, izz8 / izz4 = the inverse zigzag tables.
MOBI_ST_FN void mobi_tail_scan_mb(MobiTailScan &s, const uint32_t *w, int n, uint32_t woff, uint32_t t8, const uint8_t *izz8, const uint8_t *izz4) {
  int end = n;
  while (end > 0 && !s.done) { // blocks from the last to the first: a block = a run of words of one (area, 4x4 sub-block)
    const uint32_t t = w[end - 1] & 0x1FF, area = t >> 6;
    const bool is8 = (t8 >> area) & 1;
    const uint32_t key = is8 ? area << 2 : (area << 2) | ((t >> 4) & 3);
    int beg = end - 1;
    while (beg > 0) {
      const uint32_t u = w[beg - 1] & 0x1FF, ua = u >> 6;
      if ((is8 ? ua << 2 : (ua << 2) | ((u >> 4) & 3)) != key) break;
      beg--;
    }
    const int p = (is8 ? izz8[t & 63] : izz4[t & 15]) + 1, variant = mobi_variant(is8, p);
    const uint32_t rec = 0x80000000u | ((uint32_t)(end - beg) << 22) | (woff + (uint32_t)beg);
    if (!s.last) { s.last = rec; s.last_is8 = is8; s.last_variant = (uint32_t)variant; }
    if (is8) {
      if (!s.last8) s.last8 = rec;
      if (variant == MOBI_V16 && !s.last8v16) s.last8v16 = rec;
      if (variant == MOBI_VALL) { s.last8all = rec; s.done = true; }
    } else if (variant == MOBI_VALL && !s.last8 && !s.last4all) s.last4all = rec;
    end = beg;
  }
}
// pay = the clip's payload, scale = the frame's dequant scales by natural index (80 words: mobi_build_scale_table of the frame's quantiser)
MOBI_ST_FN void mobi_tail_store(uint32_t *c, int n_zero, uint32_t rec, const uint32_t *pay, const int32_t *scale, bool is8) {
  for (int i = 0; i < n_zero; i++) c[i] = 0;
  const uint32_t *w = pay + (rec & 0x3FFFFFu);
  for (uint32_t i = 0, n = (rec >> 22) & 0x7F; i < n; i++) {
    const uint32_t p = is8 ? w[i] & 63 : w[i] & 15;
    c[p] = (uint32_t)scale[is8 ? p : 64 + p] * (uint32_t)(int32_t)(int16_t)(w[i] >> 16); // (dequant word >> 8) * level, MD.cs:3427-3429
  }
}
MOBI_ST_FN void mobi_tail_finish(const MobiTailScan &s, const uint32_t *pay, const int32_t *scale, const MobiDevTail &in, MobiDevTail &out) {
  for (int i = 0; i < 64; i++) { out.ib[i] = in.ib[i]; out.scratch[i] = in.scratch[i]; }
  if (s.last8) mobi_tail_store(out.ib, 64, s.last8, pay, scale, true);            // MD.cs:2933-2936: all 64 words zeroed, then the stores
  if (s.last4all) {                                                                  // a full 4x4 transform behind it: its first pass in [106..121]
    uint32_t c[64];
    for (int i = 16; i < 32; i++) c[i] = 0;
    mobi_tail_store(c, 16, s.last4all, pay, scale, false);
    mobi_ib_after16x4(c);
    for (int i = 16; i < 32; i++) out.ib[i] = c[i];
  }
  if (s.last && !s.last_is8) mobi_tail_store(out.ib, 16, s.last, pay, scale, false); // the last block of all was a 4x4 one: [90..105] are its
  if (s.last && s.last_is8 && s.last_variant == MOBI_V3) mobi_ib_after3(out.ib);
  if (s.last8all) {
    uint32_t c[64];
    mobi_tail_store(c, 64, s.last8all, pay, scale, true);
    mobi_scratch_from64(c, out.scratch);
  }
  if (s.last8v16) {
    uint32_t c[64];
    mobi_tail_store(c, 64, s.last8v16, pay, scale, true);
    mobi_scratch_from16(c, out.scratch);
  }
}
#endif
