// mobi_lsparse.h -- the bitstream parser as a LOCK-STEP state machine: one clip per LANE, 1 .. 64 clips per wave.
//
// mobi_dparse.hip runs the syntax walk of mobi_parse.cpp on one lane of a wave: 1060 vector + 800 scalar instructions per macroblock for
// one clip, the other 63 lanes idle.  Here every lane of a wave owns a clip and all of them run the SAME instruction stream: the walk is cut
// into regions (macroblock start, partition node, coded-block pattern, intra header / areas, next block, one residual token, macroblock
// end) laid out in the order a macroblock passes through them; every round of the loop each lane executes the regions its state lets it
// enter, falling through from one to the next, and a region costs the wave its instructions once however many lanes are in it.  The lanes
// are never synchronised on macroblocks: each walks its own frame; the number of rounds is the longest lane's, and the law of large numbers
// keeps that close to the mean (a frame is 1200 macroblocks).  Behind every round of the whole walk come cheap ones of "one token" only
// (ls_round: ls_token_fast -- which also finds an inter macroblock's next block wherever that reads at most one bit): tokens are most of
// what a frame consists of.
//
// This is the FAST path only.  It produces exactly what mobi_parse_frames produces (descriptors, payload, intra items, result record,
// persistent state) for streams that decode without incident; at anything else -- every condition under which the reference throws, a
// refusal, a Elias-gamma code longer than 15 bits, data that ends inside the frame, an invalid table entry -- the lane BAILS OUT: it stops,
// nothing of its persistent state is written, and mobi_parse_frames (which knows the reference's behaviour at every one of those) parses that
// clip again from the same state.  The dependency lists of intra macroblocks (MbDesc.w4..w7) are not made here: they depend on geometry and
// macroblock types only, so a second, fully parallel kernel fills them in (ls_intra_deps below).
//
// The bit reader is not the reference's 32-bit window with 16-bit refills (MD.cs:2970-3015) but a 64-bit one refilled 32 bits at a time
// from a per-lane ring in LDS; what the reference would have read is the same bits, and the Offset it would leave follows from the number
// of bits consumed: it refills lazily, one 16-bit word whenever fewer than 16 bits remain, so after c bits it has fetched 1 + ceil(c / 16)
// words (every syntax element that consumes more than 16 bits between two of its checks is a bail-out here).
//
// Host-compilable: tests/tools/mobi_lsparse_host.cpp runs the same functions lane by lane on the CPU against mobi_parse.cpp.
#ifndef MOBI_LSPARSE_H
#define MOBI_LSPARSE_H
#include <stdint.h>

#include "../../include/mobiclip_hip.h"
#include "mobi_cmd.h"
#include "mobi_dparse_tables.h"

#if defined(__HIPCC__)
#define LS_FN __host__ __device__ __forceinline__
#else
#define LS_FN static inline
#endif

enum { LS_DONE = 0, LS_MB_BEGIN, LS_NODE, LS_P_CBP, LS_I_HDR, LS_I_SUBAREA, LS_I_SUB4, LS_I_CHROMA, LS_I_FIXED, LS_I_FSUB, LS_NEXT, LS_TOKEN, LS_MB_END,
       LS_NEXT_SLOW, LS_TOKEN_SLOW }; // what the cheap rounds leave to the whole walk: a 4x4 area's pattern, an escape token, anything odd
#define LS_MAGIC MOBI_LS_MAGIC
#ifndef LS_K
#define LS_K 3 // A/B on one box (tools/exp_lsab.sh, 24576 clips, 12 per wave): 2: 24.84, 3: 24.45, 4: 24.70, 5: 25.25 ms per P-frame step
#endif
enum { LS_TOKEN_ROUNDS = LS_K,  // ls_token_fast() on its own this many times behind every round of the whole walk
       LS_ROUND_BYTES = 64, // what one round can take from the ring at most (below)
       LS_RING = 128 };      // bytes of bitstream per lane in LDS
// ls_refill() reads the ring without looking at the write pointer: a round must not be able to take more than LS_ROUND_BYTES from it, or a
// mis-tuned build would parse stale ring bytes and still call the clip finished -- so such a build does not compile.  A round
// is the macroblock part, the intra part, "next block", LS_K + 1 token parts; the most a lane can take on its way through them:
//   an intra macroblock   a partition code that says "intra" (12 bits) + the header and every area's mode in one visit of the intra part
//                         (~310) + a token (28; what follows a block of an intra macroblock reads nothing here) per token part
//   an inter macroblock   a partition code and two vector components (12 + 2 x 15), the coded-block pattern and the first area's flag or
//                         4x4 pattern (15 + 15), "next block" (15), and per token part a token and the next area's flag or pattern (28 + 15)
#if !defined(LS_CHEAP_NT)
static_assert(12 + 310 + (LS_K + 1) * 28 <= LS_ROUND_BYTES * 8 && 72 + 15 + (LS_K + 1) * 43 <= LS_ROUND_BYTES * 8, "LS_K: a round could outrun the bitstream ring");
#else
static_assert(12 + 310 + (LS_K + 1) * 43 <= LS_ROUND_BYTES * 8 && 72 + (LS_K + 1) * (15 + 43) <= LS_ROUND_BYTES * 8, "LS_K: a round could outrun the bitstream ring");
#endif

struct LsCtx { // wave-uniform
  const uint8_t *T; // the table blob (mobi_dparse_tables.h)
  int width, height, stride, lg, mbw, mbh, n_mbs, version;
  uint32_t pay_cap;
};

struct LsLane {
  // bit reader: W holds the next navail bits of the stream at its top
  uint64_t W;
  int navail;
  uint32_t cbits; // bits consumed since Data[Offset]
  uint32_t nxt;   // the 32 bits behind the window
  uint32_t rd;    // stream byte offset of the four bytes behind those
  // state machine
  int st, ret, bail;
  // frame
  uint32_t quant, yuvfmt, tables_set;
  int vlc, frames_started, iframe;
  // macroblock
  int mb, mx, my, cur_off, mb_type, predx, predy;
  // The MV row cache (Internal[221..], MD.cs:145-208: macroblock mx reads entries mx, mx + 1, mx + 2 -- left, top, top right -- and owns entry
  // mx + 1), r06: NOT in LDS.  A vector is dx | dy << 16 (both within +-8191 in any frame a lane finishes).  What a macroblock needs sits in
  // registers -- mv_left (the macroblock before it in the row), mv_a / mv_b (entries mx + 1, mx + 2 as the row above left them) -- the entry
  // it owns (mv_cur) goes to the store's global copy when the macroblock ends, and the one word per macroblock that has to come back from
  // there (entry mx + 3 of the row above: the NEXT macroblock's top right) is asked for a macroblock ahead (mv_pref).  The first two entries
  // of the row being written (row_e1, row_e2) are the next row's first top and top right.
  uint32_t mv_left, mv_a, mv_b, mv_cur, mv_pref, row_e1, row_e2;
  uint32_t pay_pos, mb_pay, hdr_words, n_coefs, cbp6, t8mask, w3, n_items;
  // partition tree
  int sp, nleaf;
  uint32_t l0a, l0b, l1a, l1b;
  // walk over the coded blocks
  uint32_t area_mask, sub_mask;
  int cur_area;
  int blk_p, blk_n, blk_tile;
  uint32_t blk_flags;
  // intra walk
  uint32_t i_cbp, i_m4;
  int i_k, i_mode, i_sub, i_subkind, i_chroma;
  // where this clip's output goes
  MbDesc *desc;
  uint32_t *pay;      // the arena, or this clip's part of it (pay_base = 0 then: see MobiDevParseArgs.pay_local)
  uint32_t pay_base, clip;
  uint32_t *items;
};

// four equal words at a 4-byte aligned address: one store on the device
#if defined(__HIP_DEVICE_COMPILE__)
typedef uint32_t ls_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
#define LS_STORE4(p, v) (*(ls_u32x4 *)(p) = ls_u32x4{(v), (v), (v), (v)})
#else
#define LS_STORE4(p, v) ((p)[0] = (p)[1] = (p)[2] = (p)[3] = (v))
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(LS_INTRA_NOUNROLL)
#define LS_INTRA_LOOP_PRAGMA _Pragma("unroll 1")
#else
#define LS_INTRA_LOOP_PRAGMA
#endif
// "some lane of the wave" (on the CPU: this lane)
#if defined(__HIP_DEVICE_COMPILE__)
#define LS_ANY(cond) (__builtin_amdgcn_ballot_w64(cond) != 0)
#else
#define LS_ANY(cond) (cond)
#endif
LS_FN int ls_clz(uint32_t v) { return v ? __builtin_clz(v) : 32; }
LS_FN int ls_ctz(uint32_t v) { return __builtin_ctz(v); }
LS_FN int ls_min(int a, int b) { return a < b ? a : b; }
LS_FN int ls_max(int a, int b) { return a > b ? a : b; }
LS_FN void ls_bail(LsLane &s, int why) {
  if (!s.bail) s.bail = why;
  s.st = LS_DONE;
}
LS_FN uint32_t ls_win(const LsLane &s) { return (uint32_t)(s.W >> 32); }
LS_FN void ls_take(LsLane &s, int k) { // 0 <= k <= 32
  s.W <<= k;
  s.navail -= k;
  s.cbits += (uint32_t)k;
}
// more than 32 bits in the window afterwards (the ring always holds what an iteration can ask for: see the kernel's ring service)
template <class S>
LS_FN void ls_refill(LsLane &s, S &m) {
  if (s.navail <= 32) { // the 32 bits behind the window wait in a register; the ones behind those are asked for now and not looked at before the next refill
    s.W |= (uint64_t)s.nxt << (32 - s.navail);
    s.navail += 32;
    const uint32_t d = m.ring32(s.rd); // two 16-bit little-endian words, the first one's bits first (MD.cs:2978-2990)
    s.nxt = (d << 16) | (d >> 16);
    s.rd += 4;
  }
}
// Elias-gamma (MD.cs:2992-3015).  Codes of more than 15 bits: bail out (the reference's reader refills once per code)
LS_FN uint32_t ls_ue(LsLane &s) {
  const uint32_t w = ls_win(s);
  const int z = ls_clz(w);
  if (z >= 8) { ls_bail(s, 1); return 0; }
  const uint32_t v = (z ? ((w << (z + 1)) >> (32 - z)) : 0u) + (1u << z) - 1u;
  ls_take(s, 2 * z + 1);
  return v;
}
LS_FN int ls_se(LsLane &s) {
  const uint32_t w = ls_win(s);
  const int z = ls_clz(w);
  if (z >= 8) { ls_bail(s, 1); return 0; }
  const uint32_t u = (z ? ((w << (z + 1)) >> (32 - z)) : 0u) + (1u << z);
  int v = (int)u;
  if (v & 1) v = (int)(1u - u);
  v >>= 1;
  ls_take(s, 2 * z + 1);
  return v;
}

// The partition-code table of this parser's copy of the blob: entry = code | code length << 4, or 0xFF where the reference throws (a code the
// shape does not have) -- one look-up instead of three behind one another.  i = 0 .. 1023, every entry on its own (any order, any thread).
LS_FN void ls_prepare_tables(uint8_t *T, int i) {
  const uint32_t code = T[MOBI_DT_PLUT + i], sh = (uint32_t)i >> 6;
  const bool ok = code < T[MOBI_DT_PNB + sh] && code <= 9;
  T[MOBI_DT_PLUT + i] = ok ? (uint8_t)(code | (T[MOBI_DT_PBITS + sh * 12 + (ok ? code : 0)] << 4)) : (uint8_t)0xFF;
}

// ---------------------------------------------------------------- frame header (MD.cs:113-143, :224-236)
template <class S>
LS_FN void ls_setup_quant(LsLane &s, S &m, const LsCtx &c, uint32_t q) { // MD.cs:3884-3925
  if (c.version == MOBI_VERSION_MOFLEX3DS) q = q < 12 ? 12 : q > 52 ? 52 : q;
  s.quant = q;
  if (q >= 54) { ls_bail(s, 2); return; }
  s.tables_set = 1;
  m.mc(1) = 9; m.mc(2) = 9; m.mc(3) = 9; m.mc(4) = 9;
  m.mc(8) = 9; m.mc(0x10) = 9; m.mc(0x18) = 9; m.mc(0x20) = 9;
}
// s.quant, yuvfmt, tables_set, frames_started (already advanced) and the mode cache are loaded; the ring holds the first 64 bytes
template <class S>
LS_FN void ls_begin_frame(LsLane &s, S &m, const LsCtx &c, uint32_t len) {
  s.W = 0; s.navail = 0; s.cbits = 0; s.rd = 0;
  s.bail = 0; s.st = LS_MB_BEGIN; s.ret = LS_DONE;
  s.mb = 0; s.mx = 0; s.my = 0; s.cur_off = 0; s.mb_type = MOBI_MB_INTER; // (predx, predy: loaded with the state, Internal[219], [220])
  s.mv_left = s.mv_a = s.mv_b = s.mv_cur = s.mv_pref = s.row_e1 = s.row_e2 = 0;
  s.pay_pos = 0; s.mb_pay = 0; s.hdr_words = 0; s.n_coefs = 0; s.cbp6 = s.t8mask = s.w3 = 0; s.n_items = 0;
  s.sp = 0; s.nleaf = 0; s.l0a = s.l0b = s.l1a = s.l1b = 0;
  s.area_mask = s.sub_mask = 0; s.cur_area = 0; s.blk_p = s.blk_n = s.blk_tile = 0; s.blk_flags = 0;
  s.i_cbp = s.i_m4 = 0; s.i_k = s.i_mode = s.i_sub = s.i_subkind = s.i_chroma = 0;
  s.vlc = 0; s.iframe = 0;
  if (len < 2) { ls_bail(s, 3); return; }
  s.nxt = 0;
  ls_refill(s, m); // (primes nxt)
  s.navail = 0;
  ls_refill(s, m);
  ls_refill(s, m);
  s.iframe = (int)(ls_win(s) >> 31);
  ls_take(s, 1);
  if (s.iframe) {
    s.yuvfmt = ls_win(s) >> 31;
    s.vlc = (int)((ls_win(s) >> 30) & 1);
    ls_take(s, 2);
    const uint32_t q = ls_win(s) >> 26;
    ls_take(s, 6);
    if (s.quant != q) ls_setup_quant(s, m, c, q);
  } else {
    const uint32_t q = s.quant;
    const int dq = ls_se(s);
    if (s.bail) return;
    if (c.version == MOBI_VERSION_MOFLEX3DS && q == 0) ls_setup_quant(s, m, c, q);
    else if (dq != 0) ls_setup_quant(s, m, c, q + (uint32_t)dq);
    s.vlc = 0;
    // (the cache is zeroed here in the reference, MD.cs:144-154: the top row reads zeros without looking -- my == 0 -- every entry 1 .. mbw is
    // written before a later row reads it, and the two that no macroblock owns are written now: what the frame leaves is the whole cache)
    m.mvp_store(0, 0);
    m.mvp_store(c.mbw + 1, 0);
  }
}

// ---------------------------------------------------------------- pieces of the walk
LS_FN int ls_area_offset(const LsLane &s, const LsCtx &c, int area, int sub) {
  const int S = c.stride;
  const int o = (area < 4) ? s.cur_off + (area >> 1) * 8 * S + (area & 1) * 8 : s.cur_off / 2 + (area == 5 ? S / 2 : 0);
  return o + (sub >> 1) * 4 * S + (sub & 1) * 4;
}
LS_FN void ls_check_intra_reads(LsLane &s, const LsCtx &c, int mode, int o) { // the reference indexes below the plane: it throws
  const uint32_t top = 0x1E5, left = 0x0F6;
  if ((((top >> mode) & 1) && o < c.stride) || (((left >> mode) & 1) && o < 1)) ls_bail(s, 4);
}
LS_FN void ls_block(LsLane &s, int area, int sub, bool is8) { // one transform block's tokens follow (MD.cs:3330)
  if (s.quant < 12) { ls_bail(s, 5); return; }
  s.blk_p = 0;
  s.blk_n = is8 ? 64 : 16;
  s.blk_tile = is8 ? area * 64 : area * 64 + sub * 16;
  s.blk_flags = (is8 ? 1u : 0u) | (s.vlc == 1 ? 2u : 0u) | (s.tables_set ? 4u : 0u);
}
// predicted-mode code (MD.cs:1840-1859, 2785-2804, 2841-2858)
template <class S>
LS_FN int ls_pmode(LsLane &s, S &m, int ci, bool four) {
  int pred = ls_min(m.mc(ci - 8), m.mc(ci - 1));
  if (pred == 9) pred = 3;
  int v = (int)(ls_win(s) >> 28), nb = 1, mode = pred;
  if (v >= pred) v++;
  if (v < 9) { mode = v; nb = 4; }
  if (four) m.mc(ci) = (uint8_t)mode;
  else m.mc(ci) = m.mc(ci + 1) = m.mc(ci + 8) = m.mc(ci + 9) = (uint8_t)mode;
  ls_take(s, nb);
  return mode;
}
// one motion-compensated leaf (MD.cs:400-456): would CopyBlock throw?  then bail out.  Its cells go straight into the cell map when the
// macroblock is split (the map is only kept if it turns out to be more than two halves)
// one leaf's cells of the 8 x 8 MV cell map at the head of the macroblock's payload (a row of the leaf at a time: 8, 4, 2 or 1 cells; the lanes
// of a wave run as many rows as the tallest leaf has)
LS_FN void ls_cells(LsLane &s, int x, int y, int wi, int hi, uint32_t cell) {
  const int w = 16 >> wi, h = 16 >> hi;
  uint32_t *cells = s.pay + s.pay_base + s.mb_pay + (y >> 1) * 8 + (x >> 1);
  for (int r = 0; r < (h >> 1); r++, cells += 8) {
    if (w == 16) { LS_STORE4(cells, cell); LS_STORE4(cells + 4, cell); }
    else if (w == 8) LS_STORE4(cells, cell);
    else if (w == 4) { cells[0] = cell; cells[1] = cell; }
    else cells[0] = cell;
  }
}
template <class S>
LS_FN void ls_leaf(LsLane &s, S &m, const LsCtx &c, int wi, int hi, int x, int y, int ref, int dx, int dy) {
  const int w = 16 >> wi, h = 16 >> hi, S_ = c.stride;
  s.mv_cur = mobi_leaf_w1(dx, dy); // (the row cache holds a vector as two int16 in one word -- a vector beyond +-8191 ends the lane right below)
  if (ref > ls_min(5, s.frames_started - 1)) { ls_bail(s, 6); return; }
  if (dx < -MOBI_MV_LIMIT || dx > MOBI_MV_LIMIT || dy < -MOBI_MV_LIMIT || dy > MOBI_MV_LIMIT) { ls_bail(s, 7); return; }
  const int o = s.cur_off + y * S_ + x, ylen = S_ * c.height;
  const int pos = o + (dy >> 1) * S_ + (dx >> 1);
  const int hi_y = pos + (h - 1) * S_ + w - 1 + (dx & 1) + ((dy & 1) ? S_ : 0);
  const int cdx = dx >> 1, cdy = dy >> 1;
  const int cpos = o / 2 + (cdy >> 1) * S_ + (cdx >> 1);
  const int hi_c = cpos + S_ / 2 + ((h >> 1) - 1) * S_ + (w >> 1) - 1 + (cdx & 1) + ((cdy & 1) ? S_ : 0);
  if (pos < 0 || hi_y >= ylen || cpos < 0 || hi_c >= ylen / 2) { ls_bail(s, 8); return; }
  const uint32_t w0 = mobi_leaf_w0(x, y, wi, hi, ref), w1 = mobi_leaf_w1(dx, dy);
  const bool first = s.nleaf == 0, second = s.nleaf == 1; // (selects, not stores behind a compare: those became an indexed store into scratch)
  s.l0a = first ? w0 : s.l0a; s.l0b = first ? w1 : s.l0b;
  s.l1a = second ? w0 : s.l1a; s.l1b = second ? w1 : s.l1b;
  s.nleaf++;
#ifdef LS_CELLS_ALWAYS
  if (wi | hi) ls_cells(s, x, y, wi, hi, mobi_cell(dx, dy, ref));
#else
  // The MV cell map is only read for deeper trees: a macroblock of one or two leaves travels as leaf records (MbDesc, hdr_words = 0: its
  // level words start where the map would).  Two leaves are 44 % of the macroblocks, and their cells -- 4 or 8 rows of stores that the
  // whole wave steps through -- were written for nothing: they wait in the lane's two leaf records until a third leaf says the map is needed.
  if (s.nleaf >= 3) {
    if (s.nleaf == 3) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
      for (int k = 0; k < 2; k++) {
        const uint32_t a = k ? s.l1a : s.l0a, b = k ? s.l1b : s.l0b;
        ls_cells(s, (int)(a & 15) * 2, (int)((a >> 4) & 15) * 2, (int)((a >> 8) & 3), (int)((a >> 10) & 3),
                 mobi_cell((int)(int16_t)(b & 0xFFFF), (int)(int16_t)(b >> 16), (int)((a >> 12) & 7)));
      }
    }
    ls_cells(s, x, y, wi, hi, mobi_cell(dx, dy, ref));
  }
#endif
}
LS_FN int ls_classify(const LsLane &s) {
  if (s.nleaf != 2) return MOBI_DUAL_NONE;
  const uint32_t a = s.l0a & 0xFFF, b = s.l1a & 0xFFF;
  if (a == (0u | (1u << 10)) && b == ((4u << 4) | (1u << 10))) return MOBI_DUAL_TB;
  if (a == (0u | (1u << 8)) && b == (4u | (1u << 8))) return MOBI_DUAL_LR;
  return MOBI_DUAL_NONE;
}

// ---- one residual token (MD.cs:3330-3432).  Also called on its own (ls_token_rounds): tokens are half of all syntax elements, and a round of
// the whole walk costs the wave twenty times what this region does ----
template <class S>
LS_FN void ls_token(LsLane &s, S &m, const LsCtx &c) {
  const uint8_t *T = c.T;
  const bool mine = s.st == LS_TOKEN_SLOW; // (LS_TOKEN itself belongs to ls_token_fast)
  if (mine) s.st = LS_TOKEN;
  if (mine) {
    ls_refill(s, m);
    const uint16_t *A = (const uint16_t *)(T + ((s.blk_flags & 2) ? MOBI_DT_A1 : MOBI_DT_A0));
    const uint8_t *B = T + ((s.blk_flags & 2) ? MOBI_DT_B1 : MOBI_DT_B0);
    int skip = 0, value = 0, len = 0;
    uint32_t e = 0, last = 0;
    bool raw = false;
    uint32_t w = ls_win(s);
    int esc = 0; // 0: plain, 1: "0" level escape, 2: "10" run escape
    if ((w >> 25) == 3) {
      if (!((w >> 24) & 1)) { esc = 1; ls_take(s, 8); }
      else if (!((w >> 23) & 1)) { esc = 2; ls_take(s, 9); }
      else { raw = true; ls_take(s, 9); }
      w = ls_win(s);
    }
    if (raw) { // last(1) run(6) level(s12)
      last = w >> 31;
      skip = (int)((w >> 25) & 0x3F);
      value = (int32_t)(w << 7) >> 20;
      ls_take(s, 19);
    } else {
      e = A[w >> 20];
      len = (int)(e & 0xF);
      value = (int)((e >> 4) & 0x1F);
      skip = (int)((e >> 9) & 0x3F);
      last = e >> 15;
      if (esc == 1) value += B[e >> 9];
      if (esc == 2) skip += B[0x80 + value + ((e >> 15) << 6)];
      if (len == 0) ls_bail(s, 14);
      else {
        if ((w >> (32 - len)) & 1) value = -value;
        ls_take(s, len);
      }
    }
    if (value == 0) ls_bail(s, 17); // a token without a level: the command list would not name every token of the frame (mobi_state.h)
    if (!s.bail) {
      s.blk_p += skip;
      if (s.blk_p >= s.blk_n) ls_bail(s, 15);
      else {
        const int idx = (s.blk_flags & 4) ? T[((s.blk_flags & 1) ? MOBI_DT_ZZ8 : MOBI_DT_ZZ4) + s.blk_p] : 0;
        s.blk_p++;
        if (value != 0) s.pay[s.pay_base + s.mb_pay + s.hdr_words + s.n_coefs++] = (uint32_t)(s.blk_tile + idx) | ((uint32_t)(int)(int16_t)value << 16);
        if (last & 1) {
          s.st = s.ret;
          if (s.st == LS_NEXT && !s.sub_mask && !s.area_mask) s.st = LS_MB_END;
        }
      }
    }
  }
  if (s.bail) s.st = LS_DONE;
}

// ---- inter macroblock: the next coded block (loc_11652C, MD.cs:2909-2929).  Like ls_token() also called on its own ----
template <class S>
LS_FN void ls_next(LsLane &s, S &m, const LsCtx &c) {
  const uint8_t *T = c.T;
  const bool mine = s.st == LS_NEXT_SLOW; // (LS_NEXT itself belongs to ls_next_fast)
  if (mine) s.st = LS_NEXT;
  if (mine) {
    ls_refill(s, m);
    if (!s.sub_mask && s.area_mask) {
      const int a = ls_ctz(s.area_mask);
      s.area_mask &= s.area_mask - 1;
      if (ls_win(s) >> 31) {
        ls_take(s, 1);
        s.t8mask |= 1u << a;
        ls_block(s, a, 0, true);
        s.ret = LS_NEXT;
        if (!s.bail) s.st = LS_TOKEN;
      } else {
        const uint32_t u = ls_ue(s);
        if (!s.bail) {
          if (u >= 16) ls_bail(s, 12);
          else { s.sub_mask = T[MOBI_DT_CBP4_P + u]; s.cur_area = a; }
        }
      }
    }
    if (s.st == LS_NEXT) {
      if (s.sub_mask) {
        const int sub = ls_ctz(s.sub_mask);
        s.sub_mask &= s.sub_mask - 1;
        ls_block(s, s.cur_area, sub, false);
        s.ret = LS_NEXT;
        if (!s.bail) s.st = LS_TOKEN;
      } else if (!s.area_mask) s.st = LS_MB_END;
    }
  }
  if (s.bail) s.st = LS_DONE;
}

// ---- the cheap rounds: the two regions again, for the common case only and without a branch inside -- the next block when it is a 4x4
// block of a pattern already read or an area coded as one 8x8 transform, a plain table token -- and everything else handed to the whole
// walk (LS_NEXT_SLOW, LS_TOKEN_SLOW: ls_next / ls_token above, called from ls_step_main).  The general forms cost 240 vector + 175 scalar
// instructions per round in the compiler's hands (every `if` a compare, a saved mask, a branch); these are what a round is made of. ----
template <class S>
LS_FN void ls_next_fast(LsLane &s, S &m, const LsCtx &c) {
  if (s.st == LS_NEXT) {
    ls_refill(s, m);
    bool slow = s.quant < 12, got = false, is8 = false;
    int tile = 0;
    if (!s.sub_mask && s.area_mask && !slow) {
      const int a = ls_ctz(s.area_mask);
      const uint32_t w = ls_win(s);
      if (w >> 31) { // one 8x8 transform
        s.area_mask &= s.area_mask - 1;
        ls_take(s, 1);
        s.t8mask |= 1u << a;
        tile = a * 64;
        got = is8 = true;
      } else { // which of its 4x4 blocks are coded (MD.cs:2917-2927)
        const int z = ls_clz(w);
        const uint32_t u = (z ? ((w << (z + 1)) >> (32 - z)) : 0u) + (1u << z) - 1u;
        if (z >= 8 || u >= 16) slow = true;
        else {
          s.area_mask &= s.area_mask - 1;
          ls_take(s, 2 * z + 1);
          s.sub_mask = c.T[MOBI_DT_CBP4_P + u];
          s.cur_area = a;
        }
      }
    }
    if (!got && !slow && s.sub_mask) {
      const int sub = ls_ctz(s.sub_mask);
      s.sub_mask &= s.sub_mask - 1;
      tile = s.cur_area * 64 + sub * 16;
      got = true;
    }
    if (got) {
      s.blk_p = 0;
      s.blk_n = is8 ? 64 : 16;
      s.blk_tile = tile;
      s.blk_flags = (is8 ? 1u : 0u) | (s.vlc == 1 ? 2u : 0u) | (s.tables_set ? 4u : 0u);
      s.ret = LS_NEXT;
      s.st = LS_TOKEN;
    } else if (slow) s.st = LS_NEXT_SLOW;
    else if (!s.area_mask) s.st = LS_MB_END;
  }
}
template <class S>
LS_FN void ls_token_fast(LsLane &s, S &m, const LsCtx &c) {
  if (s.st == LS_TOKEN) {
    const uint8_t *T = c.T;
    ls_refill(s, m); // (more than 32 bits: the longest token, prefix included, has 28)
    const uint32_t w0 = ls_win(s);
    // 0: a table code; 1, 2: the escape prefix 0000011 + "0" / "10", a table code whose level / run grows by a second table's entry;
    // 3: prefix + "11", last(1) run(6) level(s12) spelled out (MD.cs:3346-3420)
    const int kind = (w0 >> 25) != 3 ? 0 : !((w0 >> 24) & 1) ? 1 : !((w0 >> 23) & 1) ? 2 : 3;
    const int pre = kind == 0 ? 0 : kind == 1 ? 8 : 9;
    const uint32_t w = (uint32_t)((s.W << pre) >> 32);
    const uint32_t e = ((const uint16_t *)(T + ((s.blk_flags & 2) ? MOBI_DT_A1 : MOBI_DT_A0)))[w >> 20];
    int len = (int)(e & 0xF), skip = (int)((e >> 9) & 0x3F), value = (int)((e >> 4) & 0x1F);
    uint32_t last = e >> 15;
    if (kind == 1 || kind == 2) { // (one look-up for both: the index differs)
      const int add = (T + ((s.blk_flags & 2) ? MOBI_DT_B1 : MOBI_DT_B0))[kind == 1 ? (int)(e >> 9) : 0x80 + value + (int)((e >> 15) << 6)];
      value += kind == 1 ? add : 0;
      skip += kind == 2 ? add : 0;
    }
    bool neg = len ? ((w >> (32 - len)) & 1) != 0 : false;
    if (kind == 3) {
      last = w >> 31;
      skip = (int)((w >> 25) & 0x3F);
      value = (int32_t)(w << 7) >> 20;
      neg = false;
      len = 19;
    }
    const int p = s.blk_p + skip;
    if (len == 0 || p >= s.blk_n || value == 0) s.st = LS_TOKEN_SLOW; // what the whole walk bails out on
    else {
      value = neg ? -value : value;
      ls_take(s, pre + len);
      const int idx = (s.blk_flags & 4) ? T[((s.blk_flags & 1) ? MOBI_DT_ZZ8 : MOBI_DT_ZZ4) + p] : 0;
      s.blk_p = p + 1;
      s.pay[s.pay_base + s.mb_pay + s.hdr_words + s.n_coefs++] = (uint32_t)(s.blk_tile + idx) | ((uint32_t)(int)(int16_t)value << 16);
      if (last & 1) {
        s.st = (s.ret == LS_NEXT && !s.sub_mask && !s.area_mask) ? LS_MB_END : (s.ret == LS_I_FSUB && !s.sub_mask) ? LS_I_FIXED : s.ret;
#ifndef LS_NO_FOLD
        // An inter macroblock's next block, where finding it reads at most one bit (ls_next_fast's first and third cases: the next 4x4 block
        // of a pattern already read, an area coded as one 8x8 transform): the lane stays with its tokens instead of waiting for a visit of
        // "next block" -- which served 2.4 lanes of 12 per visit and was a fifth of the walk's instructions (r05).  At least four bits are
        // left in the window: the refill above left more than 32 and the longest token has 28.
        if (s.st == LS_NEXT || s.st == LS_I_FSUB) { // (LS_I_FSUB: the next 4x4 block of an intra area whose pattern is read: sub_mask != 0)
          const bool more4 = s.sub_mask != 0;
          const bool next8 = !more4 && (ls_win(s) >> 31) != 0; // (area_mask != 0 here: the state would be LS_MB_END)
          const int a = ls_ctz(more4 ? s.sub_mask : s.area_mask); // (the masks are not both zero)
          s.sub_mask = more4 ? (s.sub_mask & (s.sub_mask - 1)) : s.sub_mask;
          s.area_mask = next8 ? (s.area_mask & (s.area_mask - 1)) : s.area_mask;
          s.t8mask |= next8 ? 1u << a : 0u;
          ls_take(s, next8 ? 1 : 0);
          s.blk_tile = more4 ? s.cur_area * 64 + a * 16 : next8 ? a * 64 : s.blk_tile;
          s.blk_n = next8 ? 64 : s.blk_n; // (more4: the block before was a 4x4 one of the same area)
          s.blk_flags = next8 ? (s.blk_flags | 1u) : s.blk_flags;
          s.blk_p = 0;
          s.st = (more4 || next8) ? LS_TOKEN : LS_NEXT;
#ifndef LS_NO_FOLD_PATTERN
          // ... and the third case, an area coded as 4x4 blocks: its pattern (MD.cs:2917-2927; up to 15 bits: a refill first) and the
          // first block of it.  With that "next block" has no visits of its own left in a stream that decodes without incident.
          if (LS_ANY(s.st == LS_NEXT)) {
            if (s.st == LS_NEXT) {
              ls_refill(s, m);
              const uint32_t w = ls_win(s); // (its first bit is 0)
              const int z = ls_clz(w);
              const uint32_t u = (z ? ((w << (z + 1)) >> (32 - z)) : 0u) + (1u << z) - 1u;
              const uint32_t pat = (z < 8 && u < 16) ? c.T[MOBI_DT_CBP4_P + u] : 0u;
              if (pat) { // (anything else -- a code the table does not have, a pattern without blocks -- is left to ls_next_fast / ls_next)
                const int ar = ls_ctz(s.area_mask), sub = ls_ctz(pat);
                s.area_mask &= s.area_mask - 1;
                ls_take(s, 2 * z + 1);
                s.cur_area = ar;
                s.sub_mask = pat & (pat - 1);
                s.blk_tile = ar * 64 + sub * 16;
                s.blk_n = 16;
                s.blk_flags &= ~1u;
                s.st = LS_TOKEN;
              }
            }
          }
#endif
        }
#endif
      }
    }
  }
}

// ---------------------------------------------------------------- one round of the walk, in three parts the kernel schedules separately:
// ls_step_main (macroblock end / start, partition node, inter CBP), ls_step_intra (the intra regions: half of the walk's instructions for
// one macroblock in twenty of a P-frame), and ls_next + ls_token above
template <class S>
LS_FN void ls_step_main(LsLane &s, S &m, const LsCtx &c) {
  const uint8_t *T = c.T;
  // ---- macroblock end: the descriptor (mobi_cmd.h).  First, so that a lane whose last token came in the extra rounds goes on to
  // the next macroblock in the same call ----
  if (s.st == LS_MB_END) {
    uint32_t w2 = s.n_coefs, w3 = s.w3, w4 = 0, w5 = 0, w6 = 0, w7 = 0, nl = 0;
    int dual = MOBI_DUAL_NONE;
    if (s.mb_type == MOBI_MB_INTER) {
      nl = (uint32_t)s.nleaf;
      dual = ls_classify(s);
      if (nl == 1 || dual) {
        // leaf records: positions and phases instead of motion vectors (MD.cs:400-416).  (No arrays here: an array indexed by a loop
        // counter lives in scratch memory on the GPU, and a wave alone on its SIMD waits out every one of those round trips.)
        const int S_ = c.stride;
        auto leaf = [&](uint32_t a0, uint32_t a1, int i, uint32_t &py, uint32_t &pc) {
          const int ref = (a0 >> 12) & 7;
          const int dx = (int16_t)(a1 & 0xFFFF), dy = (int16_t)(a1 >> 16), cdx = dx >> 1, cdy = dy >> 1;
          py = (uint32_t)(s.cur_off + (dy >> 1) * S_ + (dx >> 1));
          pc = (uint32_t)(s.cur_off / 2 + (cdy >> 1) * S_ + (cdx >> 1));
          w2 |= (uint32_t)ref << (10 + 3 * i);
          w2 |= (uint32_t)((dx & 1) | ((dy & 1) << 1)) << (16 + 4 * i);
          w2 |= (uint32_t)((cdx & 1) | ((cdy & 1) << 1)) << (18 + 4 * i);
        };
        leaf(s.l0a, s.l0b, 0, w3, w4);
        w5 = w6 = 0;
        if (nl == 2) leaf(s.l1a, s.l1b, 1, w5, w6);
      }
    } else {
      uint32_t *rec_out = s.pay + s.pay_base + s.mb_pay;
      for (int i = 0; i < MOBI_INTRA_RECORDS; i++) rec_out[i] = m.rec(i);
      w4 = w5 = w6 = w7 = MOBI_DEP_NONE | (MOBI_DEP_NONE << 16); // ls_intra_deps fills them in
      s.items[s.n_items++] = (s.clip << 13) | (uint32_t)s.mb;
    }
    MbDesc d;
    d.payload_off = s.pay_base + s.mb_pay;
    d.w1 = (uint32_t)s.mb_type | (nl << 1) | (s.cbp6 << 8) | (s.t8mask << 14) | ((s.quant & 63) << 20) | ((uint32_t)dual << 26);
    d.w2 = w2; d.w3 = w3; d.w4 = w4; d.w5 = w5; d.w6 = w6; d.w7 = w7;
    s.desc[s.mb] = d;
    s.pay_pos = s.mb_pay + s.hdr_words + s.n_coefs;
    if (!s.iframe) { // the row cache: this macroblock's entry, and what the next macroblock will find as its left, top and top right
      m.mvp_store(s.mx + 1, s.mv_cur);
      s.row_e1 = s.mx == 0 ? s.mv_cur : s.row_e1;
      s.row_e2 = s.mx == 1 ? s.mv_cur : s.row_e2;
      const bool row_end = s.mx + 1 == c.mbw;
      const uint32_t top = row_end ? s.row_e1 : s.mv_b, tr = row_end ? (c.mbw >= 2 ? s.row_e2 : 0u) : s.mv_pref;
      s.mv_left = row_end ? 0u : s.mv_cur;
      s.mv_a = top;
      s.mv_b = tr;
    }
    s.mb++;
    if (++s.mx == c.mbw) { s.mx = 0; s.my++; }
    s.st = LS_MB_BEGIN;
  }
  // ---- macroblock start (MD.cs:145-222) ----
  if (s.st == LS_MB_BEGIN) {
    if (s.mb >= c.n_mbs) {
      s.st = LS_DONE;
    } else {
      s.cur_off = s.my * 16 * c.stride + s.mx * 16;
      s.n_coefs = 0; s.cbp6 = s.t8mask = s.w3 = 0;
      s.mb_pay = s.pay_pos; s.hdr_words = 0; s.nleaf = 0;
      if (s.pay_pos + MOBI_MV_CELLS + 384 > c.pay_cap) ls_bail(s, 9);
      else if (s.iframe) {
        ls_refill(s, m);
        s.i_subkind = (int)(ls_win(s) >> 31);
        ls_take(s, 1);
        s.mb_type = MOBI_MB_INTRA;
        s.st = LS_I_HDR;
      } else {
        // left, top, top-right (MD.cs:163-169)
        const uint32_t va = s.mv_left, vb = s.mv_a, vc = s.mv_b;
        s.mv_cur = 0; // (an intra macroblock leaves 0: MD.cs:205-206)
        s.mv_pref = (s.my > 0 && s.mx + 3 <= c.mbw) ? m.mvp_load(s.mx + 3) : 0u; // the next macroblock's top right, as the row above left it
        const int a0 = (int)(int16_t)(va & 0xFFFF), a1 = (int)(int16_t)(va >> 16), b0 = (int)(int16_t)(vb & 0xFFFF), b1 = (int)(int16_t)(vb >> 16),
                  c0 = (int)(int16_t)(vc & 0xFFFF), c1 = (int)(int16_t)(vc >> 16);
        s.predx = ls_max(ls_min(a0, b0), ls_min(ls_max(a0, b0), c0));
        s.predy = ls_max(ls_min(a1, b1), ls_min(ls_max(a1, b1), c1));
        s.mb_type = MOBI_MB_INTER;
        m.stk(0) = 0;
        s.sp = 1;
        s.st = LS_NODE;
      }
    }
  }
  // ---- one node of the partition tree (MD.cs:469-1746) ----
  if (s.st == LS_NODE) {
    ls_refill(s, m);
    const uint32_t it = m.stk(--s.sp);
    const int wi = it & 3, hi = (it >> 2) & 3, x = ((it >> 4) & 15) * 2, y = ((it >> 8) & 15) * 2;
    const int sh = wi * 4 + hi, w = 16 >> wi, h = 16 >> hi;
    const uint32_t pe = T[MOBI_DT_PLUT + sh * 64 + (ls_win(s) >> T[MOBI_DT_PSHIFT + sh])], code = pe & 15; // (ls_prepare_tables: code | length << 4)
    if (pe == 0xFF) ls_bail(s, 10);
    else {
      ls_take(s, (int)(pe >> 4));
      if (code <= 5) {
        int dx = s.predx, dy = s.predy;
        if (code) {
          ls_refill(s, m);
          dx += ls_se(s);
          dy += ls_se(s);
        }
        if (!s.bail) ls_leaf(s, m, c, wi, hi, x, y, code ? (int)code : 1, dx, dy);
      } else if (code <= 7) {
        if (sh != 0) ls_bail(s, 11);
        else { s.mb_type = MOBI_MB_INTRA; s.i_subkind = code == 7; s.st = LS_I_HDR; }
      } else if (code == 8) {
        if (h == 2) ls_bail(s, 11);
        else {
          m.stk(s.sp++) = (uint32_t)(wi | ((hi + 1) << 2) | ((x >> 1) << 4) | (((y + h / 2) >> 1) << 8));
          m.stk(s.sp++) = (uint32_t)(wi | ((hi + 1) << 2) | ((x >> 1) << 4) | ((y >> 1) << 8));
        }
      } else if (code == 9) {
        if (w == 2) ls_bail(s, 11);
        else {
          m.stk(s.sp++) = (uint32_t)((wi + 1) | (hi << 2) | (((x + w / 2) >> 1) << 4) | ((y >> 1) << 8));
          m.stk(s.sp++) = (uint32_t)((wi + 1) | (hi << 2) | ((x >> 1) << 4) | ((y >> 1) << 8));
        }
      }
      if (s.st == LS_NODE && s.sp == 0) s.st = LS_P_CBP;
    }
  }
  // ---- inter macroblock: which areas are coded (loc_1161A0, MD.cs:1818-1833) ----
  if (s.st == LS_P_CBP) {
    ls_refill(s, m);
    s.hdr_words = (s.nleaf == 1 || ls_classify(s)) ? 0 : MOBI_MV_CELLS;
    const uint32_t u = ls_ue(s);
    if (!s.bail) {
      if (u >= 64) ls_bail(s, 12);
      else {
        s.cbp6 = T[MOBI_DT_CBP_P + u];
        s.area_mask = s.cbp6;
        s.sub_mask = 0;
        s.st = s.cbp6 ? LS_NEXT : LS_MB_END;
#ifndef LS_NO_FOLD
        // the first coded area as one 8x8 transform (one bit; at least 17 are left behind a pattern code of at most 15): straight to its tokens
        if (s.cbp6 && s.quant >= 12 && (ls_win(s) >> 31)) {
          const int a = ls_ctz(s.area_mask);
          s.area_mask &= s.area_mask - 1;
          ls_take(s, 1);
          s.t8mask |= 1u << a;
          s.blk_p = 0;
          s.blk_n = 64;
          s.blk_tile = a * 64;
          s.blk_flags = 1u | (s.vlc == 1 ? 2u : 0u) | (s.tables_set ? 4u : 0u);
          s.ret = LS_NEXT;
          s.st = LS_TOKEN;
        }
#ifndef LS_NO_FOLD_PATTERN
        else if (s.cbp6 && s.quant >= 12) { // ... or as 4x4 blocks: the pattern (a code of at most 15 bits whose first is that 0; 17 are left)
          const uint32_t w = ls_win(s);
          const int z = ls_clz(w | 1u);
          const uint32_t u = (z ? ((w << (z + 1)) >> (32 - z)) : 0u) + (1u << z) - 1u;
          const uint32_t pat = (z < 8 && u < 16) ? T[MOBI_DT_CBP4_P + u] : 0u;
          if (pat) {
            const int ar = ls_ctz(s.area_mask), sub = ls_ctz(pat);
            s.area_mask &= s.area_mask - 1;
            ls_take(s, 2 * z + 1);
            s.cur_area = ar;
            s.sub_mask = pat & (pat - 1);
            s.blk_p = 0;
            s.blk_n = 16;
            s.blk_tile = ar * 64 + sub * 16;
            s.blk_flags = (s.vlc == 1 ? 2u : 0u) | (s.tables_set ? 4u : 0u);
            s.ret = LS_NEXT;
            s.st = LS_TOKEN;
          }
        }
#endif
#endif
      }
    }
  }
  ls_next(s, m, c);  // (LS_NEXT_SLOW only: the cheap rounds take LS_NEXT / LS_TOKEN themselves)
  ls_token(s, m, c); // (LS_TOKEN_SLOW only)
  if (s.bail) s.st = LS_DONE;
}
template <class S>
LS_FN void ls_step_intra(LsLane &s, S &m, const LsCtx &c) {
  const uint8_t *T = c.T;
  // ---- intra macroblock header: CBP, then the luma mode of a "full" one (MD.cs:1759-1807) ----
  if (s.st == LS_I_HDR) {
    ls_refill(s, m);
    s.hdr_words = MOBI_INTRA_RECORDS;
    for (int i = 0; i < MOBI_INTRA_RECORDS; i++) m.rec(i) = 0;
    const uint32_t u = ls_ue(s);
    if (!s.bail) {
      if (u >= 64) ls_bail(s, 12);
      else {
        s.i_cbp = T[MOBI_DT_CBP_I + u];
        s.i_k = 0;
        s.i_chroma = 0;
        if (s.i_subkind) s.st = LS_I_SUBAREA;
        else {
          ls_refill(s, m);
          int md = (int)(ls_win(s) >> 29);
          ls_take(s, 3);
          if (md == 2) {
            md = 9;
            ls_refill(s, m);
            const int p = ls_se(s);
            ls_check_intra_reads(s, c, 2, s.cur_off);
            if (p < -32768 || p > 32767) ls_bail(s, 13);
            s.w3 = 1u | ((uint32_t)(uint16_t)(int16_t)p << 16);
          }
          s.i_mode = md;
          s.st = LS_I_FIXED;
        }
      }
    }
  }
  // ---- "sub" intra macroblock, one luma area (DecIntraSubBlockPMode, MD.cs:1789-1807, :2776) ----
  LS_INTRA_LOOP_PRAGMA
  for (int it = 0; it < 5 && s.st == LS_I_SUBAREA; it++) {
    if (s.i_k == 4) s.st = LS_I_CHROMA;
    else {
      const int k = s.i_k, cik = 9 + (k & 1) * 2 + (k >> 1) * 0x10;
      const bool coded = (s.i_cbp >> k) & 1;
      bool whole = true;
      ls_refill(s, m);
      if (coded) {
        if (ls_win(s) >> 31) ls_take(s, 1);
        else whole = false;
      }
      if (whole) {
        const int md = ls_pmode(s, m, cik, false);
        int p = 0;
        if (md == 2) {
          ls_refill(s, m);
          p = ls_se(s);
          if (p < -32768 || p > 32767) ls_bail(s, 13);
        }
        ls_check_intra_reads(s, c, md, ls_area_offset(s, c, k, 0));
        m.rec(k * 4) |= mobi_intra_rec(md, coded, 0, 0, (int16_t)p);
        s.i_k++;
        if (coded && !s.bail) {
          s.cbp6 |= 1u << k;
          s.t8mask |= 1u << k;
          ls_block(s, k, 0, true);
          s.ret = LS_I_SUBAREA;
          if (!s.bail) s.st = LS_TOKEN;
        }
      } else {
        const uint32_t u4 = ls_ue(s);
        if (!s.bail) {
          if (u4 >= 20) ls_bail(s, 12);
          else { s.i_m4 = T[MOBI_DT_CBP4_I + u4]; s.i_sub = 0; s.st = LS_I_SUB4; }
        }
      }
    }
  }
  LS_INTRA_LOOP_PRAGMA
  for (int it = 0; it < 5 && s.st == LS_I_SUB4; it++) {
    if (s.i_sub == 4) { s.i_k++; s.st = LS_I_SUBAREA; }
    else {
      const int k = s.i_k, sub = s.i_sub, cik = 9 + (k & 1) * 2 + (k >> 1) * 0x10;
      ls_refill(s, m);
      const int md = ls_pmode(s, m, cik + (sub & 1) + (sub >> 1) * 8, true);
      int p = 0;
      if (md == 2) {
        ls_refill(s, m);
        p = ls_se(s);
        if (p < -32768 || p > 32767) ls_bail(s, 13);
      }
      ls_check_intra_reads(s, c, md, ls_area_offset(s, c, k, sub));
      const int cd = (s.i_m4 >> sub) & 1;
      m.rec(k * 4 + sub) |= mobi_intra_rec(md, cd, 1, 0, (int16_t)p);
      s.i_sub++;
      if (cd && !s.bail) {
        s.cbp6 |= 1u << k;
        ls_block(s, k, sub, false);
        s.ret = LS_I_SUB4;
        if (!s.bail) s.st = LS_TOKEN;
      }
    }
  }
  // ---- chroma mode of an intra macroblock (loc_116290, MD.cs:1864-1880) ----
  auto chroma_mode = [&]() {
    ls_refill(s, m);
    int md = (int)(ls_win(s) >> 29);
    ls_take(s, 3);
    if (md == 2) {
      md = 9;
      for (int area = 4; area < 6; area++) {
        ls_refill(s, m);
        const int p = ls_se(s);
        ls_check_intra_reads(s, c, 2, ls_area_offset(s, c, area, 0));
        if (p < -32768 || p > 32767) ls_bail(s, 13);
        m.rec(area * 4) |= mobi_intra_rec(0, 0, 0, 1, (int16_t)p);
      }
    }
    s.i_mode = md;
    s.i_k = 4;
    s.i_chroma = 1;
  };
  if (s.st == LS_I_CHROMA) {
    chroma_mode();
    if (!s.bail) s.st = LS_I_FIXED;
  }
  // ---- one area whose mode is known (sub_116508, MD.cs:2869-2896) ----
  // (area after area in one visit while they are not coded: a visit costs the wave a round of the whole walk)
  LS_INTRA_LOOP_PRAGMA
  for (int it = 0; it < 8 && s.st == LS_I_FIXED; it++) {
    if (s.i_k == 4 && !s.i_chroma) chroma_mode();
    else if (s.i_k == 6) s.st = LS_MB_END;
    else {
      const int k = s.i_k, md = s.i_mode;
      const bool coded = (s.i_cbp >> k) & 1;
      s.i_k++;
      if (!coded) {
        ls_check_intra_reads(s, c, md, ls_area_offset(s, c, k, 0));
        m.rec(k * 4) |= mobi_intra_rec(md, 0, 0, 0, 0);
      } else {
        ls_refill(s, m);
        if (ls_win(s) >> 31) {
          ls_take(s, 1);
          ls_check_intra_reads(s, c, md, ls_area_offset(s, c, k, 0));
          m.rec(k * 4) |= mobi_intra_rec(md, 1, 0, 0, 0);
          s.cbp6 |= 1u << k;
          s.t8mask |= 1u << k;
          ls_block(s, k, 0, true);
          s.ret = LS_I_FIXED;
          if (!s.bail) s.st = LS_TOKEN;
        } else {
          const uint32_t u = ls_ue(s);
          if (!s.bail) {
            if (u >= 20) ls_bail(s, 12);
            else {
              const uint32_t m4 = T[MOBI_DT_CBP4_I + u];
              for (int sub = 0; sub < 4; sub++) {
                ls_check_intra_reads(s, c, md, ls_area_offset(s, c, k, sub));
                m.rec(k * 4 + sub) |= mobi_intra_rec(md, (m4 >> sub) & 1, 1, 0, 0);
              }
              if (m4) s.cbp6 |= 1u << k;
              s.sub_mask = m4;
              s.cur_area = k;
              if (!s.bail) s.st = LS_I_FSUB;
            }
          }
        }
      }
    }
  }
  if (s.st == LS_I_FSUB) {
    if (s.sub_mask) {
      const int sub = ls_ctz(s.sub_mask);
      s.sub_mask &= s.sub_mask - 1;
      ls_block(s, s.cur_area, sub, false);
      s.ret = LS_I_FSUB;
      if (!s.bail) s.st = LS_TOKEN;
    } else s.st = LS_I_FIXED;
  }
  if (s.bail) s.st = LS_DONE; // (a region that bailed out half way may have gone on to set a state)
}
LS_FN bool ls_in_intra(const LsLane &s) { return s.st >= LS_I_HDR && s.st <= LS_I_FSUB; }
// One round as the kernel runs it: the whole walk once, then the cheap rounds (per lane here; on the GPU each part runs while any lane of the
// wave is in it)
// with_intra: the kernel leaves the intra part out of some rounds of a wave that holds P-frames only (mobi_lsparse.hip, LS_INTRA_PERIOD): its
// lanes wait there for the next round that has it -- nothing else changes for them
template <class S>
LS_FN void ls_round(LsLane &s, S &m, const LsCtx &c, bool with_intra = true) {
  ls_step_main(s, m, c);
  if (with_intra) ls_step_intra(s, m, c);
  ls_next_fast(s, m, c);
  ls_token_fast(s, m, c);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LS_CHEAP_LOOP)
#pragma unroll // (unrolled: as a loop it began and ended with 32 register copies, the lane state carried round it)
#elif defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int k = 0; k < LS_TOKEN_ROUNDS; k++) {
    // The exit is the WAVE's (no lane has a block or a token to read), not the lane's: a lane leaving a loop on its own makes every piece of
    // its state a value that has to be kept apart from the others' -- a third of the loop's vector instructions were register copies.
#if !defined(LS_CHEAP_NT) // (r05: "next block" is folded into the token part wherever it reads at most one bit; what is left of it -- a 4x4
    if (!LS_ANY(s.st == LS_TOKEN)) break; //  pattern, 420 of a frame's 2500 -- waits for the round's one visit: 24.97 -> 24.45 ms)
#else
    if (!LS_ANY(s.st == LS_TOKEN || s.st == LS_NEXT)) break;
    ls_next_fast(s, m, c);
#endif
    ls_token_fast(s, m, c);
  }
}

// What the reference's reader would report as consumed after c bits of a stream of len bytes (see the header); -1: it would have thrown
LS_FN int ls_consumed(uint32_t cbits, uint32_t len) {
  const uint32_t f = (cbits + 15) / 16, fmax = len / 2 - 1; // len >= 2
  if (f <= fmax) return (int)(2 + 2 * f);
  return (len & 1) ? -1 : (int)len;
}

// ---------------------------------------------------------------- dependency lists of the intra macroblocks (one lane per macroblock)
struct LsGeom { int width, height, stride, lg, mbw; };
LS_FN int ls_owner_luma(const LsGeom &g, int a) {
  if (a < 0) return -1;
  const int row = a >> g.lg, col = a & (g.stride - 1);
  if (col >= g.width || row >= g.height) return -1;
  return (row >> 4) * g.mbw + (col >> 4);
}
LS_FN int ls_owner_chroma(const LsGeom &g, int a) {
  if (a < 0) return -1;
  const int row = a >> g.lg, col = a & (g.stride - 1);
  const int x = col >= g.stride / 2 ? col - g.stride / 2 : col;
  if (x >= g.width / 2 || row >= g.height / 2) return -1;
  return (row >> 3) * g.mbw + (x >> 3);
}
#if defined(__HIP_DEVICE_COMPILE__)
LS_FN void ls_or_u32(uint32_t *p, uint32_t v) { atomicOr(p, v); }
#else
LS_FN void ls_or_u32(uint32_t *p, uint32_t v) { *p |= v; }
#endif
// the raster-earlier macroblocks the prediction halo of macroblock mb touches, as mobi_dparse.hip's end_mb lists them.  desc = the clip's row.
// Returns false when there are more than MOBI_INTRA_DEPS (mobi_parse_frames refuses such a stream).
LS_FN bool ls_intra_deps(const LsGeom &g, MbDesc *desc, int mb) {
  uint32_t deps[MOBI_INTRA_DEPS];
  int n = 0;
  bool ok = true;
  const int S = g.stride, o = ((mb / g.mbw) * 16) * S + (mb % g.mbw) * 16;
  int probes[21];
  probes[0] = ls_owner_luma(g, o - S - 1); probes[1] = ls_owner_luma(g, o - S); probes[2] = ls_owner_luma(g, o - S + 16);
  probes[3] = ls_owner_luma(g, o - 1); probes[4] = ls_owner_luma(g, o + 16); probes[5] = ls_owner_luma(g, o + S - 1); probes[6] = ls_owner_luma(g, o + S + 16);
  for (int v = 0; v < 2; v++) {
    const int b = o / 2 + v * (S / 2);
    int *p = probes + 7 + 7 * v;
    p[0] = ls_owner_chroma(g, b - S - 1); p[1] = ls_owner_chroma(g, b - S); p[2] = ls_owner_chroma(g, b - S + 8);
    p[3] = ls_owner_chroma(g, b - 1); p[4] = ls_owner_chroma(g, b + 8); p[5] = ls_owner_chroma(g, b + S - 1); p[6] = ls_owner_chroma(g, b + S + 8);
  }
  for (int i = 0; i < 21; i++) {
    const int ow = probes[i];
    if (ow < 0 || ow >= mb) continue;
    bool seen = false;
    for (int k = 0; k < n; k++) seen = seen || (int)(deps[k] & 0x1FFF) == ow;
    if (seen) continue;
    if (n == MOBI_INTRA_DEPS) { ok = false; break; }
    const bool intra = (desc[ow].w1 & 1) == MOBI_MB_INTRA;
    deps[n++] = (uint32_t)ow | (intra ? 0u : MOBI_DEP_INTER);
    if (intra) { // w3 [1] has intra dependencies, [2] has intra dependents (mobi_recon_intra_cl); other lanes mark other words' bits at the same time
      ls_or_u32(&desc[ow].w3, 4u);
      ls_or_u32(&desc[mb].w3, 2u);
    }
  }
  for (int k = n; k < MOBI_INTRA_DEPS; k++) deps[k] = MOBI_DEP_NONE;
  desc[mb].w4 = deps[0] | (deps[1] << 16);
  desc[mb].w5 = deps[2] | (deps[3] << 16);
  desc[mb].w6 = deps[4] | (deps[5] << 16);
  desc[mb].w7 = deps[6] | (deps[7] << 16);
  return ok;
}
#endif
