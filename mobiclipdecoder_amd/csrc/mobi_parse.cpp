// mobi_parse.cpp -- serial bitstream parser -> per-macroblock command list (see mobi_parse.h).
#include "mobi_parse.h"
#include "mobi_recon_math.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "../../include/mobiclip_hip.h"
#include "mobi_dparse_tables.h"
#include "mobi_tables.h"

std::atomic<unsigned long> mobi_refusal_count[MOBI_REFUSE_CLASSES]; // (parse-pool threads count concurrently: relaxed adds)
std::atomic<unsigned long> mobi_literal_frame_count;
std::atomic<unsigned long> mobi_scratch_read_count; // walks that read the transforms' scratch, Internal[154..217] (r01-r04 refused them): a measuring aid

namespace {
inline uint32_t shl(uint32_t x, int n) { return x << (n & 31); } // C# masks shift counts to 5 bits
inline uint32_t shr(uint32_t x, int n) { return x >> (n & 31); }
inline int clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; } // MD.cs:3927
} // namespace

void ParsedFrame::clear() {
  memset(&hdr, 0, sizeof(hdr));
  desc.clear();
  payload.clear();
  intra_mbs.clear();
  intra_items.clear();
  class_start.clear();
  level_start.clear();
}

MobiStreamParser::MobiStreamParser(uint32_t width, uint32_t height, int version) : version_(version) {
  g_.width = (int)width;
  g_.height = (int)height;
  g_.stride = (width <= 256) ? 256 : (width <= 512) ? 512 : 1024; // MD.cs:50-52
  g_.mbw = (int)width / 16;
  g_.mbh = (int)height / 16;
  g_.lg = g_.stride == 256 ? 8 : g_.stride == 512 ? 9 : 10;
  ver_ = (version == MOBI_VERSION_MOFLEX3DS) ? 0 : 1;
  memset(dq8_, 0, sizeof(dq8_));
  memset(dq4_, 0, sizeof(dq4_));
  memset(mcache_, 0, sizeof(mcache_));
  memset(recs_, 0, sizeof(recs_));
  mvc_.assign(2 * (g_.mbw + 2), 0);
}

// ------------------------------------------------------------------ bit reader (MD.cs:2970-3015)
uint32_t MobiStreamParser::data_u16(long off) const { // IOUtil.ReadU16LE
  if (off < 0 || off + 1 >= len_) fail(MOBI_E_INDEX);
  return ((uint32_t)data_[off + 1] << 8) | data_[off];
}
void MobiStreamParser::fill_bits() { // FillBits: one 16-bit LE word, no refill at/after Data.Length
  if (off_ >= len_) return;
  uint32_t w = data_u16(off_);
  off_ += 2;
  nbr_ += 16;
  win_ |= shl(w, 16 - nbr_);
}
void MobiStreamParser::take(int n) {
  win_ = shl(win_, n);
  nbr_ -= n;
  if (nbr_ < 0) fill_bits();
}
uint32_t MobiStreamParser::ue() { // Elias-gamma, value = 2^z - 1 + suffix
  int z = clz32(win_);
  win_ = shl(win_, z);
  win_ += win_;
  uint32_t v = (z == 0) ? 0 : shr(win_, 32 - z);
  v += shl(1u, z);
  v--;
  win_ = shl(win_, z);
  nbr_ -= 2 * z;
  if (--nbr_ < 0) fill_bits();
  return v;
}
int MobiStreamParser::se() { // odd codes map to non-positive values (MD.cs:3009-3010)
  int z = clz32(win_);
  win_ = shl(win_, z);
  win_ += win_;
  uint32_t u = (z == 0) ? 0 : shr(win_, 32 - z);
  u += shl(1u, z);
  int v = (int)u;
  if (v & 1) v = (int)(1u - u);
  v >>= 1;
  win_ = shl(win_, z);
  nbr_ -= 2 * z;
  if (--nbr_ < 0) fill_bits();
  return v;
}

// ------------------------------------------------------------------ quantiser (MD.cs:3884-3925)
void MobiStreamParser::setup_quant(uint32_t q) {
  if (version_ == MOBI_VERSION_MOFLEX3DS) q = std::min<uint32_t>(std::max<uint32_t>(q, 12), 52);
  quant_ = q; // assigned before the table index can throw: the old tables then serve the new Quantizer (tq_)
  if (q >= sizeof(mobi_qdiv6)) fail(MOBI_E_INDEX);
  tq_ = q;
  build_dq();
  static const int border[8] = {1, 2, 3, 4, 8, 0x10, 0x18, 0x20}; // "no neighbour" marks, re-armed only here
  for (int b : border) mcache_[b] = 9;
}
void MobiStreamParser::build_dq() { // Internal[10..89] for quantiser tq_ (MD.cs:3892-3911)
  if (tq_ >= sizeof(mobi_qdiv6)) { memset(dq8_, 0, sizeof(dq8_)); memset(dq4_, 0, sizeof(dq4_)); return; }
  const int sh = mobi_qdiv6[tq_] + 8, m = mobi_qmod6[tq_];
  for (int i = 0; i < 16; i++) dq4_[i] = (uint32_t)mobi_zz4[i] | shl(mobi_dq4[m * 16 + i], sh);
  for (int i = 0; i < 64; i++) dq8_[i] = (uint32_t)mobi_zz8[i] | shl(mobi_dq8[m * 64 + i], sh - 2);
}

// ------------------------------------------------------------------ the state that survives a frame (mobi_state.h)
void MobiStreamParser::import_state(const MobiDevState &st, const MobiDevTail &tail) {
  quant_ = st.quant;
  yuvfmt_ = st.yuvfmt;
  frames_started_ = st.frames_started;
  tq_ = st.tables_set ? st.quant : (uint32_t)MOBI_TQ_NONE; // (a frame after which the two differ never finishes on the device)
  build_dq();
  memcpy(mcache_, st.mcache, sizeof(mcache_));
  predx_ = st.predx;
  predy_ = st.predy;
  memcpy(ib_, tail.ib, sizeof(ib_));
  memcpy(scr_, tail.scratch, sizeof(scr_));
  pend64_ = pend16_ = false;
  for (size_t i = 0; i < mvc_.size(); i++) mvc_[i] = tail.mvc[i];
  memset(itail_, 0, sizeof(itail_)); // (only a walk writes there, and a frame with a walk is not the device's)
  i218_ = 0;
  vlc_table_ = 0;
}
void MobiStreamParser::export_state(MobiDevState &st, MobiDevTail &tail) {
  memset(&st, 0, sizeof(st));
  memset(&tail, 0, sizeof(tail));
  st.quant = quant_;
  st.yuvfmt = yuvfmt_;
  st.frames_started = frames_started_;
  st.tables_set = tq_ != MOBI_TQ_NONE;
  memcpy(st.mcache, mcache_, sizeof(mcache_));
  st.predx = predx_;
  st.predy = predy_;
  scratch_materialise();
  memcpy(tail.ib, ib_, sizeof(ib_));
  memcpy(tail.scratch, scr_, sizeof(scr_));
  for (size_t i = 0; i < mvc_.size(); i++) tail.mvc[i] = mvc_[i];
}
uint32_t MobiStreamParser::internal_word(uint32_t idx) { return internal_read(idx); }
bool MobiStreamParser::device_ready() const {
  if (!last_frame_ok_ || frame_literal_ || frame_host_only_) return false;
  if (tq_ != quant_ && (tq_ != MOBI_TQ_NONE || quant_ != 0)) return false; // tables of another quantiser than Quantizer, or none at all under a Quantizer that is
                                                                           // not 0 (a first SetupQuantizationTables that threw, ModsDS q >= 54): the device's state has
                                                                           // one field for both and would name scale row `quant` where this parser names tq_
  for (uint32_t v : itail_)
    if (v) return false;
  return true;
}

// ------------------------------------------------------------------ per-MB assembly
void MobiStreamParser::begin_mb(int mb, int type) {
  cur_mb_ = mb;
  cur_x_ = (mb % g_.mbw) * 16;
  cur_y_ = (mb / g_.mbw) * 16;
  cur_off_ = (long)cur_y_ * g_.stride + cur_x_;
  n_leaf_words_ = 0;
  n_coefs_ = 0;
  memset(recs_, 0, sizeof(recs_));
  cbp6_ = t8mask_ = w3_ = 0;
  any_wide_ = false;
  mb_type_ = type;
}
void MobiStreamParser::end_mb() {
  MbDesc d{};
  d.payload_off = (uint32_t)out_->payload.size();
  uint32_t nl = 0;
  int dual = MOBI_DUAL_NONE;
  d.w2 = (uint32_t)n_coefs_;
  d.w3 = w3_;
  if (mb_type_ == MOBI_MB_INTER) {
    nl = (uint32_t)(n_leaf_words_ / 2);
    if (nl == 2) { // two halves (leaf word 0: x/2 | y/2<<4 | wi<<8 | hi<<10 | ref<<12)
      const uint32_t a = leaves_[0] & 0xFFF, b = leaves_[2] & 0xFFF;
      if (a == (0u | (1u << 10)) && b == ((4u << 4) | (1u << 10))) dual = MOBI_DUAL_TB;
      if (a == (0u | (1u << 8)) && b == (4u | (1u << 8))) dual = MOBI_DUAL_LR;
    }
    if (nl == 1 || dual) { // leaf records: positions and phases instead of motion vectors (MD.cs:400-416)
      const long S = g_.stride;
      uint32_t pos[4] = {0, 0, 0, 0};
      for (uint32_t i = 0; i < nl; i++) {
        const int ref = (leaves_[2 * i] >> 12) & 7;
        const int dx = (int16_t)(leaves_[2 * i + 1] & 0xFFFF), dy = (int16_t)(leaves_[2 * i + 1] >> 16), cdx = dx >> 1, cdy = dy >> 1;
        pos[2 * i] = (uint32_t)(int32_t)(cur_off_ + (long)(dy >> 1) * S + (dx >> 1));
        pos[2 * i + 1] = (uint32_t)(int32_t)(cur_off_ / 2 + (long)(cdy >> 1) * S + (cdx >> 1));
        d.w2 |= (uint32_t)ref << (10 + 3 * i);
        d.w2 |= (uint32_t)((dx & 1) | ((dy & 1) << 1)) << (16 + 4 * i);
        d.w2 |= (uint32_t)((cdx & 1) | ((cdy & 1) << 1)) << (18 + 4 * i);
      }
      d.w3 = pos[0]; d.w4 = pos[1]; d.w5 = pos[2]; d.w6 = pos[3];
    } else {
      build_cells();
      out_->payload.insert(out_->payload.end(), cells_, cells_ + MOBI_MV_CELLS);
    }
  } else {
    out_->payload.insert(out_->payload.end(), recs_, recs_ + MOBI_INTRA_RECORDS);
  }
  out_->payload.insert(out_->payload.end(), coefs_, coefs_ + n_coefs_);
  if (mb_type_ == MOBI_MB_INTRA && any_wide_) out_->payload.insert(out_->payload.end(), (const uint32_t *)wide_, (const uint32_t *)wide_ + MOBI_WIDE_PARAMS);
  d.w1 = (uint32_t)mb_type_ | (nl << 1) | (cbp6_ << 8) | (t8mask_ << 14) | ((tq_ & 63) << 20) | ((uint32_t)dual << 26);
  out_->desc.push_back(d);
}
long MobiStreamParser::area_offset(int area, int sub) const {
  const long S = g_.stride;
  long o = (area < 4) ? cur_off_ + (area >> 1) * 8 * S + (area & 1) * 8 : cur_off_ / 2 + (area == 5 ? S / 2 : 0);
  return o + (sub >> 1) * 4 * S + (sub & 1) * 4;
}

// ------------------------------------------------------------------ motion (MD.cs:400-456)
// Would CopyBlock throw?  Rows are visited top to bottom, so first row / last row bound the rest.
void MobiStreamParser::check_window(long pos, int w, int h, int phase, long plane_len) const {
  if (pos < 0) fail(MOBI_E_INDEX);
  long last = pos + (long)(h - 1) * g_.stride;
  long hi; // highest index touched (phase 0: Array.Copy end is exclusive)
  switch (phase) {
    case 0: hi = last + w - 1; break;
    case 1: hi = last + w; break;
    case 2: hi = last + w - 1 + g_.stride; break;
    default: hi = last + w + g_.stride; break;
  }
  if (hi >= plane_len) fail(MOBI_E_INDEX);
}
void MobiStreamParser::mc_leaf(int wi, int hi, int x, int y, int ref, int dx, int dy, int mv_slot) {
  const long S = g_.stride;
  const int w = 16 >> wi, h = 16 >> hi;
  mvc_[mv_slot] = dx; // every leaf overwrites the MB's exported MV (MD.cs:411-412)
  mvc_[mv_slot + 1] = dy;
  if (ref > std::min(5, frames_started_ - 1)) fail(MOBI_E_NULLREF); // Y[ref] == null
  const long off = cur_off_ + (long)y * S + x;
  const long cdx = (long)dx >> 1, cdy = (long)dy >> 1;
  const long cpos = off / 2 + (cdy >> 1) * S + (cdx >> 1);
  const int cph = (int)((cdx & 1) | ((cdy & 1) << 1));
  if (dx >= -MOBI_MV_LIMIT && dx <= MOBI_MV_LIMIT && dy >= -MOBI_MV_LIMIT && dy <= MOBI_MV_LIMIT) {
    // the three windows at once: the luma one, and of the two chroma ones U starts first and V (S/2 further) ends last
    const long pos = off + (long)(dy >> 1) * S + (dx >> 1), ylen = S * g_.height;
    const long hi_y = pos + (long)(h - 1) * S + w - 1 + (dx & 1) + ((dy & 1) ? S : 0); // phase 0: Array.Copy end is exclusive
    const long hi_c = cpos + S / 2 + (long)((h >> 1) - 1) * S + (w >> 1) - 1 + (cdx & 1) + ((cdy & 1) ? S : 0);
    if (pos < 0 || hi_y >= ylen || cpos < 0 || hi_c >= ylen / 2) fail(MOBI_E_INDEX);
  } else {
    check_window(off + (long)(dy >> 1) * S + (dx >> 1), w, h, (dx & 1) | ((dy & 1) << 1), S * g_.height);
    check_window(cpos, w >> 1, h >> 1, cph, S * g_.height / 2);
    check_window(cpos + S / 2, w >> 1, h >> 1, cph, S * g_.height / 2);
    // A vector beyond the command list's fields whose windows lie inside the planes (r01-r04: refused).  The reference addresses LINEARLY
    // (MD.cs:400-416): luma source = off + (dy >> 1) * S + (dx >> 1), chroma = off / 2 + (dy >> 2) * S + (dx >> 2), phases = the low bits --
    // so (dx - 4 t S, dy + 4 t) is the same copy for every t (t rows down and t * S samples back, in luma: 2 t rows and 2 t S samples;
    // no low bit of dx, dx >> 1, dy, dy >> 1 moves).  The t that brings |dx| below 2 S leaves |dy| within the plane's height (the window
    // was just checked), i.e. both inside the cell map's 14 bits.  The MV row cache above keeps the vector as it was read.
    const long t = ((long)dx + (dx >= 0 ? 2 * S : -2 * S)) / (4 * S);
    const long ndx = (long)dx - 4 * t * S, ndy = (long)dy + 4 * t;
    if (ndx < -MOBI_MV_LIMIT || ndx > MOBI_MV_LIMIT || ndy < -MOBI_MV_LIMIT || ndy > MOBI_MV_LIMIT) fail(MOBI_E_INDEX); // (cannot happen: see above)
    dx = (int)ndx;
    dy = (int)ndy;
    frame_host_only_ = true;
  }
  leaves_[n_leaf_words_++] = mobi_leaf_w0(x, y, wi, hi, ref); // at most 64 leaves: the tree bottoms out at 2x2
  leaves_[n_leaf_words_++] = mobi_leaf_w1(dx, dy);
}
// the 64-entry MV cell map of a macroblock with a deeper partition tree, from its leaves (only those need it)
void MobiStreamParser::build_cells() {
  for (int i = 0; i + 1 < n_leaf_words_; i += 2) {
    const uint32_t w0 = leaves_[i], mv = leaves_[i + 1];
    const int x = (int)(w0 & 15) * 2, y = (int)((w0 >> 4) & 15) * 2, w = 16 >> ((w0 >> 8) & 3), h = 16 >> ((w0 >> 10) & 3);
    const uint32_t cell = mobi_cell((int16_t)(mv & 0xFFFF), (int16_t)(mv >> 16), (int)((w0 >> 12) & 7));
    for (int cy = y >> 1; cy < (y + h) >> 1; cy++)
      for (int cx = x >> 1; cx < (x + w) >> 1; cx++) cells_[cy * 8 + cx] = cell;
  }
}
// ReadPBlock*/SwitchPBlock* (MD.cs:469-1746) as one table-driven routine; x,y are MB-relative.
void MobiStreamParser::pblock(int wi, int hi, int x, int y, int mv_slot) {
  const int s = wi * 4 + hi, w = 16 >> wi, h = 16 >> hi;
  uint32_t code = mobi_part_lut[ver_][s][win_ >> mobi_part_shift[ver_][s]];
  if (code >= mobi_part_nbits_len[ver_][s]) fail(MOBI_E_INDEX);
  take(mobi_part_bits[ver_][s][code]);
  if (code == 0) {
    mc_leaf(wi, hi, x, y, 1, predx_, predy_, mv_slot);
  } else if (code <= 5) {
    int dx = se();
    int dy = se();
    mc_leaf(wi, hi, x, y, (int)code, dx + predx_, dy + predy_, mv_slot);
  } else if (code == 6 || code == 7) {
    if (s != 0) fail(MOBI_E_PARTCODE);
    mb_type_ = MOBI_MB_INTRA;
    if (code == 6) intra_full(); else intra_sub();
    return;
  } else if (code == 8) {
    if (h == 2) fail(MOBI_E_PARTCODE);
    pblock(wi, hi + 1, x, y, mv_slot);
    pblock(wi, hi + 1, x, y + h / 2, mv_slot);
  } else if (code == 9) {
    if (w == 2) fail(MOBI_E_PARTCODE);
    pblock(wi + 1, hi, x, y, mv_slot);
    pblock(wi + 1, hi, x + w / 2, y, mv_slot);
  }
  if (s == 0) p_residual();
}

// ------------------------------------------------------------------ residual (MD.cs:3330-3432)
// The reference's coefficient store is  r8 = Internal[r12++]; Internal[90 + (r8 & 0xFF)] = (r8 >> 8) * value  (MD.cs:3424-3429): r12 walks
// the dequant words (Internal[10..73] / [74..89]) and a run that is too long simply walks on -- into the 4x4 words, the coefficient block
// itself ([90..153], with whatever earlier blocks and transforms left there), the transforms' scratch, the table select [218], the MV
// predictors and row cache -- while below quantiser 12 (ModsDS) the words' own low byte carries table bits and the "zigzag index" reaches
// 255.  r03 refused both.  r04 keeps the words of Internal[] such a walk can touch (all but the scratch) and walks with it: the block's
// coefficients are then whatever Internal[90..] holds when the transform starts, shipped as LITERAL values (see literal_frame).
// The first passes the last transforms left in Internal[154..217] (mobi_state.h: what each variant writes), made when somebody looks
void MobiStreamParser::scratch_materialise() {
  if (pend64_) mobi_scratch_from64(sc64_, scr_);
  if (pend16_) {
    uint32_t c[64];
    for (int k = 0; k < 4; k++)
      for (int m = 0; m < 4; m++) c[8 * k + m] = sc16_[4 * k + m];
    mobi_scratch_from16(c, scr_);
  }
  pend64_ = pend16_ = false;
}
uint32_t MobiStreamParser::internal_read(uint32_t idx) {
  if (idx >= 392) fail(MOBI_E_INDEX); // managed array bounds
  if (idx < 10) fail(MOBI_E_INDEX);   // (never: r12 starts at 10 and only grows)
  if (idx < 74) return dq8_[idx - 10];
  if (idx < 90) return dq4_[idx - 74];
  if (idx < 154) return ib_[idx - 90];
  if (idx < 218) { mobi_scratch_read_count.fetch_add(1, std::memory_order_relaxed); scratch_materialise(); return scr_[idx - 154]; }
  if (idx == 218) return i218_;
  if (idx == 219) return (uint32_t)predx_;
  if (idx == 220) return (uint32_t)predy_;
  if (idx - 221 < mvc_.size()) return (uint32_t)mvc_[idx - 221];
  return itail_[idx];
}
void MobiStreamParser::internal_write(uint32_t idx, uint32_t v) { // idx = 90 + a byte: 90..345
  if (idx < 154) ib_[idx - 90] = v;
  else if (idx < 218) { scratch_materialise(); scr_[idx - 154] = v; }
  else if (idx == 218) { i218_ = v; vlc_table_ = v == 1; } // (the tables of the block being read were chosen at its start, MD.cs:3332-3333)
  else if (idx == 219) predx_ = (int)v; // (dead: set again before the next macroblock's first leaf, MD.cs:207-208)
  else if (idx == 220) predy_ = (int)v;
  else if (idx - 221 < mvc_.size()) mvc_[idx - 221] = (int)v; // the MV row cache: later macroblocks' predictors see it
  else itail_[idx] = v;
}
void MobiStreamParser::resid_block(int area, int sub, bool is8) {
  const int N = is8 ? 64 : 16;
  const uint32_t *dq = is8 ? dq8_ : dq4_;
  const uint32_t start = is8 ? 10 : 74;
  const uint16_t *A = vlc_table_ == 1 ? mobi_vx2table1_a : mobi_vx2table0_a;
  const uint8_t *B = vlc_table_ == 1 ? mobi_vx2table1_b : mobi_vx2table0_b;
  memset(ib_, 0, sizeof(uint32_t) * N); // MD.cs:2933-2936 / 2948-2951, 2960-2963
  bool big = false;                     // some coefficient of the block is beyond int16 (only a frame of literal values minds: literal_frame)
  bool odd = tq_ < 12;                  // tables built below 12: every 8x8 dequant word may point anywhere (MD.cs:3907-3911 vs :3426)
  uint32_t r12 = start;
  const int n0 = n_coefs_;
  const int tile = is8 ? area * 64 : area * 64 + sub * 16;
  for (;;) {
    int skip, value;
    uint32_t e;
    if ((win_ >> 25) == 3) { // escape prefix 0000011
      win_ <<= 7;
      bool c = (win_ >> 31) == 1;
      win_ <<= 1;
      if (!c) { // "0": table code, level += B[last<<6|run]
        nbr_ -= 8;
        if (nbr_ < 0) fill_bits();
        e = A[win_ >> 20];
        value = (int)((e >> 4) & 0x1F) + B[e >> 9];
        win_ = shl(win_, (int)(e & 0xF) - 1);
        if (win_ >> 31) value = -value;
        win_ <<= 1;
        nbr_ -= (int)(e & 0xF);
        if (nbr_ < 0) fill_bits();
        skip = (int)((e >> 9) & 0x3F);
        e >>= 15;
      } else {
        c = (win_ >> 31) == 1;
        win_ <<= 1;
        nbr_ -= 9;
        if (nbr_ < 0) fill_bits();
        if (!c) { // "10": table code, run += B[0x80 + level + (last<<6)]
          e = A[win_ >> 20];
          value = (int)((e >> 4) & 0x1F);
          skip = (int)((e >> 9) & 0x3F) + B[0x80 + value + ((e >> 15) << 6)];
          win_ = shl(win_, (int)(e & 0xF) - 1);
          if (win_ >> 31) value = -value;
          win_ <<= 1;
          nbr_ -= (int)(e & 0xF);
          if (nbr_ < 0) fill_bits();
          e >>= 15;
        } else { // "11": raw last(1) run(6) level(s12)
          e = win_ >> 31;
          win_ <<= 1;
          skip = (int)(win_ >> 26);
          win_ <<= 6;
          nbr_ -= 7;
          if (nbr_ < 0) fill_bits();
          value = (int32_t)win_ >> 20;
          win_ <<= 12;
          nbr_ -= 12;
          if (nbr_ < 0) fill_bits();
        }
      }
    } else {
      e = A[win_ >> 20];
      value = (int)((e >> 4) & 0x1F);
      win_ = shl(win_, (int)(e & 0xF) - 1);
      if (win_ >> 31) value = -value;
      win_ <<= 1;
      nbr_ -= (int)(e & 0xF);
      if (nbr_ < 0) fill_bits();
      skip = (int)((e >> 9) & 0x3F);
      e >>= 15;
    }
    r12 += (uint32_t)skip;
    if (!odd && r12 < start + (uint32_t)N) { // the word is the block's own and its low byte a position inside the block
      const uint32_t word = dq[r12 - start], cv = (word >> 8) * (uint32_t)value; // (int * int in the reference: the low 32 bits either way)
      ib_[word & 0xFF] = cv;
      big |= cv + 32768u > 65535u;
      // The reference picks a reduced IDCT from the final scan index (MD.cs:2939-2942, 2954-2955); the reduced transforms only
      // look at part of the block, but for q >= 12 nothing they skip can be nonzero: scan positions 0, 0..2, 0..9 map inside the
      // respective regions (tests/test_oracle_identities.py pins that property of the zigzag tables), so every level is kept.
      if (value != 0) coefs_[n_coefs_++] = (uint32_t)(tile + (int)(word & 0xFF)) | ((uint32_t)(int)(int16_t)value << 16); // = mobi_coef()
      else frame_host_only_ = true; // (a token without a level: the device parsers stop there, mobi_state.h)
    } else { // MD.cs:3424-3429 as written
      odd = true;
      const uint32_t r8 = internal_read(r12);
      internal_write(90 + (r8 & 0xFF), (r8 >> 8) * (uint32_t)value);
    }
    r12++;
    if (e & 1) break;
  }
  // the transform variant the reference runs (by the final index, MD.cs:2939-2942, 2954-2955, 2966-2967) and what it reads
  enum { V1, V3, V16, VALL };
  const int variant = is8 ? (r12 <= 11 ? V1 : r12 <= 13 ? V3 : r12 <= 20 ? V16 : VALL) : (r12 <= 75 ? V1 : VALL);
  if (odd) {
    // The level words written so far describe the block only if every store went where it should: replace them by the block as the
    // transform will see it, value by value (positions the variant does not read are not part of it).
    frame_literal_ = true;
    n_coefs_ = n0;
    for (int p = 0; p < N; p++) {
      const bool read = variant == VALL || p == 0 || (variant == V3 && (p == 1 || p == 8)) || (variant == V16 && (p & 7) < 4 && p < 32);
      const int32_t v = (int32_t)ib_[p];
      if (!read || v == 0) continue;
      if (v != (int16_t)v) { // a literal travels in the level's 16 bits
        if (!surely_faults(is8, variant)) refuse(MOBI_REFUSE_RUN);
        frame_fault_ = true; // (the frame is rejected at its end, as a fault the kernels find is; the word itself no longer matters)
        continue;
      }
      coefs_[n_coefs_++] = (uint32_t)(tile + p) | 0x8000u | ((uint32_t)v << 16); // bit 15: a value, not a level (literal_frame clears it)
    }
  }
  if (big && !odd && !surely_faults(is8, variant)) big_unsure_ = true;
  // what the transforms themselves leave in Internal[90..217] (a later run past a block may read it; mobi_state.h)
  if (is8 && variant == V3) mobi_ib_after3(ib_);         // IDCT3Px8 keeps its first pass in Internal[90..97] (MD.cs:3661-3707)
  else if (!is8 && variant == VALL) mobi_ib_after16x4(ib_); // IDCT16Px4's first pass goes to Internal[106..121] (MD.cs:3728-3784)
  else if (is8 && variant == VALL) { // IDCT64Px8's first pass fills Internal[154..217] (MD.cs:3452-3500): kept as its coefficients until read
    memcpy(sc64_, ib_, sizeof(sc64_));
    pend64_ = true;
    pend16_ = false;
  } else if (is8 && variant == V16) { // IDCT16Px8's goes to Internal[154..185] (MD.cs:3577-3612)
    for (int k = 0; k < 4; k++)
      for (int m = 0; m < 4; m++) sc16_[4 * k + m] = ib_[8 * k + m];
    pend16_ = true;
  }
}
// A coefficient outside int16 in a frame that ships literal values.  The transform of such a block leaves the clamp table's domain whatever
// the prediction when some residual is beyond +-319 (index = 0x40 + pixel + residual must stay in [0, 384), MobiConst.cs:587, MD.cs:3551):
// the reference throws at this block, and this library rejects the frame with MOBI_E_CLAMP as it does when the kernels find the fault.  A
// block that stays within +-319 everywhere with a coefficient that large would need its sums to cancel through 32-bit wrap-around: the one
// input left that is refused (MOBI_REFUSE_RUN).  ib_ = the block as its transform finds it.
bool MobiStreamParser::surely_faults(bool is8, int variant) const {
  int res_min = 0, res_max = 0;
  auto see = [&](int r) { res_min = std::min(res_min, r); res_max = std::max(res_max, r); };
  if (variant == 0) { // IDCT1Px8 / IDCT1Px4 (MD.cs:3710-3725, 3787-3798)
    see(((int)ib_[0] + 32) >> 6);
  } else if (is8) { // every 8x8 variant equals the full transform on the coefficients it reads (tests/test_oracle_identities.py)
    int c[64], t[64], in[8], out[8];
    for (int p = 0; p < 64; p++) {
      const bool read = variant == 3 || p == 0 || (variant == 1 && (p == 1 || p == 8)) || (variant == 2 && (p & 7) < 4 && p < 32);
      c[p] = read ? (int)ib_[p] : 0;
    }
    for (int k = 0; k < 8; k++) {
      for (int m = 0; m < 8; m++) in[m] = c[8 * k + m];
      if (k == 0) in[0] += 32;
      mobi_bfly8(in, out);
      for (int m = 0; m < 8; m++) t[8 * m + k] = out[m];
    }
    for (int i = 0; i < 8; i++) {
      mobi_bfly8(&t[8 * i], out);
      for (int j = 0; j < 8; j++) see(out[j] >> 6);
    }
  } else {
    int t[16], in[4], out[4];
    for (int k = 0; k < 4; k++) {
      for (int m = 0; m < 4; m++) in[m] = (int)ib_[4 * k + m];
      if (k == 0) in[0] += 32;
      mobi_bfly4(in, out);
      for (int m = 0; m < 4; m++) t[4 * m + k] = out[m];
    }
    for (int i = 0; i < 4; i++) {
      mobi_bfly4(&t[4 * i], out);
      for (int j = 0; j < 4; j++) see(out[j] >> 6);
    }
  }
  return res_min < -319 || res_max > 319;
}
// A frame in which some block's stores left their place (frame_literal_): every residual of the frame is shipped DEQUANTISED, as the value
// the transform reads, and the frame's macroblocks name scale row MOBI_SCALE_LITERAL (all ones) instead of their quantiser -- the
// kernels multiply a "level" by a scale of one and never know.  (Per frame, not per block: the octet kernel takes one scale row for its
// eight macroblocks.)  A value that does not fit the level's 16 bits is the one thing that cannot travel: refused.
void MobiStreamParser::literal_frame(ParsedFrame &out) {
  mobi_literal_frame_count.fetch_add(1, std::memory_order_relaxed);
  int32_t sc[MOBI_SCALE_STRIDE];
  mobi_build_scale_table((int)tq_, sc);
  for (MbDesc &d : out.desc) {
    const bool intra = (d.w1 & 1) == MOBI_MB_INTRA;
    const uint32_t nl = (d.w1 >> 1) & 0x7F, dual = (d.w1 >> 26) & 3, t8 = (d.w1 >> 14) & 0x3F;
    uint32_t *w = out.payload.data() + d.payload_off + (intra ? MOBI_INTRA_RECORDS : (nl > 1 && !dual) ? MOBI_MV_CELLS : 0);
    for (uint32_t i = 0, n = d.w2 & 0x3FF; i < n; i++) {
      if (w[i] & 0x8000u) { w[i] &= ~0x8000u; continue; }
      const int t = (int)(w[i] & 0x1FF), p = t & 63;
      const int32_t v = sc[((t8 >> (t >> 6)) & 1) ? p : 64 + (p & 15)] * (int32_t)(int16_t)(w[i] >> 16);
      if (v != (int16_t)v) { // an ordinary block with a coefficient beyond int16: resid_block looked at its transform (big_unsure_)
        if (big_unsure_) refuse(MOBI_REFUSE_RUN);
        frame_fault_ = true;
        w[i] = (uint32_t)t; // (level 0: the frame is rejected, the word no longer matters)
        continue;
      }
      w[i] = (uint32_t)t | ((uint32_t)v << 16);
    }
    d.w1 = (d.w1 & ~(63u << 20)) | ((uint32_t)MOBI_SCALE_LITERAL << 20);
  }
}
void MobiStreamParser::resid_area(int area) { // loc_11652C, MD.cs:2909-2929
  if (win_ >> 31) {
    win_ += win_;
    nbr_--;
    t8mask_ |= 1u << area;
    resid_block(area, 0, true);
  } else {
    uint32_t u = ue();
    if (u >= sizeof(mobi_cbp4_inter)) fail(MOBI_E_INDEX);
    uint32_t m = mobi_cbp4_inter[u];
    for (int sub = 0; sub < 4; sub++)
      if ((m >> sub) & 1) resid_block(area, sub, false);
  }
}
void MobiStreamParser::p_residual() { // loc_1161A0, MD.cs:1818-1833
  uint32_t u = ue();
  if (u >= sizeof(mobi_cbp_inter)) fail(MOBI_E_INDEX);
  cbp6_ = mobi_cbp_inter[u];
  for (int area = 0; area < 6; area++)
    if ((cbp6_ >> area) & 1) resid_area(area);
}

// ------------------------------------------------------------------ intra syntax
// Reads PredictIntra (MD.cs:1883-2774) / the plane predictors (:3017-3327) make outside the block:
// the top row needs Offset-Stride >= 0, the left column Offset-1 >= 0; anything else is in range.
void MobiStreamParser::check_intra_reads(int mode, long off, bool) const {
  static const bool top[10] = {1, 0, 1, 0, 0, 1, 1, 1, 1, 0}, left[10] = {0, 1, 1, 0, 1, 1, 1, 1, 0, 0};
  if (top[mode] && off < g_.stride) fail(MOBI_E_INDEX);
  if (left[mode] && off < 1) fail(MOBI_E_INDEX);
}
// predicted-mode code shared by loc_116220 / loc_116368 / sub_1163DC (MD.cs:1840-1859, 2785-2804, 2841-2858)
int MobiStreamParser::pmode(int ci, bool four) {
  int pred = std::min(mcache_[ci - 8], mcache_[ci - 1]);
  if (pred == 9) pred = 3;
  int v = (int)(win_ >> 28), nb = 1, mode = pred;
  if (v >= pred) v++;
  if (v < 9) { mode = v; nb = 4; }
  if (four) mcache_[ci] = (uint8_t)mode;
  else mcache_[ci] = mcache_[ci + 1] = mcache_[ci + 8] = mcache_[ci + 9] = (uint8_t)mode;
  take(nb);
  return mode;
}
// A plane parameter for record r (24: the 16x16 plane's): the record's 16-bit field, or -- a code of 33 bits and more -- a wide parameter
// behind the macroblock's level words (mobi_cmd.h).  Returns the bits to OR into the record (MbDesc.w3 for r = 24: shifted by the caller).
uint32_t MobiStreamParser::plane_param(int p, int r) {
  if (p >= -32768 && p <= 32767) return (uint32_t)(uint16_t)(int16_t)p << 16;
  if (!any_wide_) memset(wide_, 0, sizeof(wide_));
  any_wide_ = true;
  frame_host_only_ = true;
  wide_[r] = p;
  return MOBI_REC_WIDE;
}
// sub_116508 (MD.cs:2869-2896) or a bare PredictIntra: one 8x8 area whose mode is already known
void MobiStreamParser::intra_area_fixed(int area, int mode, bool coded) {
  if (!coded) {
    check_intra_reads(mode, area_offset(area, 0), false);
    recs_[area * 4] |= mobi_intra_rec(mode, 0, 0, 0, 0);
    return;
  }
  if (win_ >> 31) {
    win_ += win_;
    nbr_--;
    check_intra_reads(mode, area_offset(area, 0), false);
    recs_[area * 4] |= mobi_intra_rec(mode, 1, 0, 0, 0);
    cbp6_ |= 1u << area;
    t8mask_ |= 1u << area;
    resid_block(area, 0, true);
  } else {
    uint32_t u = ue();
    if (u >= sizeof(mobi_cbp4_intra)) fail(MOBI_E_INDEX);
    uint32_t m4 = mobi_cbp4_intra[u];
    for (int sub = 0; sub < 4; sub++) {
      check_intra_reads(mode, area_offset(area, sub), true);
      int c = (m4 >> sub) & 1;
      recs_[area * 4 + sub] |= mobi_intra_rec(mode, c, 1, 0, 0);
      if (c) {
        cbp6_ |= 1u << area;
        resid_block(area, sub, false);
      }
    }
  }
}
void MobiStreamParser::intra_chroma(uint32_t cbp) { // loc_116290, MD.cs:1864-1880
  int m = (int)(win_ >> 29);
  take(3);
  if (m == 2) {
    m = 9;
    for (int area = 4; area < 6; area++) {
      const int p = se();
      check_intra_reads(2, area_offset(area, 0), false);
      recs_[area * 4] |= mobi_intra_rec(0, 0, 0, 1, 0) | plane_param(p, area * 4);
    }
  }
  intra_area_fixed(4, m, (cbp >> 4) & 1);
  intra_area_fixed(5, m, (cbp >> 5) & 1);
}
void MobiStreamParser::intra_full() { // DecIntraFullBlockPMode, MD.cs:1759-1786
  uint32_t u = ue();
  if (u >= sizeof(mobi_cbp_intra)) fail(MOBI_E_INDEX);
  uint32_t cbp = mobi_cbp_intra[u];
  int m = (int)(win_ >> 29);
  take(3);
  if (m == 2) {
    m = 9;
    const int p = se();
    check_intra_reads(2, cur_off_, false);
    const uint32_t pp = plane_param(p, 24);
    w3_ = 1u | (pp == MOBI_REC_WIDE ? MOBI_W3_WIDE : pp);
  }
  for (int k = 0; k < 4; k++) intra_area_fixed(k, m, (cbp >> k) & 1);
  intra_chroma(cbp);
}
void MobiStreamParser::intra_sub() { // DecIntraSubBlockPMode, MD.cs:1789-1807
  uint32_t u = ue();
  if (u >= sizeof(mobi_cbp_intra)) fail(MOBI_E_INDEX);
  uint32_t cbp = mobi_cbp_intra[u];
  static const int ci[4] = {9, 0xB, 0x19, 0x1B}, d5[4] = {0, 1, 8, 9};
  for (int k = 0; k < 4; k++) {
    bool coded = (cbp >> k) & 1;
    bool whole = true;
    if (coded) { // loc_116368, MD.cs:2776
      if (win_ >> 31) { win_ <<= 1; nbr_--; }
      else whole = false;
    }
    if (whole) {
      int m = pmode(ci[k], false);
      uint32_t pp = 0;
      if (m == 2) pp = plane_param(se(), k * 4); // the predictor itself reads its parameter (MD.cs:1915-1919)
      check_intra_reads(m, area_offset(k, 0), false);
      recs_[k * 4] |= mobi_intra_rec(m, coded, 0, 0, 0) | pp;
      if (coded) {
        cbp6_ |= 1u << k;
        t8mask_ |= 1u << k;
        resid_block(k, 0, true);
      }
    } else {
      uint32_t u4 = ue();
      if (u4 >= sizeof(mobi_cbp4_intra)) fail(MOBI_E_INDEX);
      uint32_t m4 = mobi_cbp4_intra[u4];
      for (int sub = 0; sub < 4; sub++) {
        int m = pmode(ci[k] + d5[sub], true);
        uint32_t pp = 0;
        if (m == 2) pp = plane_param(se(), k * 4 + sub);
        check_intra_reads(m, area_offset(k, sub), true);
        int c = (m4 >> sub) & 1;
        recs_[k * 4 + sub] |= mobi_intra_rec(m, c, 1, 0, 0) | pp;
        if (c) {
          cbp6_ |= 1u << k;
          resid_block(k, sub, false);
        }
      }
    }
  }
  intra_chroma(cbp);
}

// ------------------------------------------------------------------ frames (MD.cs:97-259)
void MobiStreamParser::parse_p(ParsedFrame &out) {
  if (--nbr_ < 0) fill_bits();
  if (version_ == MOBI_VERSION_MOFLEX3DS) {
    uint32_t q = quant_;
    int dq = se();
    if (q == 0) setup_quant(q);
    else if (dq != 0) setup_quant(q + (uint32_t)dq);
  } else {
    int dq = se();
    if (dq != 0) setup_quant(quant_ + (uint32_t)dq);
  }
  vlc_table_ = 0;
  i218_ = 0;
  std::fill(mvc_.begin(), mvc_.end(), 0);
  out.hdr.frame_type = 0;
  for (int mb = 0; mb < g_.mbw * g_.mbh; mb++) {
    int mx = mb % g_.mbw;
    const int *e = &mvc_[2 * mx]; // entries: left, top, top-right (MD.cs:163-169)
    auto med3 = [](int a, int b, int c) { return std::max(std::min(a, b), std::min(std::max(a, b), c)); };
    predx_ = med3(e[0], e[2], e[4]);
    predy_ = med3(e[1], e[3], e[5]);
    int slot = 2 * (mx + 1);
    mvc_[slot] = mvc_[slot + 1] = 0;
    begin_mb(mb, MOBI_MB_INTER);
    pblock(0, 0, 0, 0, slot);
    end_mb();
  }
}
void MobiStreamParser::parse_i(ParsedFrame &out) {
  yuvfmt_ = win_ >> 31;
  win_ += win_;
  vlc_table_ = (int)(win_ >> 31);
  i218_ = (uint32_t)vlc_table_;
  win_ += win_;
  nbr_ -= 3;
  if (nbr_ < 0) fill_bits();
  uint32_t q = win_ >> 26;
  take(6);
  if (quant_ != q) setup_quant(q);
  out.hdr.frame_type = 1;
  for (int mb = 0; mb < g_.mbw * g_.mbh; mb++) {
    bool sub = (win_ >> 31) == 1;
    win_ += win_;
    nbr_--;
    if (nbr_ < 0) fill_bits();
    begin_mb(mb, MOBI_MB_INTRA);
    if (sub) intra_sub(); else intra_full();
    end_mb();
  }
}

// Dependency levels: an intra MB may run once every raster-earlier MB whose pixels its halo touches
// has run.  Inter MBs all run first (level 0).  Raster-later owners are "not yet decoded" = 0 in the
// reference (fresh plane, MD.cs:107) and are masked by the kernel, so they are not dependencies.
void MobiStreamParser::finish_levels(ParsedFrame &out) {
  const int n = (int)out.desc.size();
  const long S = g_.stride;
  std::vector<uint16_t> level(n, 0);
  std::vector<uint8_t> flag(n, 0); // launch-item flags: 2 = has intra dependencies, 4 = has intra dependents
  int maxl = 0, n_intra = 0;
  for (int mb = 0; mb < n; mb++) {
    if ((out.desc[mb].w1 & 1) != MOBI_MB_INTRA) continue;
    n_intra++;
    long off = (long)(mb / g_.mbw) * 16 * S + (mb % g_.mbw) * 16;
    int lv = 0;
    uint16_t deps[MOBI_INTRA_DEPS];
    int n_deps = 0;
    auto dep = [&](int o) {
      if (o < 0 || o >= mb) return; // raster-later owners read as the fresh plane's zeros (the kernel masks them)
      if (level[o] > lv) lv = level[o]; // levels order intra macroblocks only (inter ones are level 0) ...
      // ... but the dependency list names every raster-earlier owner: in a one-launch step the inter quads run alongside
      for (int k = 0; k < n_deps; k++)
        if ((deps[k] & 0x1FFF) == o) return;
      if (n_deps == MOBI_INTRA_DEPS) fail(MOBI_E_UNSUPPORTED); // cannot happen: the halo touches at most 7 macroblocks
      deps[n_deps++] = (uint16_t)(o | (level[o] == 0 ? MOBI_DEP_INTER : 0));
      if (level[o] != 0) { flag[mb] |= 2; flag[o] |= 4; }
    };
    // The halo is the row above (columns -1 .. +23 luma, -1 .. +15 chroma) and the columns left and right of the macroblock.
    // Its owners change only at 16-pixel (8 for chroma) boundaries and, in the side columns, between the first row and the
    // rest (row wrap when width == stride), so these probes meet every distinct owner, in the order a full scan would.
    dep(g_.owner_luma(off - S - 1));
    dep(g_.owner_luma(off - S));
    dep(g_.owner_luma(off - S + 16));
    dep(g_.owner_luma(off - 1));
    dep(g_.owner_luma(off + 16));
    dep(g_.owner_luma(off + S - 1));
    dep(g_.owner_luma(off + S + 16));
    for (int v = 0; v < 2; v++) {
      const long base = off / 2 + v * (S / 2);
      dep(g_.owner_chroma(base - S - 1));
      dep(g_.owner_chroma(base - S));
      dep(g_.owner_chroma(base - S + 8));
      dep(g_.owner_chroma(base - 1));
      dep(g_.owner_chroma(base + 8));
      dep(g_.owner_chroma(base + S - 1));
      dep(g_.owner_chroma(base + S + 8));
    }
    level[mb] = (uint16_t)(lv + 1);
    if (lv + 1 > maxl) maxl = lv + 1;
    for (int k = n_deps; k < MOBI_INTRA_DEPS; k++) deps[k] = MOBI_DEP_NONE;
    MbDesc &d = out.desc[mb];
    d.w4 = deps[0] | ((uint32_t)deps[1] << 16);
    d.w5 = deps[2] | ((uint32_t)deps[3] << 16);
    d.w6 = deps[4] | ((uint32_t)deps[5] << 16);
    d.w7 = deps[6] | ((uint32_t)deps[7] << 16);
  }
  // Launch order inside a level: by CLASS = (at a picture edge ? 8 : 0) + (has intra dependents ? 4 : 0) + min(split areas, 3).  A wave of
  // mobi_recon_intra carries four macroblocks and runs as many steps as the longest of them has (6 + 3 per split area); three in four have
  // no split area, but four taken as they come have one somewhere in 71 % of the waves.  Sorted, the waves of one kind run their own
  // number of steps; and a wave in which nobody has dependents does not write through, drain and publish (r04).
  auto klass = [&](int mb) {
    const MbDesc &d = out.desc[mb];
    int splits = 0;
    for (int a = 0; a < 6; a++) splits += (out.payload[d.payload_off + 4 * a] >> 5) & 1; // the area's first block record, bit 5 (mobi_kernels.hip, the schedule)
    const int mbx = mb % g_.mbw;
    const bool interior = mbx >= 1 && mbx + 1 < g_.mbw && mb >= g_.mbw;
    return (interior ? 0 : 8) + ((flag[mb] & 4) ? 4 : 0) + std::min(splits, 3);
  };
  out.class_start.assign((size_t)(maxl + 1) * MOBI_INTRA_CLASSES + 1, 0); // [L * MOBI_INTRA_CLASSES + class] -> first index into intra_mbs / intra_items; [.. + 1] = its end
  std::vector<uint8_t> kl(n, 0);
  for (int mb = 0; mb < n; mb++)
    if (level[mb]) {
      kl[mb] = (uint8_t)klass(mb);
      out.class_start[(size_t)level[mb] * MOBI_INTRA_CLASSES + kl[mb] + 1]++;
    }
  for (size_t k = 1; k < out.class_start.size(); k++) out.class_start[k] += out.class_start[k - 1];
  out.level_start.assign(maxl + 2, 0);
  for (int l = 1; l <= maxl + 1; l++) out.level_start[l] = l <= maxl ? out.class_start[(size_t)l * MOBI_INTRA_CLASSES] : (uint32_t)n_intra;
  out.intra_mbs.resize(n_intra);
  std::vector<uint32_t> cursor(out.class_start.begin(), out.class_start.end() - 1);
  out.intra_items.resize((size_t)n_intra * 4);
  for (int mb = 0; mb < n; mb++)
    if (level[mb]) {
      const uint32_t at = cursor[(size_t)level[mb] * MOBI_INTRA_CLASSES + kl[mb]]++;
      out.intra_mbs[at] = (uint32_t)mb;
      const MbDesc d = out.desc[mb];
      uint32_t *it = &out.intra_items[(size_t)at * 4];
      it[0] = (uint32_t)mb;
      it[1] = d.w1;
      it[2] = d.payload_off;
      it[3] = (d.w3 & (0xFFFF0001u | MOBI_W3_WIDE)) | flag[mb] | (kl[mb] >= 8 ? 8u : 0u) | ((d.w2 & 0x3FFu) << 5);
      out.desc[mb].w3 |= flag[mb]; // (the device parsers' item lists carry no flags: mobi_recon_intra_cl reads these two bits from the descriptor)
    }
  out.hdr.n_mbs = (uint32_t)n;
  out.hdr.n_intra = (uint32_t)n_intra;
  out.hdr.n_levels = (uint32_t)maxl;
  out.hdr.payload_words = (uint32_t)out.payload.size();
  out.hdr.quantizer = quant_;
  out.hdr.cmd_bytes = (uint32_t)out.cmd_bytes();
}

int MobiStreamParser::parse_frame(const uint8_t *data, size_t len, int32_t *offset, ParsedFrame &out) {
  out.clear();
  if (version_ != MOBI_VERSION_MODSDS && version_ != MOBI_VERSION_MOFLEX3DS) return MOBI_E_VERSION;
  data_ = data;
  len_ = (long)len;
  off_ = *offset;
  out_ = &out;
  int rc = MOBI_OK;
  frames_started_++; // ring rotation + fresh planes happen before anything can throw (MD.cs:102-108)
  try {
    nbr_ = 0;
    win_ = data_u16(off_);
    off_ += 2;
    win_ <<= 16;
    bool iframe = (win_ >> 31) == 1;
    win_ += win_;
    frame_literal_ = frame_fault_ = big_unsure_ = frame_host_only_ = false;
    if (iframe) parse_i(out); else parse_p(out);
    if (frame_literal_) literal_frame(out);
    if (frame_fault_) fail(MOBI_E_CLAMP); // (behind the whole parse, where the kernels' own clamp faults are reported: Offset is the frame's end)
    finish_levels(out);
  } catch (const Err &e) {
    rc = e.code;
  }
  *offset = off_;
  last_frame_ok_ = rc == MOBI_OK;
  return rc;
}

// Dequant scales by natural coefficient index for quantizer q (MD.cs:3892-3911), the table the kernels index
// with MbDesc.w1[25:20].  q outside [12,53] is never used by a residual (resid_block rejects it).
void mobi_build_scale_table(int q, int32_t out[MOBI_SCALE_STRIDE]) {
  memset(out, 0, sizeof(int32_t) * MOBI_SCALE_STRIDE);
  if (q == MOBI_SCALE_LITERAL) { // literal frames (MobiStreamParser::literal_frame): the "levels" are the coefficients themselves
    for (int i = 0; i < MOBI_SCALE_STRIDE; i++) out[i] = 1;
    return;
  }
  if (q < 0 || q >= (int)sizeof(mobi_qdiv6)) return;
  const int sh = mobi_qdiv6[q] + 8, m = mobi_qmod6[q];
  for (int i = 0; i < 64; i++) out[mobi_zz8[i]] = (int32_t)((((uint32_t)mobi_dq8[m * 64 + i]) << (sh - 2)) >> 8);
  for (int i = 0; i < 16; i++) out[64 + mobi_zz4[i]] = (int32_t)((((uint32_t)mobi_dq4[m * 16 + i]) << sh) >> 8);
}

// Directional intra predictors (modes 0, 1, 4..8 of PredictIntra, MD.cs:1883-2774) as data: every predicted sample is
// (t0 + t1 + t2 + t3 + 2) >> 2 over four neighbour samples -- F3(a, b, c) = taps a, b, b, c; F2(a, b) = a, a, b, b; a copy =
// a, a, a, a -- so a table of four tile offsets per (mode, block size, sample) replaces the per-sample case analysis of
// mobi_pred_px in the intra kernel (lane-divergent branches there are scalar work: 771 scalar instructions per macroblock in
// r01).  The table is DERIVED from mobi_pred_px by probing it with unit impulses, then checked against it on random
// neighbourhoods; offsets are relative to the block's top-left sample in a tile of the given pitch.
bool mobi_build_intra_taps(int16_t *out, int pitch) {
  static const int modes[MOBI_TAP_MODES] = {0, 1, 4, 5, 6, 7, 8};
  uint32_t rng = 0x4D4F4249u;
  auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return (int)(rng >> 24); };
  for (int mi = 0; mi < MOBI_TAP_MODES; mi++) {
    for (int four = 0; four < 2; four++) {
      const int n = four ? 4 : 8;
      for (int y = 0; y < n; y++) {
        for (int x = 0; x < n; x++) {
          int16_t *e = out + 4 * ((four ? MOBI_TAP_4X4 + mi * 16 : mi * 64) + y * n + x);
          int nt = 0;
          // candidate neighbours: the row above from the corner to column n + 4 (mode 8 of an 8x8 block reads 13 of them), then the left column
          auto probe = [&](int py, int px) {
            auto nb = [&](int dy, int dx) { return (dy == py && dx == px) ? 4 : 0; };
            return mobi_pred_px(modes[mi], n, y, x, 1, 1, nb);
          };
          auto add = [&](int py, int px) {
            const int w = probe(py, px);
            if (w < 0 || w > 4) return false; // 3: a clamped index names one neighbour twice (F3(L6, L7, L7))
            for (int k = 0; k < w; k++) {
              if (nt == 4) return false;
              e[nt++] = (int16_t)(py * pitch + px);
            }
            return true;
          };
          for (int px = -1; px <= n + 4; px++)
            if (!add(-1, px)) return false;
          for (int py = 0; py < n; py++)
            if (!add(py, -1)) return false;
          if (nt != 4) return false;
          for (int trial = 0; trial < 8; trial++) { // the table against the function it came from
            int top[16], left[8];
            for (int k = 0; k < 16; k++) top[k] = rnd();
            for (int k = 0; k < 8; k++) left[k] = rnd();
            auto nb = [&](int dy, int dx) { return dy < 0 ? top[dx + 1] : left[dy]; };
            int sum = 2;
            for (int k = 0; k < 4; k++) {
              const int o = e[k];
              const int dy = (o + pitch + 1) / pitch - 1, dx = o - dy * pitch; // o = dy * pitch + dx with dx in [-1, n + 4]
              sum += nb(dy, dx);
            }
            if ((sum >> 2) != mobi_pred_px(modes[mi], n, y, x, 1, 1, nb)) return false;
          }
        }
      }
    }
  }
  return true;
}

// The tables the device-side parser (mobi_dparse.hip) keeps in LDS, as one blob (layout: MOBI_DT_* in mobi_dparse.h).
void mobi_dparse_build_tables(int version, uint8_t out[MOBI_DT_BYTES]) {
  const int v = (version == MOBI_VERSION_MOFLEX3DS) ? 0 : 1;
  memset(out, 0, MOBI_DT_BYTES);
  memcpy(out + MOBI_DT_A0, mobi_vx2table0_a, sizeof(mobi_vx2table0_a));
  memcpy(out + MOBI_DT_A1, mobi_vx2table1_a, sizeof(mobi_vx2table1_a));
  memcpy(out + MOBI_DT_B0, mobi_vx2table0_b, sizeof(mobi_vx2table0_b));
  memcpy(out + MOBI_DT_B1, mobi_vx2table1_b, sizeof(mobi_vx2table1_b));
  memcpy(out + MOBI_DT_PLUT, mobi_part_lut[v], sizeof(mobi_part_lut[v]));
  memcpy(out + MOBI_DT_PBITS, mobi_part_bits[v], sizeof(mobi_part_bits[v]));
  memcpy(out + MOBI_DT_PSHIFT, mobi_part_shift[v], sizeof(mobi_part_shift[v]));
  memcpy(out + MOBI_DT_PNB, mobi_part_nbits_len[v], sizeof(mobi_part_nbits_len[v]));
  memcpy(out + MOBI_DT_CBP_I, mobi_cbp_intra, sizeof(mobi_cbp_intra));
  memcpy(out + MOBI_DT_CBP_P, mobi_cbp_inter, sizeof(mobi_cbp_inter));
  memcpy(out + MOBI_DT_CBP4_I, mobi_cbp4_intra, sizeof(mobi_cbp4_intra));
  memcpy(out + MOBI_DT_CBP4_P, mobi_cbp4_inter, sizeof(mobi_cbp4_inter));
  memcpy(out + MOBI_DT_ZZ8, mobi_zz8, sizeof(mobi_zz8));
  memcpy(out + MOBI_DT_ZZ4, mobi_zz4, sizeof(mobi_zz4));
  static_assert(sizeof(mobi_vx2table0_a) == 8192 && sizeof(mobi_part_lut[0]) == 1024 && sizeof(mobi_part_bits[0]) == 192, "blob layout");
}
