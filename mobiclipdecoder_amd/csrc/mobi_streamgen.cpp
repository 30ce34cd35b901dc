// mobi_streamgen.cpp -- seeded synthetic Mobiclip bitstream generator (see mobi_streamgen.h).
//
// Writer side of the syntax parsed by the reference decoder
// (LibMobiclip/Codec/Mobiclip/MobiclipDecoder.cs, "MD.cs"; grammar: SURVEY.md appendix A).
// Bit packing per LibMobiclip/Codec/Mobiclip/BitWriter.cs:16-65.  The generator mirrors the
// decoder-side state that the syntax depends on (MV row cache + median predictor MD.cs:163-208,
// intra-mode byte cache MD.cs:1835-1862/3913-3924, quantizer MD.cs:3884-3891) so that it can code
// the symbols it draws.  All code words come from inverting the decoder LUTs in mobi_tables.h.
#include "mobi_streamgen.h"
#include "mobi_tables.h"

#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

struct Rng { // xoshiro256**
  uint64_t s[4];
  static uint64_t splitmix(uint64_t &x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  explicit Rng(uint64_t seed) { for (auto &v : s) v = splitmix(seed); }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() {
    uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
  }
  uint32_t below(uint32_t n) { return n ? (uint32_t)((next() >> 11) % n) : 0; }
  bool chance(int permille) { return (int)below(1000) < permille; }
  int range(int lo, int hi) { return lo + (int)below((uint32_t)(hi - lo + 1)); }
};

struct BitWriter { // BitWriter.cs:16-65
  std::vector<uint8_t> out;
  uint32_t acc = 0;
  int n = 0;
  void put(uint32_t v, int nb) {
    while (nb > 16) { put(v >> (nb - 16), 16); nb -= 16; v &= (1u << nb) - 1; }
    if (nb <= 0) return;
    acc |= (v & ((1u << nb) - 1)) << ((32 - nb) - n);
    n += nb;
    if (n >= 16) flush16();
  }
  void flush16() {
    out.push_back((uint8_t)(acc >> 16));
    out.push_back((uint8_t)(acc >> 24));
    acc <<= 16;
    n -= 16;
  }
  void align() { if (n > 0) { flush16(); n = 0; acc = 0; } }
  static int blen(uint32_t v) { int k = 0; while (v) { v >>= 1; k++; } return k; }
  void ue(uint32_t v) { // WriteVarIntUnsigned
    int nb = blen((v + 1) / 2);
    put(0, nb); put(1, 1); put(v - ((1u << nb) - 1), nb);
  }
  void se(int v) { // WriteVarIntSigned
    uint32_t val = (v <= 0) ? (uint32_t)(1 - v * 2) : (uint32_t)(v * 2);
    int nb = blen(val / 2);
    put(0, nb); put(1, 1); put(val - (1u << nb), nb);
  }
};

// ---- inverse of the residual run/level/last LUT (MobiConst.cs:10-14 entry format) ----
struct VlcInv {
  // key (last,run,level) -> code prefix (nb-1 bits, sign follows); nbits==0: not in table
  uint16_t code[2][64][32];
  uint8_t nbits[2][64][32]; // total length incl. sign
  void build(const uint16_t *A) {
    memset(code, 0, sizeof(code));
    memset(nbits, 0, sizeof(nbits));
    for (int i = 0; i < 4096; i++) {
      if ((i >> 5) == 3) continue; // 0000011 prefix is the escape (MD.cs:3342)
      uint16_t e = A[i];
      int nb = e & 0xF, v = (e >> 4) & 0x1F, run = (e >> 9) & 0x3F, last = e >> 15;
      if (nb < 2 || nb > 12 || v == 0) continue;
      if (nbits[last][run][v]) continue;
      nbits[last][run][v] = (uint8_t)nb;
      code[last][run][v] = (uint16_t)(i >> (12 - (nb - 1)));
    }
  }
};

struct PartInv { uint8_t code[10]; uint8_t nbits[10]; }; // per shape: code word for partition codes 0..9

struct Sym { int last, run, level; };

// full inverse transforms (second statement of MD.cs:3435-3561 / :3728-3784, used only to keep
// generated residuals inside the reference's clamp-table domain)
void bfly8(const int in[8], int out[8]) {
  int a0 = in[0] + in[4], a1 = in[0] - in[4];
  int a2 = in[2] + (in[6] >> 1), a3 = (in[2] >> 1) - in[6];
  int e0 = a0 + a2, e1 = a1 + a3, e2 = a1 - a3, e3 = a0 - a2;
  int b0 = in[1] + in[7] - in[3] - (in[3] >> 1);
  int b1 = in[7] - in[1] + in[5] + (in[5] >> 1);
  int b2 = in[5] - (in[7] + (in[7] >> 1)) - in[3];
  int b3 = in[3] + in[5] + in[1] + (in[1] >> 1);
  int o0 = b2 + (b3 >> 2), o3 = b3 - (b2 >> 2);
  int o1 = b0 + (b1 >> 2), o2 = (b0 >> 2) - b1;
  out[0] = e0 + o3; out[7] = e0 - o3;
  out[1] = e1 + o2; out[6] = e1 - o2;
  out[2] = e2 + o1; out[5] = e2 - o1;
  out[3] = e3 + o0; out[4] = e3 - o0;
}
void idct8_full(const int c[64], int res[64]) {
  int tmp[64], in[8], out[8];
  for (int k = 0; k < 8; k++) {
    for (int m = 0; m < 8; m++) in[m] = c[8 * k + m];
    if (k == 0) in[0] += 32;
    bfly8(in, out);
    for (int m = 0; m < 8; m++) tmp[8 * m + k] = out[m];
  }
  for (int i = 0; i < 8; i++) {
    bfly8(&tmp[8 * i], out);
    for (int j = 0; j < 8; j++) res[8 * i + j] = out[j] >> 6;
  }
}
void bfly4(const int in[4], int out[4]) {
  int a = in[0] + in[2], b = in[0] - in[2];
  int c = (in[1] >> 1) - in[3], d = in[1] + (in[3] >> 1);
  out[0] = a + d; out[3] = a - d; out[1] = b + c; out[2] = b - c;
}
void idct4_full(const int c[16], int res[16]) {
  int tmp[16], in[4], out[4];
  for (int k = 0; k < 4; k++) {
    for (int m = 0; m < 4; m++) in[m] = c[4 * k + m];
    if (k == 0) in[0] += 32;
    bfly4(in, out);
    for (int m = 0; m < 4; m++) tmp[4 * m + k] = out[m];
  }
  for (int i = 0; i < 4; i++) {
    bfly4(&tmp[4 * i], out);
    for (int j = 0; j < 4; j++) res[4 * i + j] = out[j] >> 6;
  }
}

struct Gen {
  mobi_gen_params P;
  Rng rng;
  BitWriter bw;
  int S, mbw, mbh, ver; // ver: 0 Moflex, 1 Mods (table index)
  // mirrored decoder state
  uint8_t mcache[40];   // intra-mode byte cache = bytes of Internal[0..9]
  uint32_t quant = 0;   // Quantizer (0 = never set)
  int vlc_table = 0;    // Internal[218]
  int frames_done = 0;  // number of decoded frames so far (reference slots 1..min(5,frames_done) exist)
  std::vector<int> mvc; // MV row cache, 2 ints per entry, mbw+2 entries
  int predx = 0, predy = 0;
  uint32_t scale8[64], scale4[16]; // dequant scale by scan position (word >> 8), MD.cs:3897-3912
  VlcInv vinv[2];
  PartInv pinv[16];
  uint8_t inv_cbp_inter[64], inv_cbp_intra[64], inv_cbp4_inter[16], inv_cbp4_intra[16];

  explicit Gen(const mobi_gen_params &p) : P(p), rng(p.seed) {
    S = (p.width <= 256) ? 256 : (p.width <= 512) ? 512 : 1024; // MD.cs:50-52
    mbw = (int)p.width / 16;
    mbh = (int)p.height / 16;
    ver = (p.version == 2) ? 0 : 1;
    memset(mcache, 0, sizeof(mcache));
    mvc.assign(2 * (mbw + 2), 0);
    vinv[0].build(mobi_vx2table0_a);
    vinv[1].build(mobi_vx2table1_a);
    for (int s = 0; s < 16; s++) {
      int peek = 32 - mobi_part_shift[ver][s];
      for (int c = 0; c < 10; c++) { pinv[s].code[c] = 0; pinv[s].nbits[c] = 0; }
      for (int i = (1 << peek) - 1; i >= 0; i--) {
        int c = mobi_part_lut[ver][s][i];
        if (c >= mobi_part_nbits_len[ver][s]) continue;
        int nb = mobi_part_bits[ver][s][c];
        if (nb == 0 || nb > peek) continue;
        pinv[s].nbits[c] = (uint8_t)nb;
        pinv[s].code[c] = (uint8_t)(i >> (peek - nb));
      }
    }
    for (int i = 63; i >= 0; i--) { inv_cbp_inter[mobi_cbp_inter[i]] = (uint8_t)i; inv_cbp_intra[mobi_cbp_intra[i]] = (uint8_t)i; }
    memset(inv_cbp4_inter, 0, 16); memset(inv_cbp4_intra, 0, 16);
    for (int i = 15; i >= 1; i--) inv_cbp4_inter[mobi_cbp4_inter[i]] = (uint8_t)i; // ue >= 1 (leading 0 bit, MD.cs:2911-2919)
    for (int i = 16; i >= 1; i--) inv_cbp4_intra[mobi_cbp4_intra[i]] = (uint8_t)i;
  }

  // SetupQuantizationTables mirror, MD.cs:3884-3925
  void setup_quant(uint32_t q) {
    if (ver == 0) { if (q < 12) q = 12; if (q > 52) q = 52; }
    quant = q;
    int sh = mobi_qdiv6[q] + 8, m = mobi_qmod6[q];
    for (int i = 0; i < 16; i++) scale4[i] = (((uint32_t)mobi_dq4[(m << 4) + i]) << sh) >> 8;
    sh -= 2;
    for (int i = 0; i < 64; i++) scale8[i] = (((uint32_t)mobi_dq8[(m << 6) + i]) << sh) >> 8;
    mcache[1] = mcache[2] = mcache[3] = mcache[4] = 9;
    mcache[8] = mcache[0x10] = mcache[0x18] = mcache[0x20] = 9;
  }

  // ------------------------------------------------------------ residual blocks
  bool vlc_has(int last, int run, int lv) const { return run >= 0 && run < 64 && lv >= 1 && lv < 32 && vinv[vlc_table].nbits[last][run][lv]; }
  void put_lut(int last, int run, int lv, bool neg) {
    const VlcInv &v = vinv[vlc_table];
    int nb = v.nbits[last][run][lv];
    bw.put(v.code[last][run][lv], nb - 1);
    bw.put(neg ? 1 : 0, 1);
  }
  // ReadDCTMatrix writer, MD.cs:3334-3423
  void put_sym(const Sym &s, int force) {
    const uint8_t *B = vlc_table ? mobi_vx2table1_b : mobi_vx2table0_b;
    int a = std::abs(s.level);
    bool neg = s.level < 0;
    if (force != 3) {
      if (force == 0 && vlc_has(s.last, s.run, a)) { put_lut(s.last, s.run, a, neg); return; }
      int off = B[(s.last << 6) | s.run];
      if (force != 2 && vlc_has(s.last, s.run, a - off)) { // escape "0": level offset
        bw.put(3, 7); bw.put(0, 1);
        put_lut(s.last, s.run, a - off, neg);
        return;
      }
      if (a < 32) {
        int roff = B[0x80 + a + (s.last << 6)];
        if (vlc_has(s.last, s.run - roff, a)) { // escape "10": run offset
          bw.put(3, 7); bw.put(2, 2);
          put_lut(s.last, s.run - roff, a, neg);
          return;
        }
      }
      if (force == 1 && vlc_has(s.last, s.run, a)) { put_lut(s.last, s.run, a, neg); return; }
    }
    bw.put(3, 7); bw.put(3, 2); // escape "11": raw
    bw.put((uint32_t)s.last, 1); bw.put((uint32_t)s.run, 6); bw.put((uint32_t)s.level & 0xFFF, 12);
  }

  // draw one residual block (n = 8 or 4), keep |residual| <= 64 so MinMaxTable stays in its domain
  void gen_block(int n) {
    const int N = n * n;
    const uint8_t *zz = (n == 8) ? mobi_zz8 : mobi_zz4;
    const uint32_t *sc = (n == 8) ? scale8 : scale4;
    int lev[64];
    for (int attempt = 0;; attempt++) {
      memset(lev, 0, sizeof(lev));
      bool dense = (n == 8) && rng.chance(P.dense_prob) && attempt < 4;
      const bool lowfreq = !dense && P.lowfreq_prob > 0 && rng.chance(P.lowfreq_prob); // DC / three lowest positions only
      int span = dense ? 64 : lowfreq ? 3 : (P.scan_span < N ? P.scan_span : N);
      if (span < 1) span = 1;
      int cnt = dense ? 64 : rng.range(1, lowfreq ? 2 : P.max_coefs < span ? P.max_coefs : span);
      if (attempt >= 6) cnt = 1;
      for (int k = 0; k < cnt; k++) {
        int pos = dense ? k : (int)rng.below((uint32_t)span);
        int mag = 1;
        if (attempt < 3) while (mag < 24 && rng.chance(500)) mag++; // geometric(0.5)+1
        lev[pos] = rng.chance(500) ? -mag : mag;
      }
      if (attempt >= 8) { memset(lev, 0, sizeof(lev)); lev[0] = 1; }
      int c[64], res[64];
      memset(c, 0, sizeof(c));
      for (int p = 0; p < N; p++) if (lev[p]) c[zz[p]] = (int)sc[p] * lev[p];
      if (n == 8) idct8_full(c, res); else idct4_full(c, res);
      bool ok = true;
      for (int i = 0; i < N; i++) if (res[i] < -64 || res[i] > 64) ok = false;
      if (ok || attempt >= 9) break;
    }
    int lastpos = -1;
    for (int p = 0; p < N; p++) if (lev[p]) lastpos = p;
    int prev = -1;
    for (int p = 0; p <= lastpos; p++) {
      if (!lev[p]) continue;
      Sym s{p == lastpos, p - prev - 1, lev[p]};
      int force = 0;
      if (rng.chance(P.escape_prob)) force = rng.range(1, 3);
      put_sym(s, force);
      prev = p;
    }
  }

  // loc_11652C writer, MD.cs:2909-2929
  void gen_resid8_or_4x4() {
    if (rng.chance(P.t8_prob)) { bw.put(1, 1); gen_block(8); return; }
    int m = rng.range(1, 15);
    bw.ue(inv_cbp4_inter[m]);
    for (int k = 0; k < 4; k++) if ((m >> k) & 1) gen_block(4);
  }
  // loc_1161A0 writer, MD.cs:1818-1833
  void gen_p_residual() {
    int m = 0;
    for (int k = 0; k < 6; k++) if (rng.chance(P.cbp_prob)) m |= 1 << k;
    bw.ue(inv_cbp_inter[m]);
    for (int k = 0; k < 6; k++) if ((m >> k) & 1) gen_resid8_or_4x4();
  }

  // ------------------------------------------------------------ motion
  // would CopyBlock (MD.cs:418-456) stay inside the allowed area for this leaf?
  bool window_ok(int x, int y, int w, int h, int dx, int dy, int pw, int ph, long plane_len, int col0) const {
    int x0 = x + (dx >> 1), y0 = y + (dy >> 1);
    int cols = w + (dx & 1), rows = h + (dy & 1);
    if (!P.edge_mode) return x0 >= 0 && y0 >= 0 && x0 + cols <= pw && y0 + rows <= ph;
    long first = (long)y0 * S + col0 + x0;
    long last = first + (long)(rows - 1) * S + (cols - 1);
    return first >= 0 && last < plane_len;
  }
  bool mv_ok(int x, int y, int w, int h, int dx, int dy) const {
    int W = (int)P.width, H = (int)P.height;
    if (!window_ok(x, y, w, h, dx, dy, W, H, (long)S * H, 0)) return false;
    int cdx = dx >> 1, cdy = dy >> 1;
    if (!window_ok(x / 2, y / 2, w / 2, h / 2, cdx, cdy, W / 2, H / 2, (long)S * H / 2, 0)) return false;
    return window_ok(x / 2, y / 2, w / 2, h / 2, cdx, cdy, W / 2, H / 2, (long)S * H / 2, S / 2);
  }
  // one MC leaf: codes 0..5 of any ReadPBlock*, MD.cs:400-416
  void gen_leaf(int s, int x, int y, int w, int h, int io, bool allow_skip) {
    int nref = frames_done < 5 ? frames_done : 5;
    if (allow_skip && pinv[s].nbits[0] && mv_ok(x, y, w, h, predx, predy)) {
      bw.put(pinv[s].code[0], pinv[s].nbits[0]);
      mvc[io] = predx; mvc[io + 1] = predy;
      return;
    }
    int ref = 1;
    if (nref > 1 && rng.chance(P.pm_multiref)) ref = rng.range(2, nref);
    int dx = 0, dy = 0;
    for (int tries = 0; tries < 64; tries++) {
      int r = tries < 48 ? P.mv_range : 2;
      int cx = predx + rng.range(-r, r), cy = predy + rng.range(-r, r);
      if (std::abs(cx) > 63 || std::abs(cy) > 63) continue; // keeps every MV delta (incl. the (0,0) fallback) < 128: ue/se codes stay <= 16 bits
      if (mv_ok(x, y, w, h, cx, cy)) { dx = cx; dy = cy; goto found; }
    }
    dx = 0; dy = 0; // (0,0) always reads the co-located block
  found:
    bw.put(pinv[s].code[ref], pinv[s].nbits[ref]);
    bw.se(dx - predx);
    bw.se(dy - predy);
    mvc[io] = dx; mvc[io + 1] = dy;
  }
  // ReadPBlock tree, MD.cs:469-1746. depth_budget: how many more split levels we may take
  void gen_pblock(int wi, int hi, int x, int y, int io, int splits_left, bool force_split) {
    int s = wi * 4 + hi, w = 16 >> wi, h = 16 >> hi;
    bool can8 = h > 2 && pinv[s].nbits[8], can9 = w > 2 && pinv[s].nbits[9];
    bool split = (can8 || can9) && splits_left > 0 && (force_split || rng.chance(600));
    if (!split) { gen_leaf(s, x, y, w, h, io, rng.chance(250)); return; }
    bool use8 = can8 && (!can9 || rng.chance(500));
    if (use8) {
      bw.put(pinv[s].code[8], pinv[s].nbits[8]);
      gen_pblock(wi, hi + 1, x, y, io, splits_left - 1, false);
      gen_pblock(wi, hi + 1, x, y + h / 2, io, splits_left - 1, false);
    } else {
      bw.put(pinv[s].code[9], pinv[s].nbits[9]);
      gen_pblock(wi + 1, hi, x, y, io, splits_left - 1, false);
      gen_pblock(wi + 1, hi, x + w / 2, y, io, splits_left - 1, false);
    }
  }

  // ------------------------------------------------------------ intra
  static bool needs_top(int m) { return m == 0 || m == 2 || m == 5 || m == 6 || m == 7 || m == 8; }
  static bool needs_left(int m) { return m == 1 || m == 2 || m == 4 || m == 5 || m == 6 || m == 7; }
  bool legal_mode(int m, long off) const { // m in 0..8 (8x8 numbering; the 4x4 twins read the same sides)
    if (needs_top(m) && off < S + 1) return false;
    if (needs_left(m) && off < 1) return false;
    return true;
  }
  int pick_mode(long off, bool allow8) {
    if (P.intra_dc_only) return 3;
    for (int t = 0; t < 32; t++) {
      int m = rng.range(0, allow8 ? 8 : 7);
      if (m == 2 && !rng.chance(P.plane_prob)) continue;
      if (legal_mode(m, off)) return m;
    }
    return 3;
  }
  void put_plane_param() { bw.se(rng.range(-6, 6)); }
  // predicted-mode code of loc_116220 / loc_116368 / sub_1163DC, MD.cs:1840-1859
  void put_pmode(int ci, int mode, bool four) {
    int pred = mcache[ci - 8] < mcache[ci - 1] ? mcache[ci - 8] : mcache[ci - 1];
    if (pred == 9) pred = 3;
    if (mode == pred) bw.put(1, 1);
    else bw.put((uint32_t)(mode < pred ? mode : mode - 1), 4);
    if (four) mcache[ci] = (uint8_t)mode;
    else mcache[ci] = mcache[ci + 1] = mcache[ci + 8] = mcache[ci + 9] = (uint8_t)mode;
  }
  // which mode would the decoder infer with a 1-bit code?  (used to keep illegal predictions away)
  int pmode_pred(int ci) const {
    int pred = mcache[ci - 8] < mcache[ci - 1] ? mcache[ci - 8] : mcache[ci - 1];
    return pred == 9 ? 3 : pred;
  }
  // sub_116508 writer (fixed mode), MD.cs:2869-2896
  void gen_intra_full_coded(int mode) {
    if (rng.chance(P.t8_prob)) { bw.put(1, 1); gen_block(8); return; }
    int m = rng.range(0, 15);
    bw.ue(m == 0 ? 2 : inv_cbp4_intra[m]); // ue=2 -> mask 0 (all four predicted, none coded)
    for (int k = 0; k < 4; k++) if ((m >> k) & 1) gen_block(4);
    (void)mode;
  }
  // loc_116290 writer, MD.cs:1864-1880
  void gen_intra_chroma(int cbp, int mbx, int mby) {
    long offU = ((long)mby * 16 * S + mbx * 16) / 2;
    int m;
    for (;;) {
      m = pick_mode(offU, false);
      if (legal_mode(m, offU + S / 2)) break;
    }
    bw.put((uint32_t)m, 3);
    if (m == 2) { put_plane_param(); put_plane_param(); }
    for (int k = 4; k < 6; k++) if ((cbp >> k) & 1) gen_intra_full_coded(m);
  }
  int draw_cbp() { int m = 0; for (int k = 0; k < 6; k++) if (rng.chance(P.cbp_prob)) m |= 1 << k; return m; }
  // DecIntraFullBlockPMode writer, MD.cs:1759-1786
  void gen_intra_full(int mbx, int mby) {
    long off = (long)mby * 16 * S + mbx * 16;
    int cbp = draw_cbp();
    bw.ue(inv_cbp_intra[cbp]);
    int m;
    for (;;) { // one mode for all four luma blocks: must be legal for the top-left one (smallest offset)
      m = pick_mode(off, false);
      if (legal_mode(m, off + 8)) break;
    }
    bw.put((uint32_t)m, 3);
    if (m == 2) put_plane_param();
    for (int k = 0; k < 4; k++) if ((cbp >> k) & 1) gen_intra_full_coded(m);
    gen_intra_chroma(cbp, mbx, mby);
  }
  // DecIntraSubBlockPMode writer, MD.cs:1789-1807 (+ loc_116220 :1835, loc_116368 :2776)
  void gen_intra_sub(int mbx, int mby) {
    long off = (long)mby * 16 * S + mbx * 16;
    static const int bx[4] = {0, 8, 0, 8}, by[4] = {0, 0, 8, 8}, ci[4] = {9, 0xB, 0x19, 0x1B};
    int cbp = draw_cbp();
    bw.ue(inv_cbp_intra[cbp]);
    for (int k = 0; k < 4; k++) {
      long o = off + (long)by[k] * S + bx[k];
      bool coded = (cbp >> k) & 1;
      if (!coded || rng.chance(P.t8_prob)) {
        if (coded) bw.put(1, 1);
        int m = pick_mode(o, true);
        put_pmode(ci[k], m, false);
        if (m == 2) put_plane_param();
        if (coded) gen_block(8);
      } else {
        int m4 = rng.range(0, 15);
        bw.ue(m4 == 0 ? 2 : inv_cbp4_intra[m4]);
        static const int d5[4] = {0, 1, 8, 9}, sx[4] = {0, 4, 0, 4}, sy[4] = {0, 0, 4, 4};
        for (int q = 0; q < 4; q++) {
          long o4 = o + (long)sy[q] * S + sx[q];
          int m = pick_mode(o4, true);
          put_pmode(ci[k] + d5[q], m, true);
          if (m == 2) put_plane_param();
          if ((m4 >> q) & 1) gen_block(4);
        }
      }
    }
    gen_intra_chroma(cbp, mbx, mby);
  }
  void gen_intra_mb(int mbx, int mby) { // I-frame MB: 1 bit selects sub/full, MD.cs:244-249
    bool sub = !P.intra_dc_only && rng.chance(P.intra_sub_prob);
    bw.put(sub ? 1 : 0, 1);
    if (sub) gen_intra_sub(mbx, mby); else gen_intra_full(mbx, mby);
  }

  // ------------------------------------------------------------ frames
  void gen_iframe() {
    bw.put(1, 1);
    bw.put(0, 1);                    // YuvFormat
    vlc_table = rng.chance(P.table1_prob) ? 1 : 0;
    bw.put((uint32_t)vlc_table, 1);
    uint32_t q = quant ? quant : (uint32_t)P.quantizer;
    if (frames_done == 0) q = (uint32_t)P.quantizer;
    bw.put(q, 6);
    if (quant != q) setup_quant(q);
    for (int my = 0; my < mbh; my++)
      for (int mx = 0; mx < mbw; mx++) gen_intra_mb(mx, my);
  }
  void gen_pframe() {
    bw.put(0, 1);
    int dq = 0;
    if (rng.chance(P.qdelta_prob)) {
      dq = rng.range(-3, 3);
      long nq = (long)quant + dq;
      if (nq < 12 || nq > 52) dq = 0;
    }
    bw.se(dq);
    if (ver == 0) { if (quant == 0) setup_quant(0); else if (dq) setup_quant(quant + dq); }
    else if (dq) setup_quant(quant + dq);
    vlc_table = 0;
    std::fill(mvc.begin(), mvc.end(), 0);
    for (int my = 0; my < mbh; my++) {
      for (int mx = 0; mx < mbw; mx++) {
        int io = 2 * mx; // entry mx = left, mx+1 = top (this MB's slot), mx+2 = top-right
        int a[3] = {mvc[io], mvc[io + 2], mvc[io + 4]}, b[3] = {mvc[io + 1], mvc[io + 3], mvc[io + 5]};
        auto med = [](int *v) { if (v[0] > v[1]) std::swap(v[0], v[1]); if (v[1] > v[2]) std::swap(v[1], v[2]); if (v[0] > v[1]) std::swap(v[0], v[1]); return v[1]; };
        predx = med(a); predy = med(b);
        io += 2;
        mvc[io] = 0; mvc[io + 1] = 0;
        int x = mx * 16, y = my * 16;
        int r = (int)rng.below(1000);
        if (r < P.pm_intra) {
          bool sub = rng.chance(P.intra_sub_prob) && pinv[0].nbits[7];
          int c = sub ? 7 : 6;
          bw.put(pinv[0].code[c], pinv[0].nbits[c]);
          if (sub) gen_intra_sub(mx, my); else gen_intra_full(mx, my);
          continue;
        }
        int t1 = P.pm_intra + P.pm_skip, t2 = t1 + P.pm_split1, t3 = t2 + P.pm_deep;
        if (r < t1 && mv_ok(x, y, 16, 16, predx, predy)) gen_leaf(0, x, y, 16, 16, io, true);
        else if (r >= t1 && r < t2) gen_pblock(0, 0, x, y, io, 1, true);
        else if (r >= t2 && r < t3) gen_pblock(0, 0, x, y, io, 3 + (int)rng.below(4), true);
        else gen_leaf(0, x, y, 16, 16, io, false);
        gen_p_residual();
      }
    }
  }
};

} // namespace

extern "C" void mobi_gen_default_params(mobi_gen_params *p, int config, uint64_t seed) {
  memset(p, 0, sizeof(*p));
  p->seed = seed;
  p->n_frames = 33;
  p->quantizer = 25;
  p->pm_skip = 150; p->pm_split1 = 200; p->pm_deep = 50; p->pm_intra = 50; p->pm_multiref = 50;
  p->mv_range = 16;
  p->cbp_prob = 300; p->t8_prob = 800; p->dense_prob = 0; p->max_coefs = 6; p->scan_span = 16;
  p->intra_sub_prob = 500; p->plane_prob = 300; p->intra_dc_only = 0;
  p->edge_mode = 0; p->escape_prob = 20; p->qdelta_prob = 0; p->table1_prob = 0;
  switch (config) {
    case 'A': p->width = 256; p->height = 192; p->version = 1; break;
    case 'C': p->width = 848; p->height = 480; p->version = 2; p->mv_range = 64; p->dense_prob = 300; p->quantizer = 16; break;
    case 'B': default: p->width = 640; p->height = 480; p->version = 2; break;
  }
}

extern "C" int64_t mobi_gen_clip(const mobi_gen_params *p, uint8_t *out, size_t cap, uint32_t *frame_off) {
  if (!p || p->width == 0 || p->height == 0 || (p->width & 15) || (p->height & 15) || p->width > 1024) return -1;
  if (p->version != 1 && p->version != 2) return -1;
  if (p->quantizer < 12 || p->quantizer > 52 || p->n_frames < 1) return -1;
  Gen g(*p);
  size_t total = 0;
  for (int f = 0; f < p->n_frames; f++) {
    if (frame_off) frame_off[f] = (uint32_t)total;
    g.bw = BitWriter();
    bool iframe = (f == 0) || (p->iframe_interval > 0 && f % p->iframe_interval == 0);
    if (iframe) g.gen_iframe(); else g.gen_pframe();
    g.bw.align();
    g.frames_done++;
    size_t n = g.bw.out.size();
    if (out && total + n <= cap) memcpy(out + total, g.bw.out.data(), n);
    total += n;
  }
  if (frame_off) frame_off[p->n_frames] = (uint32_t)total;
  if (total > cap) return -(int64_t)total;
  return (int64_t)total;
}
