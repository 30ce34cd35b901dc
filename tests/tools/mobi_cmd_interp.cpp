// mobi_cmd_interp.cpp -- CPU interpreter of the host->GPU command list.  TEST TOOL ONLY.
//
// Built into tests/tools/libmobi_cmdinterp.so and loaded only by the CPU test-suite.  It links the
// product's bitstream parser (mobi_parse.cpp) and executes the command list it emits with the same
// per-pixel arithmetic the kernels use (mobi_recon_math.h) and the same scheduling contract:
//   * planes are NOT cleared between frames (ring slots are reused like the HBM ring),
//   * all inter macroblocks first, then intra macroblocks level by level, each level in REVERSE
//     raster order (the GPU gives no order inside a launch),
//   * "not yet decoded" neighbours are masked by macroblock ownership, not by memory content.
// Comparing its planes with the oracle's on the CPU validates parser + command format + the
// reordering/availability rules before any GPU time is spent.  libmobiclip_hip.so never links this.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/mobiclip_hip.h"
#include "../../mobiclipdecoder_amd/csrc/mobi_cmd.h"
#include "../../mobiclipdecoder_amd/csrc/mobi_parse.h"
#include "../../mobiclipdecoder_amd/csrc/mobi_recon_math.h"

namespace {

enum { TP = 32 }; // tile pitch; interior column c lives at byte 4 + c, halo column -1 at byte 3

struct Interp {
  MobiStreamParser parser;
  MobiGeom g;
  std::vector<uint8_t> slot[6]; // each: Y (S*H) followed by UV (S*H/2)
  int ring[6];
  int fault = 0;
  ParsedFrame pf;
  Interp(uint32_t w, uint32_t h, int ver) : parser(w, h, ver), g(parser.geom()) {
    for (int i = 0; i < 6; i++) { slot[i].assign((size_t)g.stride * g.height * 3 / 2, 0); ring[i] = i; }
  }
  uint8_t *Y(int r) { return slot[ring[r]].data(); }
  uint8_t *UV(int r) { return slot[ring[r]].data() + (size_t)g.stride * g.height; }
};

inline uint32_t ld4(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline void st4(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }

// ---- residual: dequantise into the coefficient tile, inverse transform, add ---------------------
void dequant_into(int q, const uint32_t *coefs, int n, uint32_t t8mask, int coef_tile[6 * 64]) {
  memset(coef_tile, 0, sizeof(int) * 6 * 64);
  int32_t sc[MOBI_SCALE_STRIDE];
  mobi_build_scale_table(q, sc);
  for (int i = 0; i < n; i++) {
    uint32_t e = coefs[i];
    int t = e & 0x1FF, level = (int32_t)e >> 16, area = t >> 6, p = t & 63;
    int scale = ((t8mask >> area) & 1) ? sc[p] : sc[64 + (p & 15)];
    coef_tile[t] = scale * level;
  }
}
// area = one 8x8 region; `tile` points at its top-left sample, pitch TP.  sub_mask: which 4x4s to run.
void resid_area(const int *c, bool is8, int sub_mask, uint8_t *tile, int pitch, int *fault) {
  int tmp[64], in[8], out[8];
  if (is8) {
    for (int k = 0; k < 8; k++) { // "lane" k: pass 1 over coefficient group k
      for (int m = 0; m < 8; m++) in[m] = c[8 * k + m];
      if (k == 0) in[0] += 32;
      mobi_bfly8(in, out);
      for (int m = 0; m < 8; m++) tmp[8 * m + k] = out[m];
    }
    for (int i = 0; i < 8; i++) { // "lane" i: pass 2 -> output row i
      mobi_bfly8(&tmp[8 * i], out);
      for (int j = 0; j < 8; j++) tile[i * pitch + j] = (uint8_t)mobi_add_clamp(tile[i * pitch + j], out[j] >> 6, fault);
    }
  } else {
    for (int s = 0; s < 4; s++) {
      if (!((sub_mask >> s) & 1)) continue;
      const int *cs = c + 16 * s;
      for (int k = 0; k < 4; k++) {
        for (int m = 0; m < 4; m++) in[m] = cs[4 * k + m];
        if (k == 0) in[0] += 32;
        mobi_bfly4(in, out);
        for (int m = 0; m < 4; m++) tmp[4 * m + k] = out[m];
      }
      uint8_t *t = tile + (s >> 1) * 4 * pitch + (s & 1) * 4;
      for (int i = 0; i < 4; i++) {
        mobi_bfly4(&tmp[4 * i], out);
        for (int j = 0; j < 4; j++) t[i * pitch + j] = (uint8_t)mobi_add_clamp(t[i * pitch + j], out[j] >> 6, fault);
      }
    }
  }
}

// ---- inter macroblock ----------------------------------------------------------------------------
void exec_inter(Interp &I, int mb, const MbDesc &d) {
  const MobiGeom &g = I.g;
  const long S = g.stride;
  const uint32_t *pl = I.pf.payload.data() + d.payload_off;
  const int nl = (d.w1 >> 1) & 0x7F, cbp6 = (d.w1 >> 8) & 0x3F, t8 = (d.w1 >> 14) & 0x3F, ncoef = d.w2 & 0x3FF;
  const long off = (long)(mb / g.mbw) * 16 * S + (mb % g.mbw) * 16;
  uint8_t ty[16 * TP], tc[2][8 * TP]; // prediction tiles (interior only; pitch TP, origin at byte 0)
  // per pixel group: the leaf record of the descriptor (single / two halves) or the MV cell under it (deeper trees).
  // A source = (reference slot, position of the macroblock origin's image, CopyBlock phase), for luma and chroma.
  const int dual = (d.w1 >> 26) & 3; // two inline halves: no cell map in the payload
  struct Src { int ref; long ypos, cpos; int yph, cph; };
  auto src_at = [&](int cellx, int celly) {
    Src s;
    if (nl == 1 || dual) {
      const int i = dual && (dual == MOBI_DUAL_TB ? celly >= 4 : cellx >= 4) ? 1 : 0;
      s.ref = (d.w2 >> (10 + 3 * i)) & 7;
      s.ypos = (int32_t)(i ? d.w5 : d.w3);
      s.cpos = (int32_t)(i ? d.w6 : d.w4);
      s.yph = (d.w2 >> (16 + 4 * i)) & 3;
      s.cph = (d.w2 >> (18 + 4 * i)) & 3;
    } else {
      const uint32_t c = pl[celly * 8 + cellx];
      const int dx = mobi_cell_dx(c), dy = mobi_cell_dy(c), cdx = dx >> 1, cdy = dy >> 1;
      s.ref = mobi_cell_ref(c);
      s.ypos = off + (long)(dy >> 1) * S + (dx >> 1);
      s.cpos = off / 2 + (long)(cdy >> 1) * S + (cdx >> 1);
      s.yph = (dx & 1) | ((dy & 1) << 1);
      s.cph = (cdx & 1) | ((cdy & 1) << 1);
    }
    return s;
  };
  auto fetch5 = [&](const uint8_t *plane, long len, long pos, uint8_t *a, uint8_t *b) {
    for (int k = 0; k < 5; k++) { // the 4-px word may stick out of a leaf's validated window: guard like the HBM slack
      long p0 = pos + k, p1 = pos + S + k;
      a[k] = (p0 >= 0 && p0 < len) ? plane[p0] : 0;
      b[k] = (p1 >= 0 && p1 < len) ? plane[p1] : 0;
    }
  };
  for (int lane = 0; lane < 64; lane++) { // luma: lane -> row lane>>2, 4 px at (lane&3)*4 = two cells
    int row = lane >> 2, c4 = (lane & 3) * 4;
    for (int half = 0; half < 2; half++) {
      const Src s = src_at(c4 / 2 + half, row >> 1);
      uint8_t a[8] = {0}, b[8] = {0};
      fetch5(I.Y(s.ref), S * g.height, s.ypos + (long)row * S + c4, a, b);
      uint32_t v = mobi_mc4(ld4(a), ld4(a + 1), ld4(b), ld4(b + 1), s.yph);
      for (int k = 2 * half; k < 2 * half + 2; k++) ty[row * TP + c4 + k] = (uint8_t)(v >> (8 * k));
    }
  }
  for (int lane = 0; lane < 32; lane++) { // chroma: lanes 0..15 U, 16..31 V; one cell per chroma sample
    int v01 = lane >> 4, row = (lane & 15) >> 1, c4 = (lane & 1) * 4;
    for (int k = 0; k < 4; k++) {
      const Src s = src_at(c4 + k, row);
      uint8_t a[8] = {0}, b[8] = {0};
      fetch5(I.UV(s.ref), S * g.height / 2, s.cpos + v01 * (S / 2) + (long)row * S + c4, a, b);
      uint32_t v = mobi_mc4(ld4(a), ld4(a + 1), ld4(b), ld4(b + 1), s.cph);
      tc[v01][row * TP + c4 + k] = (uint8_t)(v >> (8 * k));
    }
  }
  const uint32_t *cw = pl + (nl > 1 && !dual ? MOBI_MV_CELLS : 0);
  if (cbp6) {
    int coef[6 * 64];
    dequant_into((d.w1 >> 20) & 63, cw, ncoef, t8, coef);
    for (int a = 0; a < 6; a++) {
      if (!((cbp6 >> a) & 1)) continue;
      uint8_t *t = a < 4 ? ty + (a >> 1) * 8 * TP + (a & 1) * 8 : tc[a - 4];
      resid_area(coef + 64 * a, (t8 >> a) & 1, 0xF, t, TP, &I.fault);
    }
  }
  uint8_t *y0 = I.Y(0), *uv0 = I.UV(0);
  for (int r = 0; r < 16; r++) memcpy(y0 + off + r * S, ty + r * TP, 16);
  for (int v = 0; v < 2; v++)
    for (int r = 0; r < 8; r++) memcpy(uv0 + off / 2 + v * (S / 2) + r * S, tc[v] + r * TP, 8);
}

// ---- intra macroblock ----------------------------------------------------------------------------
struct TileNb { // neighbour accessor on a tile: block origin at (by, bx) inside the tile interior
  const uint8_t *t;
  int by, bx;
  int operator()(int dy, int dx) const { return t[(by + dy + 1) * TP + 4 + bx + dx]; }
};
// one block record: predict (+ residual) on the tile.  n = 8 or 4, (by,bx) inside the area-local tile.
void run_block(Interp &I, uint8_t *tile, int by, int bx, int n, int mode, int param, bool coded, const int *coef,
               bool is8, int sub, long block_off, bool is_uv) {
  const long S = I.g.stride;
  TileNb nb{tile, by, bx};
  if (mode == 2) {
    uint32_t words[64];
    for (int y = 0; y < n; y++)
      for (int x0 = 0; x0 < n; x0 += 4) words[y * (n / 4) + x0 / 4] = mobi_plane_word(n, param, y, x0, nb);
    for (int y = 0; y < n; y++)
      for (int x0 = 0; x0 < n; x0 += 4) st4(tile + (by + y + 1) * TP + 4 + bx + x0, words[y * (n / 4) + x0 / 4]);
  } else if (mode != 9) {
    int vfix = is_uv && (block_off % S) >= S / 2; // MD.cs:1886
    int left_avail = ((block_off - (vfix ? S / 2 : 0)) % S) != 0, top_avail = block_off >= S; // :1923-1924
    uint8_t px[64];
    for (int y = 0; y < n; y++)
      for (int x = 0; x < n; x++) px[y * n + x] = (uint8_t)mobi_pred_px(mode, n, y, x, top_avail, left_avail, nb);
    for (int y = 0; y < n; y++)
      for (int x = 0; x < n; x++) tile[(by + y + 1) * TP + 4 + bx + x] = px[y * n + x];
  }
  if (coded) {
    uint8_t *t = tile + (by + 1) * TP + 4 + bx;
    if (is8) resid_area(coef, true, 0, t, TP, &I.fault);
    else resid_area(coef, false, 1 << sub, t - ((sub >> 1) * 4 * TP + (sub & 1) * 4), TP, &I.fault);
  }
}
void exec_intra(Interp &I, int mb, const MbDesc &d) {
  const MobiGeom &g = I.g;
  const long S = g.stride;
  const uint32_t *rec = I.pf.payload.data() + d.payload_off;
  const int t8 = (d.w1 >> 14) & 0x3F, ncoef = d.w2 & 0x3FF;
  const long off = (long)(mb / g.mbw) * 16 * S + (mb % g.mbw) * 16;
  uint8_t *y0 = I.Y(0), *uv0 = I.UV(0);
  uint8_t ty[17 * TP], tc[2][9 * TP];
  memset(ty, 0, sizeof(ty));
  memset(tc, 0, sizeof(tc));
  // halo: neighbours owned by raster-earlier MBs are real, everything else reads as 0 (fresh plane, MD.cs:107)
  auto luma = [&](long a) -> uint8_t { int o = g.owner_luma(a); return (o >= 0 && o < mb) ? y0[a] : 0; }; // padding is always 0
  for (int c = -1; c <= MOBI_HALO_Y_RIGHT; c++) ty[0 * TP + 4 + c] = luma(off - S + c);
  for (int r = 0; r < 16; r++) ty[(r + 1) * TP + 3] = luma(off + r * S - 1);
  // right of the MB: normally a later MB or padding (-> 0), but when Stride == Width the address wraps
  // into the first MB of this row, which IS already decoded (linear addressing, SURVEY hard part 3)
  for (int r = 0; r < 16; r++)
    for (int c = 16; c <= MOBI_HALO_Y_RIGHT; c++) ty[(r + 1) * TP + 4 + c] = luma(off + r * S + c);
  for (int v = 0; v < 2; v++) {
    long base = off / 2 + v * (S / 2);
    auto chroma = [&](long a) -> uint8_t { int o = g.owner_chroma(a); return (o >= 0 && o < mb) ? uv0[a] : 0; };
    for (int c = -1; c <= MOBI_HALO_C_RIGHT; c++) tc[v][4 + c] = chroma(base - S + c);
    for (int r = 0; r < 8; r++) tc[v][(r + 1) * TP + 3] = chroma(base + r * S - 1);
    for (int r = 0; r < 8; r++)
      for (int c = 8; c <= MOBI_HALO_C_RIGHT; c++) tc[v][(r + 1) * TP + 4 + c] = chroma(base + r * S + c);
  }
  int coef[6 * 64];
  dequant_into((d.w1 >> 20) & 63, rec + MOBI_INTRA_RECORDS, ncoef, t8, coef);
  const int32_t *wide = (const int32_t *)rec + MOBI_INTRA_RECORDS + ncoef; // parameters that do not fit a record's 16 bits (mobi_cmd.h)
  auto param_of = [&](uint32_t r, int idx) { return (r & MOBI_REC_WIDE) ? wide[idx] : (int)(int16_t)(r >> 16); };
  if (d.w3 & 1) run_block(I, ty, 0, 0, 16, 2, (d.w3 & MOBI_W3_WIDE) ? wide[24] : (int16_t)(d.w3 >> 16), false, nullptr, false, 0, off, false);
  for (int a = 0; a < 6; a++) {
    uint8_t *tile = a < 4 ? ty : tc[a - 4];
    int ay = a < 4 ? (a >> 1) * 8 : 0, ax = a < 4 ? (a & 1) * 8 : 0;
    long aoff = a < 4 ? off + (long)ay * S + ax : off / 2 + (a - 4) * (S / 2);
    uint32_t r0 = rec[a * 4];
    if ((r0 >> 6) & 1) run_block(I, tile, ay, ax, 8, 2, param_of(r0, a * 4), false, nullptr, false, 0, aoff, a >= 4);
    if (!((r0 >> 5) & 1)) {
      run_block(I, tile, ay, ax, 8, r0 & 15, ((r0 >> 6) & 1) ? 0 : param_of(r0, a * 4), (r0 >> 4) & 1, coef + 64 * a, true, 0, aoff, a >= 4);
    } else {
      for (int s = 0; s < 4; s++) {
        uint32_t r = rec[a * 4 + s];
        int sy = (s >> 1) * 4, sx = (s & 1) * 4;
        int param = (s == 0 && ((r >> 6) & 1)) ? 0 : param_of(r, a * 4 + s);
        run_block(I, tile, ay + sy, ax + sx, 4, r & 15, param, (r >> 4) & 1, coef + 64 * a, false, s, aoff + (long)sy * S + sx, a >= 4);
      }
    }
  }
  for (int r = 0; r < 16; r++) memcpy(y0 + off + r * S, ty + (r + 1) * TP + 4, 16);
  for (int v = 0; v < 2; v++)
    for (int r = 0; r < 8; r++) memcpy(uv0 + off / 2 + v * (S / 2) + r * S, tc[v] + (r + 1) * TP + 4, 8);
}

} // namespace

#include "../../mobiclipdecoder_amd/csrc/mobi_tile.h"
#include "../../mobiclipdecoder_amd/csrc/mobi_dparse_tables.h"
static uint8_t g_blob[MOBI_DT_BYTES];
static const uint8_t *blob() { static bool once = (mobi_dparse_build_tables(MOBI_VERSION_MOFLEX3DS, g_blob), true); (void)once; return g_blob; }
static int mobi_test_zz8(int i) { return blob()[MOBI_DT_ZZ8 + i]; }
static int mobi_test_zz4(int i) { return blob()[MOBI_DT_ZZ4 + i]; }
extern "C" {
// the private plane layout's address map (mobi_tile.h), for tests/test_tile_layout.py
uint32_t mobi_test_ty(uint32_t a, int lgS) { return mobi_ty(a, lgS); }
uint32_t mobi_test_tc(uint32_t a, int lgS) { return mobi_tc(a, lgS); }
// how often the parser refused a stream (MOBI_E_UNSUPPORTED) in this process, by cause (mobi_parse.h: MOBI_REFUSE_*); tools/exp_refusals.py
unsigned long mobi_cmdinterp_scratch_reads(void) { return mobi_scratch_read_count; } // Internal[154..217] read by a walk (r05)
unsigned long mobi_cmdinterp_literal_frames(void) { return mobi_literal_frame_count; } // frames the host parser shipped as literal values
void mobi_cmdinterp_refusals(unsigned long out[4]) { for (int i = 0; i < MOBI_REFUSE_CLASSES; i++) out[i] = mobi_refusal_count[i]; }
void *mobi_cmdinterp_create(uint32_t w, uint32_t h, int version) {
  if ((w & 15) || (h & 15) || w == 0 || h == 0 || w > 1024) return nullptr;
  return new Interp(w, h, version);
}
void mobi_cmdinterp_destroy(void *p) { delete (Interp *)p; }
int mobi_cmdinterp_decode(void *p, const uint8_t *data, size_t len, int32_t *offset) {
  Interp &I = *(Interp *)p;
  int rc = I.parser.parse_frame(data, len, offset, I.pf);
  if (rc == MOBI_E_VERSION) return rc;
  int last = I.ring[5];
  for (int i = 5; i > 0; i--) I.ring[i] = I.ring[i - 1];
  I.ring[0] = last;
  if (rc != MOBI_OK) return rc;
  I.fault = 0;
  const int n = (int)I.pf.desc.size();
  for (int mb = n - 1; mb >= 0; mb--)
    if ((I.pf.desc[mb].w1 & 1) == MOBI_MB_INTER) exec_inter(I, mb, I.pf.desc[mb]);
  for (uint32_t L = 1; L <= I.pf.hdr.n_levels; L++)
    for (int i = (int)I.pf.level_start[L + 1] - 1; i >= (int)I.pf.level_start[L]; i--) {
      int mb = (int)I.pf.intra_mbs[i];
      exec_intra(I, mb, I.pf.desc[mb]);
    }
  return I.fault ? MOBI_E_CLAMP : MOBI_OK;
}
const uint8_t *mobi_cmdinterp_y(void *p, int idx) { return ((Interp *)p)->Y(idx); }
const uint8_t *mobi_cmdinterp_uv(void *p, int idx) { return ((Interp *)p)->UV(idx); }
int mobi_cmdinterp_stride(void *p) { return ((Interp *)p)->g.stride; }
uint32_t mobi_cmdinterp_quantizer(void *p) { return ((Interp *)p)->parser.quantizer(); }
uint32_t mobi_cmdinterp_cmd_bytes(void *p) { return ((Interp *)p)->pf.hdr.cmd_bytes; }
uint32_t mobi_cmdinterp_levels(void *p) { return ((Interp *)p)->pf.hdr.n_levels; }
// the last frame's command list as the parser left it (tests/test_launch_items.py): descriptors (8 words per macroblock), the intra
// macroblocks in launch order with the start of every level, and the launch items the parser wrote for them (4 words each)
uint32_t mobi_cmdinterp_n_mbs(void *p) { return (uint32_t)((Interp *)p)->pf.desc.size(); }
const uint32_t *mobi_cmdinterp_desc(void *p) { return (const uint32_t *)((Interp *)p)->pf.desc.data(); }
uint32_t mobi_cmdinterp_n_intra(void *p) { return (uint32_t)((Interp *)p)->pf.intra_mbs.size(); }
const uint32_t *mobi_cmdinterp_intra_mbs(void *p) { return ((Interp *)p)->pf.intra_mbs.data(); }
const uint32_t *mobi_cmdinterp_level_start(void *p) { return ((Interp *)p)->pf.level_start.data(); }
const uint32_t *mobi_cmdinterp_intra_items(void *p) { return ((Interp *)p)->pf.intra_items.data(); }
uint32_t mobi_cmdinterp_payload_words(void *p) { return (uint32_t)((Interp *)p)->pf.payload.size(); }
// r05, tests/test_parse_fallback.py: the decoder state that survives a frame (mobi_state.h).  Internal[idx] as the parser holds it; its
// export / import in the form the device parsers keep; and what mobi_parse_tail would rebuild from the last frame's command list.
uint32_t mobi_cmdinterp_internal(void *p, uint32_t idx) { return ((Interp *)p)->parser.internal_word(idx); }
void mobi_cmdinterp_export_state(void *p, MobiDevState *st, MobiDevTail *tail) { ((Interp *)p)->parser.export_state(*st, *tail); }
void mobi_cmdinterp_import_state(void *p, const MobiDevState *st, const MobiDevTail *tail, int ring_frames) {
  Interp &I = *(Interp *)p;
  I.parser.import_state(*st, *tail);
  (void)ring_frames;
}
// copy the ring (planes) of another interpreter: the host parser takes a clip over with the pictures the device decoded so far
void mobi_cmdinterp_copy_ring(void *dst, void *src) {
  Interp &D = *(Interp *)dst, &S = *(Interp *)src;
  for (int i = 0; i < 6; i++) { D.slot[i] = S.slot[i]; D.ring[i] = S.ring[i]; }
}
void mobi_cmdinterp_tail(void *p, const MobiDevTail *in, MobiDevTail *out) {
  Interp &I = *(Interp *)p;
  uint8_t izz8[64], izz4[16];
  for (int i = 0; i < 64; i++) izz8[mobi_test_zz8(i)] = (uint8_t)i;
  for (int i = 0; i < 16; i++) izz4[mobi_test_zz4(i)] = (uint8_t)i;
  MobiTailScan sc;
  mobi_tail_scan_init(sc);
  const ParsedFrame &f = I.pf;
  for (int mb = (int)f.desc.size() - 1; mb >= 0 && !sc.done; mb--) {
    const MbDesc &d = f.desc[mb];
    const int n = (int)(d.w2 & 0x3FF);
    if (!n) continue;
    const bool intra = (d.w1 & 1) == MOBI_MB_INTRA;
    const uint32_t nl = (d.w1 >> 1) & 0x7F, dual = (d.w1 >> 26) & 3;
    const uint32_t woff = d.payload_off + (intra ? MOBI_INTRA_RECORDS : (nl > 1 && !dual) ? MOBI_MV_CELLS : 0);
    mobi_tail_scan_mb(sc, f.payload.data() + woff, n, woff, (d.w1 >> 14) & 0x3F, izz8, izz4);
  }
  int32_t scale[MOBI_SCALE_STRIDE];
  mobi_build_scale_table((int)(f.desc.empty() ? 0 : (f.desc[0].w1 >> 20) & 63), scale);
  *out = *in;
  mobi_tail_finish(sc, f.payload.data(), scale, *in, *out);
}
int mobi_cmdinterp_tables(int version, uint8_t *out) { mobi_dparse_build_tables(version, out); return MOBI_DT_BYTES; } // (the device parsers' table blob)
const uint32_t *mobi_cmdinterp_payload(void *p) { return ((Interp *)p)->pf.payload.data(); }
}
