// mobi_lsparse_host.cpp -- TEST TOOL: the lock-step parser's per-lane functions (mobi_lsparse.h) run on the CPU, one clip at a time, against
// the host parser (mobi_parse.cpp) on the same frames: descriptors, payload, intra list, consumed bytes and persistent state must be equal
// whenever the lock-step parser does not bail out, and it must bail out whenever the host parser reports anything but MOBI_OK.
// Built by mobiclipdecoder_amd/build.py into tests/tools/libmobi_lsparse_host.so; used by tests/test_lsparse.py.  Not part of the product.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "mobi_gop.h"
#include "mobi_lsparse.h"
#include "mobi_parse.h"

namespace {
struct HostStore {
  uint32_t mvp_[64 + 2];
  uint32_t rec_[MOBI_INTRA_RECORDS]; // (the partition-tree stack shares these words, as on the device: mobi_lsparse.hip)
  uint8_t mc_[40];
  const uint8_t *data;
  uint32_t len2; // bytes that exist as whole 16-bit words
  uint32_t mvp_load(int i) const { return mvp_[i]; } // (on the device: the clip's MobiDevTail.mvc in HBM)
  void mvp_store(int i, uint32_t v) { mvp_[i] = v; }
  uint32_t &stk(int i) { return rec_[i]; }
  uint32_t &rec(int i) { return rec_[i]; }
  uint8_t &mc(int i) { return mc_[i]; }
  uint32_t ring32(uint32_t rd) const {
    uint32_t v = 0;
    for (int k = 0; k < 4; k++)
      if (rd + k < len2) v |= (uint32_t)data[rd + k] << (8 * k);
    return v;
  }
};
struct Clip {
  int w, h, version;
  MobiGeom g;
  std::vector<uint8_t> tables;
  HostStore m;
  LsLane s;
  uint32_t quant = 0, yuvfmt = 0, tables_set = 0;
  int frames_started = 0, predx = 0, predy = 0;
  std::vector<MbDesc> desc;
  std::vector<uint32_t> pay, items;
  long rounds = 0, rounds_by_state[16] = {0};
};
} // namespace

extern "C" {
void *mobi_lshost_create(uint32_t w, uint32_t h, int version) {
  if ((w & 15) || (h & 15) || w == 0 || h == 0 || w > 1024) return nullptr;
  Clip *c = new Clip();
  c->w = (int)w; c->h = (int)h; c->version = version;
  MobiStreamParser p(w, h, version);
  c->g = p.geom();
  c->tables.resize(MOBI_DT_BYTES);
  mobi_dparse_build_tables(version, c->tables.data());
  for (int i = 0; i < 1024; i++) ls_prepare_tables(c->tables.data(), i); // (this copy is the lock-step parser's own, as on the device)
  memset(&c->m, 0, sizeof(c->m));
  const int n = c->g.mbw * c->g.mbh;
  c->desc.resize(n);
  c->items.resize(n);
  c->pay.resize((size_t)n * 448 + 64);
  return c;
}
void mobi_lshost_destroy(void *p) { delete (Clip *)p; }
// One frame.  Returns the bail code (0 = parsed); *consumed, *n_intra, *pay_words as MobiDevResult would carry them.
int mobi_lshost_parse(void *p, const uint8_t *data, size_t len, int32_t *consumed, uint32_t *n_intra, uint32_t *pay_words, uint32_t *frame_type) {
  Clip &C = *(Clip *)p;
  LsCtx c;
  c.T = C.tables.data();
  c.width = C.w; c.height = C.h; c.stride = C.g.stride; c.lg = C.g.lg; c.mbw = C.g.mbw; c.mbh = C.g.mbh; c.n_mbs = C.g.mbw * C.g.mbh;
  c.version = C.version;
  c.pay_cap = (uint32_t)C.pay.size();
  // work on copies of what survives the frame: a bail-out must leave it as it was
  HostStore m = C.m;
  m.data = data;
  m.len2 = (uint32_t)len & ~1u;
  LsLane s;
  memset(&s, 0, sizeof(s));
  s.quant = C.quant; s.yuvfmt = C.yuvfmt; s.tables_set = C.tables_set; s.frames_started = C.frames_started + 1;
  s.predx = C.predx; s.predy = C.predy;
  s.desc = C.desc.data(); s.pay = C.pay.data(); s.pay_base = 0; s.clip = 0; s.items = C.items.data();
  ls_begin_frame(s, m, c, (uint32_t)len);
  while (s.st != LS_DONE) {
    C.rounds++;
    C.rounds_by_state[s.st & 15]++;
    ls_round(s, m, c); // as the kernel does
  }
  if (!s.bail) {
    const int used = ls_consumed(s.cbits, (uint32_t)len);
    if (used < 0) s.bail = 16;
    else *consumed = used;
  }
  if (s.bail) return s.bail;
  const LsGeom g{C.w, C.h, C.g.stride, C.g.lg, C.g.mbw};
  for (uint32_t i = 0; i < s.n_items; i++)
    if (!ls_intra_deps(g, C.desc.data(), (int)(C.items[i] & 0x1FFF))) return 17;
  C.m = m;
  C.quant = s.quant; C.yuvfmt = s.yuvfmt; C.tables_set = s.tables_set; C.frames_started = s.frames_started;
  C.predx = s.predx; C.predy = s.predy;
  *n_intra = s.n_items;
  *pay_words = s.pay_pos;
  *frame_type = (uint32_t)s.iframe;
  return 0;
}
const MbDesc *mobi_lshost_desc(void *p) { return ((Clip *)p)->desc.data(); }
const uint32_t *mobi_lshost_payload(void *p) { return ((Clip *)p)->pay.data(); }
const uint32_t *mobi_lshost_items(void *p) { return ((Clip *)p)->items.data(); }
uint32_t mobi_lshost_quant(void *p) { return ((Clip *)p)->quant; }
long mobi_lshost_rounds(void *p, long by_state[16]) {
  Clip &C = *(Clip *)p;
  if (by_state) for (int i = 0; i < 16; i++) by_state[i] = C.rounds_by_state[i];
  return C.rounds;
}

// Scheduling experiment (tools/exp_lssched.py): 64 clips as the 64 lanes of one wave, frame f of each, under a schedule of the three parts of
// the walk: every round ls_step_main; every `period`-th round `burst` times (ls_step_intra + `kb` cheap rounds); then `k` cheap rounds.
// counts[0..3] = rounds, and how many times each part ran with at least one lane in it (main, intra, cheap).  Returns 0, or a bail code.
int mobi_lshost_wave_sim(void *const *clips, const uint8_t *const *data, const size_t *len, int n_lanes, int k, int period, int burst, int kb, long counts[4]) {
  struct Lane { Clip *C; HostStore m; LsLane s; LsCtx c; };
  std::vector<Lane> L(n_lanes);
  for (int i = 0; i < n_lanes; i++) {
    Clip &C = *(Clip *)clips[i];
    Lane &l = L[i];
    l.C = &C;
    l.c.T = C.tables.data();
    l.c.width = C.w; l.c.height = C.h; l.c.stride = C.g.stride; l.c.lg = C.g.lg; l.c.mbw = C.g.mbw; l.c.mbh = C.g.mbh; l.c.n_mbs = C.g.mbw * C.g.mbh;
    l.c.version = C.version;
    l.c.pay_cap = (uint32_t)C.pay.size();
    l.m = C.m;
    l.m.data = data[i];
    l.m.len2 = (uint32_t)len[i] & ~1u;
    memset(&l.s, 0, sizeof(l.s));
    l.s.quant = C.quant; l.s.yuvfmt = C.yuvfmt; l.s.tables_set = C.tables_set; l.s.frames_started = C.frames_started + 1;
    l.s.desc = C.desc.data(); l.s.pay = C.pay.data(); l.s.pay_base = 0; l.s.clip = 0; l.s.items = C.items.data();
    ls_begin_frame(l.s, l.m, l.c, (uint32_t)len[i]);
  }
  counts[0] = counts[1] = counts[2] = counts[3] = 0;
  auto any = [&](auto pred) { for (auto &l : L) if (pred(l.s)) return true; return false; };
  auto cheap = [&](int n) {
    for (int j = 0; j < n; j++) {
      if (!any([](const LsLane &s) { return s.st == LS_NEXT || s.st == LS_TOKEN; })) break;
      counts[3]++;
      for (auto &l : L) { ls_next_fast(l.s, l.m, l.c); ls_token_fast(l.s, l.m, l.c); }
    }
  };
  for (long round = 0; any([](const LsLane &s) { return s.st != LS_DONE; }); round++) {
    counts[0]++;
    if (any([](const LsLane &s) { return s.st == LS_MB_END || s.st == LS_MB_BEGIN || s.st == LS_NODE || s.st == LS_P_CBP || s.st == LS_NEXT_SLOW || s.st == LS_TOKEN_SLOW; })) counts[1]++;
    for (auto &l : L) ls_step_main(l.s, l.m, l.c);
    if (round % period == 0)
      for (int b = 0; b < burst; b++) {
        if (!any([](const LsLane &s) { return ls_in_intra(s); })) break;
        counts[2]++;
        for (auto &l : L) ls_step_intra(l.s, l.m, l.c);
        cheap(kb);
      }
    cheap(k);
  }
  for (auto &l : L) {
    if (l.s.bail) return l.s.bail;
    l.C->m = l.m;
    l.C->quant = l.s.quant; l.C->yuvfmt = l.s.yuvfmt; l.C->tables_set = l.s.tables_set; l.C->frames_started = l.s.frames_started;
  }
  return 0;
}

// The same for any order of the parts within a round: sched = a string of M (ls_step_main), I (ls_step_intra), N (ls_next_fast), T (ls_token_fast).
// counts[0] = rounds, counts[1..4] = how often M, I, N, T ran with at least one lane in them.
int mobi_lshost_wave_sched(void *const *clips, const uint8_t *const *data, const size_t *len, int n_lanes, const char *sched, long counts[5]) {
  struct Lane { Clip *C; HostStore m; LsLane s; LsCtx c; };
  std::vector<Lane> L(n_lanes);
  for (int i = 0; i < n_lanes; i++) {
    Clip &C = *(Clip *)clips[i];
    Lane &l = L[i];
    l.C = &C;
    l.c.T = C.tables.data();
    l.c.width = C.w; l.c.height = C.h; l.c.stride = C.g.stride; l.c.lg = C.g.lg; l.c.mbw = C.g.mbw; l.c.mbh = C.g.mbh; l.c.n_mbs = C.g.mbw * C.g.mbh;
    l.c.version = C.version;
    l.c.pay_cap = (uint32_t)C.pay.size();
    l.m = C.m;
    l.m.data = data[i];
    l.m.len2 = (uint32_t)len[i] & ~1u;
    memset(&l.s, 0, sizeof(l.s));
    l.s.quant = C.quant; l.s.yuvfmt = C.yuvfmt; l.s.tables_set = C.tables_set; l.s.frames_started = C.frames_started + 1;
    l.s.desc = C.desc.data(); l.s.pay = C.pay.data(); l.s.pay_base = 0; l.s.clip = 0; l.s.items = C.items.data();
    ls_begin_frame(l.s, l.m, l.c, (uint32_t)len[i]);
  }
  for (int k = 0; k < 5; k++) counts[k] = 0;
  auto any = [&](auto pred) { for (auto &l : L) if (pred(l.s)) return true; return false; };
  while (any([](const LsLane &s) { return s.st != LS_DONE; })) {
    counts[0]++;
    for (const char *p = sched; *p; p++) {
      if (*p == 'M') {
        if (any([](const LsLane &s) { return s.st == LS_MB_END || s.st == LS_MB_BEGIN || s.st == LS_NODE || s.st == LS_P_CBP || s.st == LS_NEXT_SLOW || s.st == LS_TOKEN_SLOW; })) counts[1]++;
        for (auto &l : L) ls_step_main(l.s, l.m, l.c);
      } else if (*p == 'I') {
        if (any([](const LsLane &s) { return ls_in_intra(s); })) counts[2]++;
        for (auto &l : L) ls_step_intra(l.s, l.m, l.c);
      } else if (*p == 'N') {
        if (any([](const LsLane &s) { return s.st == LS_NEXT; })) counts[3]++;
        for (auto &l : L) ls_next_fast(l.s, l.m, l.c);
      } else if (*p == 'T') {
        if (any([](const LsLane &s) { return s.st == LS_TOKEN; })) counts[4]++;
        for (auto &l : L) ls_token_fast(l.s, l.m, l.c);
      }
    }
  }
  for (auto &l : L) {
    if (l.s.bail) return l.s.bail;
    l.C->m = l.m;
    l.C->quant = l.s.quant; l.C->yuvfmt = l.s.yuvfmt; l.C->tables_set = l.s.tables_set; l.C->frames_started = l.s.frames_started;
  }
  return 0;
}

// Population-driven schedule: every iteration runs the ONE part whose (lanes waiting for it) / (its cost) is largest.  cost[4] = M, I, N, T
// in instructions.  counts[0] = iterations, counts[1..4] = runs of M, I, N, T.
int mobi_lshost_wave_greedy(void *const *clips, const uint8_t *const *data, const size_t *len, int n_lanes, const double cost[4], long counts[5]) {
  struct Lane { Clip *C; HostStore m; LsLane s; LsCtx c; };
  std::vector<Lane> L(n_lanes);
  for (int i = 0; i < n_lanes; i++) {
    Clip &C = *(Clip *)clips[i];
    Lane &l = L[i];
    l.C = &C;
    l.c.T = C.tables.data();
    l.c.width = C.w; l.c.height = C.h; l.c.stride = C.g.stride; l.c.lg = C.g.lg; l.c.mbw = C.g.mbw; l.c.mbh = C.g.mbh; l.c.n_mbs = C.g.mbw * C.g.mbh;
    l.c.version = C.version;
    l.c.pay_cap = (uint32_t)C.pay.size();
    l.m = C.m;
    l.m.data = data[i];
    l.m.len2 = (uint32_t)len[i] & ~1u;
    memset(&l.s, 0, sizeof(l.s));
    l.s.quant = C.quant; l.s.yuvfmt = C.yuvfmt; l.s.tables_set = C.tables_set; l.s.frames_started = C.frames_started + 1;
    l.s.desc = C.desc.data(); l.s.pay = C.pay.data(); l.s.pay_base = 0; l.s.clip = 0; l.s.items = C.items.data();
    ls_begin_frame(l.s, l.m, l.c, (uint32_t)len[i]);
  }
  for (int k = 0; k < 5; k++) counts[k] = 0;
  for (;;) {
    int n[4] = {0, 0, 0, 0}, live = 0;
    for (auto &l : L) {
      const int st = l.s.st;
      if (st == LS_DONE) continue;
      live++;
      if (st == LS_NEXT) n[2]++;
      else if (st == LS_TOKEN) n[3]++;
      else if (ls_in_intra(l.s)) n[1]++;
      else n[0]++;
    }
    if (!live) break;
    int best = 0;
    if (cost[0] < 0) { // threshold policy: -cost[0] = share of the live lanes that must wait for M / I before they run; cheap parts otherwise
      const double th = -cost[0];
      const int slow = n[0] + n[1];
      if (slow >= th * live || n[2] + n[3] == 0) best = n[0] >= n[1] ? 0 : 1;
      else best = n[2] > n[3] ? 2 : 3;
      if (best <= 1 && cost[1] < 0) { // both slow parts in one go
        counts[0]++;
        if (n[0]) counts[1]++;
        if (n[1]) counts[2]++;
        for (auto &l : L) { ls_step_main(l.s, l.m, l.c); ls_step_intra(l.s, l.m, l.c); }
        continue;
      }
    } else
      for (int k = 1; k < 4; k++) if (n[k] / cost[k] > n[best] / cost[best]) best = k;
    counts[0]++;
    counts[1 + best]++;
    for (auto &l : L) {
      if (best == 0) ls_step_main(l.s, l.m, l.c);
      else if (best == 1) ls_step_intra(l.s, l.m, l.c);
      else if (best == 2) ls_next_fast(l.s, l.m, l.c);
      else ls_token_fast(l.s, l.m, l.c);
    }
  }
  for (auto &l : L) {
    if (l.s.bail) return l.s.bail;
    l.C->m = l.m;
    l.C->quant = l.s.quant; l.C->yuvfmt = l.s.yuvfmt; l.C->tables_set = l.s.tables_set; l.C->frames_started = l.s.frames_started;
  }
  return 0;
}

// The whole differential in one call: every frame of a clip through both parsers.  Returns the number of frames that compared equal
// (all of them: n_frames), or -(frame + 1) at the first difference (what differs goes to stderr).  *bails = frames the lock-step parser
// left to the other one (allowed only when allow_bail is set or the host parser did not return MOBI_OK).
int mobi_lshost_compare(uint32_t w, uint32_t h, int version, const uint8_t *data, const uint32_t *frame_off, int n_frames, int allow_bail, int *bails) {
  Clip *C = (Clip *)mobi_lshost_create(w, h, version);
  if (!C) return -1000000;
  MobiStreamParser hp(w, h, version);
  ParsedFrame pf;
  int ok = 0;
  *bails = 0;
  for (int f = 0; f < n_frames; f++) {
    const uint8_t *d = data + frame_off[f];
    const size_t len = frame_off[f + 1] - frame_off[f];
    int32_t off = 0;
    const int rc = hp.parse_frame(d, len, &off, pf);
    int32_t used = 0;
    uint32_t n_intra = 0, pay_words = 0, ftype = 0;
    const int bail = mobi_lshost_parse(C, d, len, &used, &n_intra, &pay_words, &ftype);
    auto adopt = [&]() { // what survives the frame, as the parser that finished it leaves it (mobi_parse_frames on the device)
      C->quant = hp.quantizer(); C->yuvfmt = hp.yuv_format(); C->frames_started = hp.frames_started(); C->tables_set = hp.quant_tables_set();
      memcpy(C->m.mc_, hp.mode_cache(), 40);
    };
    if (bail) {
      (*bails)++;
      if (rc == MOBI_OK && !allow_bail) { fprintf(stderr, "frame %d: bail-out %d on a frame the host parser accepts\n", f, bail); ok = -(f + 1); break; }
      adopt();
      ok++;
      continue;
    }
    if (rc != MOBI_OK) { fprintf(stderr, "frame %d: host parser says %d, the lock-step parser went through\n", f, rc); ok = -(f + 1); break; }
    if (C->yuvfmt != hp.yuv_format() || C->frames_started != hp.frames_started() || C->tables_set != (uint32_t)hp.quant_tables_set() ||
        memcmp(C->m.mc_, hp.mode_cache(), 40) != 0) {
      fprintf(stderr, "frame %d: persistent state differs\n", f);
      ok = -(f + 1);
      break;
    }
    bool same = used == off && pay_words == pf.payload.size() && ftype == pf.hdr.frame_type && n_intra == pf.hdr.n_intra && C->quant == hp.quantizer();
    if (same && ftype == 0 && hp.device_ready()) { // the MV row cache a P-frame leaves (r06: the lanes keep it in registers and in the tail itself)
      MobiDevState st;
      MobiDevTail tl;
      hp.export_state(st, tl);
      for (int i = 0; i < C->g.mbw + 2 && same; i++)
        if (C->m.mvp_[i] != mobi_leaf_w1(tl.mvc[2 * i], tl.mvc[2 * i + 1])) { fprintf(stderr, "frame %d: MV row cache entry %d: %08x vs host (%d, %d)\n", f, i, C->m.mvp_[i], tl.mvc[2 * i], tl.mvc[2 * i + 1]); same = false; }
    }
    if (!same) fprintf(stderr, "frame %d: consumed %d/%d payload %u/%zu type %u/%u intra %u/%u quant %u/%u\n", f, used, off, pay_words, pf.payload.size(), ftype,
                       pf.hdr.frame_type, n_intra, pf.hdr.n_intra, C->quant, hp.quantizer());
    for (size_t mb = 0; same && mb < pf.desc.size(); mb++) {
      const MbDesc &a = C->desc[mb], &b = pf.desc[mb];
      bool eq = a.payload_off == b.payload_off && a.w1 == b.w1 && a.w2 == b.w2 && a.w3 == b.w3;
      if ((a.w1 & 1) == MOBI_MB_INTER) eq = eq && a.w4 == b.w4 && a.w5 == b.w5 && a.w6 == b.w6 && a.w7 == b.w7;
      else { // the same SET of dependencies
        uint32_t da[8], db[8];
        for (int k = 0; k < 4; k++) {
          const uint32_t wa = (&a.w4)[k], wb = (&b.w4)[k];
          da[2 * k] = wa & 0xFFFF; da[2 * k + 1] = wa >> 16; db[2 * k] = wb & 0xFFFF; db[2 * k + 1] = wb >> 16;
        }
        for (int k = 0; k < 8; k++) {
          bool fa = da[k] == MOBI_DEP_NONE, fb = db[k] == MOBI_DEP_NONE;
          for (int j = 0; j < 8; j++) { fa = fa || da[k] == db[j]; fb = fb || db[k] == da[j]; }
          eq = eq && fa && fb;
        }
      }
      if (!eq) {
        fprintf(stderr, "frame %d mb %zu: desc %08x %08x %08x %08x %08x %08x %08x %08x  vs host %08x %08x %08x %08x %08x %08x %08x %08x\n", f, mb, a.payload_off, a.w1,
                a.w2, a.w3, a.w4, a.w5, a.w6, a.w7, b.payload_off, b.w1, b.w2, b.w3, b.w4, b.w5, b.w6, b.w7);
        same = false;
      }
    }
    for (size_t i = 0; same && i < pf.payload.size(); i++)
      if (C->pay[i] != pf.payload[i]) { fprintf(stderr, "frame %d: payload word %zu %08x vs host %08x\n", f, i, C->pay[i], pf.payload[i]); same = false; }
    // the intra list: the host's is sorted by level, this one in raster order -- the same set
    if (same) {
      std::vector<uint32_t> x(C->items.begin(), C->items.begin() + n_intra), y(pf.intra_mbs.begin(), pf.intra_mbs.end());
      for (auto &v : x) v &= 0x1FFF;
      std::sort(x.begin(), x.end());
      std::sort(y.begin(), y.end());
      if (x != y) { fprintf(stderr, "frame %d: intra lists differ\n", f); same = false; }
    }
    if (!same) { ok = -(f + 1); break; }
    ok++;
  }
  mobi_lshost_destroy(C);
  return ok;
}
// ---- r06: what does a frame's PARSE need from the frame before it?  (measure before building frame-parallel lanes)
// Every frame f of the clip is parsed twice: by parser A in stream order, and by a parser B that starts from A's state before the frame with
// everything the frame header does not fix POISONED -- the 16 interior bytes of the intra-mode cache (Internal bytes 9..12, 17..20, 25..28,
// 33..36: MD.cs:1840-1859, 2785-2843 only ever read a byte some block of the same macroblock wrote, or a border byte that
// SetupQuantizationTables sets, :3913-3924) = 0xFF, and the MV predictor Internal[219], [220] (set per macroblock before it is read,
// MD.cs:207-208).  stats: [0] frames compared (A's state before them is one the device parsers hold), [1] frames whose command list, rc and
// Offset are the same from the poisoned state, [2] frames whose state AFTERWARDS is A's once the bytes B did not write (still 0xFF) are taken
// from A's state before the frame, [3] frames skipped (host-only state), [4] the same comparison ([1] and [2] at once) for the lock-step
// parser's lane functions, [5] lane-function frames that bailed out (not compared).
int mobi_framedep_measure(uint32_t w, uint32_t h, int version, const uint8_t *data, const uint32_t *frame_off, int n_frames, long stats[6]) {
  MobiStreamParser A(w, h, version);
  Clip *C = (Clip *)mobi_lshost_create(w, h, version);
  if (!C) return -1;
  ParsedFrame pa, pb;
  static const int interior[16] = {9, 10, 11, 12, 17, 18, 19, 20, 25, 26, 27, 28, 33, 34, 35, 36};
  for (int k = 0; k < 6; k++) stats[k] = 0;
  bool ready = true; // a new decoder's state is the device's
  for (int f = 0; f < n_frames; f++) {
    const uint8_t *d = data + frame_off[f];
    const size_t len = frame_off[f + 1] - frame_off[f];
    MobiDevState s0, s1;
    MobiDevTail t0, t1;
    A.export_state(s0, t0);
    int32_t offa = 0, offb = 0;
    const int rca = A.parse_frame(d, len, &offa, pa);
    A.export_state(s1, t1);
    if (!ready) { stats[3]++; ready = A.device_ready(); continue; }
    MobiDevState sp = s0;
    for (int i : interior) sp.mcache[i] = 0xFF;
    sp.predx = 0x5A5A; sp.predy = -0x2525;
    MobiStreamParser B(w, h, version);
    B.import_state(sp, t0);
    const int rcb = B.parse_frame(d, len, &offb, pb);
    stats[0]++;
    bool same = rca == rcb && offa == offb;
    if (same && rca == MOBI_OK)
      same = pa.payload == pb.payload && pa.intra_mbs == pb.intra_mbs && pa.intra_items == pb.intra_items && pa.desc.size() == pb.desc.size() &&
             memcmp(pa.desc.data(), pb.desc.data(), pa.desc.size() * sizeof(MbDesc)) == 0 && memcmp(&pa.hdr, &pb.hdr, sizeof(FrameHdr)) == 0;
    if (same) stats[1]++;
    else fprintf(stderr, "framedep: frame %d parses differently from a poisoned state (rc %d / %d)\n", f, rca, rcb);
    if (A.device_ready()) { // the state afterwards, merged the way a chain of frame-parallel lanes would merge it
      MobiDevState sb;
      MobiDevTail tb;
      B.export_state(sb, tb);
      for (int i : interior) if (sb.mcache[i] == 0xFF) sb.mcache[i] = s0.mcache[i];
      if (pa.hdr.frame_type == 1 || rca != MOBI_OK) { if (sb.predx == 0x5A5A) sb.predx = s0.predx; if (sb.predy == -0x2525) sb.predy = s0.predy; }
      if (memcmp(&sb, &s1, sizeof(sb)) == 0 && memcmp(&tb, &t1, sizeof(tb)) == 0) stats[2]++;
      else fprintf(stderr, "framedep: frame %d leaves a different state from a poisoned start\n", f);
    } else
      stats[2]++; // (the host parser keeps the clip: nothing to merge)
    // the lock-step parser's lane functions from the same poisoned state
    {
      C->quant = sp.quant; C->yuvfmt = sp.yuvfmt; C->tables_set = sp.tables_set; C->frames_started = sp.frames_started; C->predx = sp.predx; C->predy = sp.predy;
      memcpy(C->m.mc_, sp.mcache, 40);
      for (int i = 0; i < C->g.mbw + 2; i++) C->m.mvp_[i] = mobi_leaf_w1(t0.mvc[2 * i], t0.mvc[2 * i + 1]);
      int32_t used = 0;
      uint32_t n_intra = 0, pay_words = 0, ftype = 0;
      const int bail = mobi_lshost_parse(C, d, len, &used, &n_intra, &pay_words, &ftype);
      if (bail) stats[5]++;
      else {
        bool eq = rca == MOBI_OK && used == offa && pay_words == pa.payload.size() && n_intra == pa.hdr.n_intra && C->quant == s1.quant && C->tables_set == s1.tables_set &&
                  C->yuvfmt == s1.yuvfmt && memcmp(C->pay.data(), pa.payload.data(), pa.payload.size() * 4) == 0;
        for (size_t mb = 0; eq && mb < pa.desc.size(); mb++) eq = C->desc[mb].payload_off == pa.desc[mb].payload_off && C->desc[mb].w1 == pa.desc[mb].w1 && C->desc[mb].w2 == pa.desc[mb].w2 && C->desc[mb].w3 == pa.desc[mb].w3;
        uint8_t mc[40];
        memcpy(mc, C->m.mc_, 40);
        for (int i : interior) if (mc[i] == 0xFF) mc[i] = s0.mcache[i];
        eq = eq && memcmp(mc, s1.mcache, 40) == 0;
        if (ftype == 0) eq = eq && C->predx == s1.predx && C->predy == s1.predy;
        if (eq) stats[4]++;
        else fprintf(stderr, "framedep: frame %d: the lock-step lane functions differ from a poisoned state (rc %d, consumed %d / %d, payload %u / %zu, intra %u / %u, quant %u / %u, tables %u / %u)\n", f, rca, used, offa,
                     pay_words, pa.payload.size(), n_intra, pa.hdr.n_intra, C->quant, s1.quant, C->tables_set, s1.tables_set);
      }
    }
    ready = A.device_ready();
  }
  mobi_lshost_destroy(C);
  return 0;
}
// ---- r06: the frame-parallel chain (mobi_gop.h) on the CPU, with the same functions the two device kernels call -----------------------------
// The clip's frames in groups of K: the start state of every frame of a group is PREDICTED from the group's true start state and the frame
// headers (mobi_gop_next_guess, what mobi_gop_prepare does), every frame is parsed by the lock-step lane functions from its predicted state
// (independently: nothing of frame k - 1's parse is used), and the chain (mobi_gop_chain) verifies each prediction against the merged truth and
// merges.  The result -- command lists and the state behind every frame -- must be the host parser's, which parses in stream order.
// stats: [0] frames parsed from a predicted state and verified, [1] predictions that did not hold (the rest of the group goes to the host
// parser), [2] frames a lane bailed out of or the host parser rejected (same), [3] frames left to the host parser, [4] DIFFERENCES (must be 0).
int mobi_gop_host_check(uint32_t w, uint32_t h, int version, const uint8_t *data, const uint32_t *frame_off, int n_frames, int K, long stats[5]) {
  MobiStreamParser A(w, h, version);
  Clip *C = (Clip *)mobi_lshost_create(w, h, version);
  if (!C || K < 1 || K > MOBI_GOP_MAX) return -1;
  for (int k = 0; k < 5; k++) stats[k] = 0;
  ParsedFrame pa;
  const int moflex = version == 2;
  uint8_t hdr[8];
  for (int f0 = 0; f0 < n_frames;) {
    const int kk = std::min(K, n_frames - f0);
    MobiDevState cur;
    MobiDevTail t0;
    A.export_state(cur, t0);
    if (f0 > 0 && !A.device_ready()) { // the host parser's clip: it parses until its state is one the device holds again
      int32_t off = 0;
      A.parse_frame(data + frame_off[f0], frame_off[f0 + 1] - frame_off[f0], &off, pa);
      stats[3]++;
      f0++;
      continue;
    }
    MobiDevState guess[MOBI_GOP_MAX];
    guess[0] = cur;
    for (int k = 1; k < kk; k++) {
      const uint32_t len = frame_off[f0 + k] - frame_off[f0 + k - 1];
      memset(hdr, 0, sizeof(hdr));
      memcpy(hdr, data + frame_off[f0 + k - 1], std::min<uint32_t>(len, 8));
      guess[k] = guess[k - 1];
      mobi_gop_next_guess(moflex, hdr, len, guess[k]);
    }
    bool broken = false;
    int k = 0;
    for (; k < kk && !broken; k++) {
      const uint8_t *d = data + frame_off[f0 + k];
      const size_t len = frame_off[f0 + k + 1] - frame_off[f0 + k];
      // the lane's parse, from the PREDICTED state (it does not know what the frame before left)
      C->quant = guess[k].quant; C->yuvfmt = guess[k].yuvfmt; C->tables_set = guess[k].tables_set; C->frames_started = guess[k].frames_started;
      C->predx = guess[k].predx; C->predy = guess[k].predy;
      memcpy(C->m.mc_, guess[k].mcache, 40);
      int32_t used = 0, offa = 0;
      uint32_t n_intra = 0, pay_words = 0, ftype = 0;
      const int bail = mobi_lshost_parse(C, d, len, &used, &n_intra, &pay_words, &ftype);
      // the truth
      const int rca = A.parse_frame(d, len, &offa, pa);
      MobiDevState s1;
      MobiDevTail t1;
      A.export_state(s1, t1);
      if (k > 0 && !mobi_gop_guess_ok(guess[k], cur)) { stats[1]++; broken = true; break; }
      if (bail || rca != MOBI_OK) {
        if (!bail) { fprintf(stderr, "gop check: frame %d finished by the lane functions, rejected (%d) by the host parser\n", f0 + k, rca); stats[4]++; }
        stats[2]++;
        broken = true;
        break;
      }
      MobiDevState out;
      memset(&out, 0, sizeof(out));
      out.quant = C->quant; out.yuvfmt = C->yuvfmt; out.tables_set = C->tables_set; out.frames_started = C->frames_started; out.predx = C->predx; out.predy = C->predy;
      memcpy(out.mcache, C->m.mc_, 40);
      mobi_gop_merge(cur, ftype == 1, out);
      bool eq = used == offa && pay_words == pa.payload.size() && n_intra == pa.hdr.n_intra && memcmp(C->pay.data(), pa.payload.data(), pa.payload.size() * 4) == 0;
      for (size_t mb = 0; eq && mb < pa.desc.size(); mb++) eq = C->desc[mb].payload_off == pa.desc[mb].payload_off && C->desc[mb].w1 == pa.desc[mb].w1 && C->desc[mb].w2 == pa.desc[mb].w2 && C->desc[mb].w3 == pa.desc[mb].w3;
      if (A.device_ready() && memcmp(&out, &s1, sizeof(out)) != 0) eq = false; // (a host-only state behind the frame: the host keeps the clip)
      if (!eq) { fprintf(stderr, "gop check: frame %d differs when parsed from its predicted state\n", f0 + k); stats[4]++; }
      stats[0]++;
      cur = out;
      if (!A.device_ready()) { k++; break; } // what follows starts from a state the device does not hold: the next group decides
    }
    if (broken) { // the frame at k and the ones behind it are the host parser's: A has parsed frame k already (above), the rest follows
      stats[3]++;
      for (int j = k + 1; j < kk; j++) {
        int32_t off = 0;
        A.parse_frame(data + frame_off[f0 + j], frame_off[f0 + j + 1] - frame_off[f0 + j], &off, pa);
        stats[3]++;
      }
      f0 += kk;
    } else
      f0 += k;
  }
  mobi_lshost_destroy(C);
  return 0;
}
}
