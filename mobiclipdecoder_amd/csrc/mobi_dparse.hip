// mobi_dparse.hip -- device-side bitstream parser for gfx950: one wave per clip.
//
// Same syntax walk as mobi_parse.cpp (which cites MobiclipDecoder.cs, "MD.cs", line by line): bit reader (:2970-3015),
// frame headers (:113-143, :224-236), MV prediction (:163-208), partition tree (:469-1746), residual CBP/VLC
// (:1818-1833, :2909-2968, :3330-3432), intra macroblock syntax (:1759-1880, :2776-2902).  The parse of one clip is a
// chain of data-dependent reads, so a wave runs it on ONE lane; the machine is filled by clips instead (8 waves per SIMD
// x 1024 SIMDs), which is what 288 GB of HBM buys: thousands of resident clips.  What the other 63 lanes do: copy the VLC
// tables into LDS, and blank the descriptors of a clip whose parse failed.
//
// Differences from the host parser, none visible in the planes:
//  * levels go straight to their place in the payload (no per-macroblock staging);
//  * intra macroblocks are listed in raster order (a valid order for the dependency waits of mobi_recon_intra: every
//    dependency is raster-earlier); the launch interleaves clips so that a clip's chain never fills the machine;
//  * exceptions become a sticky error code -- which, since r05, nobody outside sees: a frame this parser cannot finish (any error, any
//    refusal) is parsed again by the HOST parser inside the same call, from the state the frame started with (the state ring: state_in /
//    state_out, mobi_state.h, mobi_abi.cpp), and rc / Offset / planes are the host parser's.  All this parser owes is to be right on frames
//    that decode without incident and never to report one that does not as finished.  (What follows is how the error paths behave
//    anyway: they were written to match the reference's exceptions and still do.)
//  * The first error freezes the bit reader (Offset stays where the reference threw) and
//    the decoder state that survives a frame; the walk then runs on to the end of the current macroblock without testing the
//    code after every read -- every loop of the syntax is bounded by its structure, not by the data (a macroblock holds at
//    most 127 partition nodes, 6 areas of 4 blocks, 64 tokens per block) -- and the frame loop stops there.  Testing after
//    every read costs 23 % of the kernel (execution-mask bookkeeping of ~40 extra branches per macroblock).
#include <hip/hip_runtime.h>

#include "../../include/mobiclip_hip.h"
#include "mobi_dparse.h"
#include "mobi_kernels.h"

namespace {
enum { PWAVES = 4 }; // clips per workgroup: they share one LDS copy of the tables

struct WaveLds { // private to one wave = one clip
  uint32_t leaves[128];            // MC leaves of the current macroblock (mobi_leaf_w0 / w1 pairs)
  uint32_t recs[MOBI_INTRA_RECORDS];
  int32_t mvc[2 * (64 + 2)];       // MV row cache, Internal[221..] (MD.cs:163-208)
  uint32_t stk[16];                // partition-tree work stack
  uint8_t mcache[40];              // Internal[0..9]
};

__device__ __forceinline__ uint32_t shl(uint32_t x, int n) { return x << (n & 31); } // C# masks shift counts to 5 bits
__device__ __forceinline__ uint32_t shr(uint32_t x, int n) { return x >> (n & 31); }
__device__ __forceinline__ int clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

typedef const __attribute__((address_space(3))) uint8_t *lds_u8;
typedef const __attribute__((address_space(3))) uint16_t *lds_u16;

// Bit reader (MD.cs:2970-3015).  Small enough to travel through a function call in registers.
struct BitR {
  // stream: Data[Offset ..), 8-byte aligned on the device; two 8-byte registers run ahead of the reader
  const uint8_t *pf;
  uint64_t cur, nxt;
  int cur_n, len, off;
  int lim;      // = len while no error is pending, INT_MIN afterwards: one comparison decides whether a refill may happen
  uint32_t win; // r3
  int nbr;      // nrBitsRemaining
  int err;      // sticky: the first MOBI_E_* (the reference's exception); nothing is read after it

  __device__ __forceinline__ void fail(int c) {
    if (!err) err = c;
    lim = (int)0x80000000;
  }
  __device__ __forceinline__ void fill_bits() { // FillBits: one 16-bit LE word, no refill at/after Data.Length
    if (off + 1 >= lim) { // rare: an error is pending (reader frozen), or the data ends here
      if (!err && off < len) fail(MOBI_E_INDEX); // off + 1 == len: IOUtil.ReadU16LE past the array; off >= len: FillBits returns silently
      return;
    }
    const uint32_t w = (uint32_t)cur & 0xFFFFu;
    cur >>= 16;
    if (--cur_n == 0) {
      cur = nxt;
      cur_n = 4;
      pf += 8;
      nxt = *(const uint64_t *)pf;
    }
    off += 2;
    nbr += 16;
    win |= shl(w, 16 - nbr);
  }
  __device__ __forceinline__ void take(int n) {
    win = shl(win, n);
    nbr -= n;
    if (nbr < 0) fill_bits();
  }
  __device__ __forceinline__ uint32_t ue() { // Elias-gamma, value = 2^z - 1 + suffix
    const int z = clz32(win);
    win = shl(win, z);
    win += win;
    uint32_t v = (z == 0) ? 0 : shr(win, 32 - z);
    v += shl(1u, z);
    v--;
    win = shl(win, z);
    nbr -= 2 * z;
    if (--nbr < 0) fill_bits();
    return v;
  }
  __device__ __forceinline__ int se() { // odd codes map to non-positive values (MD.cs:3009-3010)
    const int z = clz32(win);
    win = shl(win, z);
    win += win;
    uint32_t u = (z == 0) ? 0 : shr(win, 32 - z);
    u += shl(1u, z);
    int v = (int)u;
    if (v & 1) v = (int)(1u - u);
    v >>= 1;
    win = shl(win, z);
    nbr -= 2 * z;
    if (--nbr < 0) fill_bits();
    return v;
  }
};

// ---------------------------------------------------------------- residual block (MD.cs:3330-3432)
// One call per coded transform block; the only real function call of the parser (everything else is inlined once), so
// that the token loop exists once in the instruction cache.  flags: [0] 8x8, [1] VLC table 1, [2] dequant tables set up.
struct ResidOut { BitR r; uint32_t n_coefs; };
__device__ __noinline__ ResidOut resid_block_fn(BitR r, uint32_t n_coefs, lds_u8 T, uint32_t *out, int tile, uint32_t flags) {
  const bool is8 = flags & 1;
  const int N = is8 ? 64 : 16;
  lds_u16 A = (lds_u16)(T + ((flags & 2) ? MOBI_DT_A1 : MOBI_DT_A0));
  lds_u8 B = T + ((flags & 2) ? MOBI_DT_B1 : MOBI_DT_B0);
  lds_u8 zz = T + (is8 ? MOBI_DT_ZZ8 : MOBI_DT_ZZ4);
  int p = 0;
  for (;;) {
    int skip, value;
    uint32_t e;
    if ((r.win >> 25) == 3) { // escape prefix 0000011
      r.win <<= 7;
      bool c = (r.win >> 31) == 1;
      r.win <<= 1;
      if (!c) { // "0": table code, level += B[last<<6|run]
        r.nbr -= 8;
        if (r.nbr < 0) r.fill_bits();
        e = A[r.win >> 20];
        value = (int)((e >> 4) & 0x1F) + B[e >> 9];
        r.win = shl(r.win, (int)(e & 0xF) - 1);
        if (r.win >> 31) value = -value;
        r.win <<= 1;
        r.nbr -= (int)(e & 0xF);
        if (r.nbr < 0) r.fill_bits();
        skip = (int)((e >> 9) & 0x3F);
        e >>= 15;
      } else {
        c = (r.win >> 31) == 1;
        r.win <<= 1;
        r.nbr -= 9;
        if (r.nbr < 0) r.fill_bits();
        if (!c) { // "10": table code, run += B[0x80 + level + (last<<6)]
          e = A[r.win >> 20];
          value = (int)((e >> 4) & 0x1F);
          skip = (int)((e >> 9) & 0x3F) + B[0x80 + value + ((e >> 15) << 6)];
          r.win = shl(r.win, (int)(e & 0xF) - 1);
          if (r.win >> 31) value = -value;
          r.win <<= 1;
          r.nbr -= (int)(e & 0xF);
          if (r.nbr < 0) r.fill_bits();
          e >>= 15;
        } else { // "11": raw last(1) run(6) level(s12)
          e = r.win >> 31;
          r.win <<= 1;
          skip = (int)(r.win >> 26);
          r.win <<= 6;
          r.nbr -= 7;
          if (r.nbr < 0) r.fill_bits();
          value = (int32_t)r.win >> 20;
          r.win <<= 12;
          r.nbr -= 12;
          if (r.nbr < 0) r.fill_bits();
        }
      }
    } else {
      e = A[r.win >> 20];
      value = (int)((e >> 4) & 0x1F);
      r.win = shl(r.win, (int)(e & 0xF) - 1);
      if (r.win >> 31) value = -value;
      r.win <<= 1;
      r.nbr -= (int)(e & 0xF);
      if (r.nbr < 0) r.fill_bits();
      skip = (int)((e >> 9) & 0x3F);
      e >>= 15;
    }
    p += skip; // (no test for a pending error here: the reader is frozen, p still advances, the loop ends within N tokens)
    if (p >= N) { r.fail(MOBI_E_UNSUPPORTED); break; } // the reference would walk past the dequant words (Internal[] aliasing)
    const int idx = (flags & 4) ? zz[p] : 0; // low byte of the dequant word = zigzag target (MD.cs:3426); all zero before the first SetupQuantTables
    p++;
    if (value != 0) out[n_coefs++] = (uint32_t)(tile + idx) | ((uint32_t)(int)(int16_t)value << 16);
    else r.fail(MOBI_E_UNSUPPORTED); // a token without a level: the frame's command list would not name every token (mobi_state.h): the host parser's
    if (e & 1) break;
  }
  return ResidOut{r, n_coefs};
}

struct DP {
  lds_u8 T; // tables in LDS
  WaveLds *L;
  int width, height, stride, lg, mbw, mbh, version, ver;
  BitR r;
  // decoder state
  uint32_t quant, yuvfmt, tables_set;
  int vlc, frames_started, predx, predy;
  // output
  MbDesc *desc;      // this clip's row of the table
  uint32_t *pay;     // whole arena
  uint32_t pay_base, pay_pos, pay_cap;
  uint32_t *items;
  uint32_t n_items, clip;
  // macroblock under construction
  int cur_mb, cur_x, cur_y, cur_off, n_leaf_words, mb_type;
  uint32_t cbp6, t8mask, w3, mb_pay, hdr_words, n_coefs;

  __device__ __forceinline__ void fail(int c) { r.fail(c); }
  __device__ __forceinline__ void fill_bits() { r.fill_bits(); }
  __device__ __forceinline__ void take(int n) { r.take(n); }
  __device__ __forceinline__ uint32_t ue() { return r.ue(); }
  __device__ __forceinline__ int se() { return r.se(); }

  // ---------------------------------------------------------------- quantiser (MD.cs:3884-3925)
  __device__ __forceinline__ void setup_quant(uint32_t q) {
    if (version == MOBI_VERSION_MOFLEX3DS) q = q < 12 ? 12 : q > 52 ? 52 : q;
    quant = q; // assigned before the table index can throw
    if (q >= 54) { fail(MOBI_E_INDEX); return; }
    tables_set = 1;
    L->mcache[1] = 9; L->mcache[2] = 9; L->mcache[3] = 9; L->mcache[4] = 9; // "no neighbour" marks, re-armed only here
    L->mcache[8] = 9; L->mcache[0x10] = 9; L->mcache[0x18] = 9; L->mcache[0x20] = 9;
  }

  // ---------------------------------------------------------------- geometry (MobiGeom of mobi_parse.h)
  __device__ __forceinline__ int owner_luma(int a) const {
    if (a < 0) return -1;
    const int row = a >> lg, col = a & (stride - 1);
    if (col >= width || row >= height) return -1;
    return (row >> 4) * mbw + (col >> 4);
  }
  __device__ __forceinline__ int owner_chroma(int a) const {
    if (a < 0) return -1;
    const int row = a >> lg, col = a & (stride - 1);
    const int x = col >= stride / 2 ? col - stride / 2 : col;
    if (x >= width / 2 || row >= height / 2) return -1;
    return (row >> 3) * mbw + (x >> 3);
  }
  __device__ __forceinline__ int area_offset(int area, int sub) const {
    const int S = stride;
    const int o = (area < 4) ? cur_off + (area >> 1) * 8 * S + (area & 1) * 8 : cur_off / 2 + (area == 5 ? S / 2 : 0);
    return o + (sub >> 1) * 4 * S + (sub & 1) * 4;
  }

  // ---------------------------------------------------------------- per-macroblock assembly
  __device__ __forceinline__ void begin_mb(int mb, int mx, int my, int type) {
    cur_mb = mb;
    cur_x = mx * 16;
    cur_y = my * 16;
    cur_off = cur_y * stride + cur_x;
    n_leaf_words = 0;
    n_coefs = 0;
    cbp6 = t8mask = w3 = 0;
    mb_type = type;
    mb_pay = pay_pos;
    hdr_words = 0;
    if (pay_pos + MOBI_MV_CELLS + 384 > pay_cap) fail(MOBI_E_DEVICE); // a macroblock's payload is at most 64 + 6*64 words
  }
  __device__ __forceinline__ void begin_intra_payload() {
    hdr_words = MOBI_INTRA_RECORDS;
    #pragma nounroll
    for (int i = 0; i < MOBI_INTRA_RECORDS; i++) L->recs[i] = 0;
  }
  // inter macroblock: all leaves are known once the partition tree is done; decide how they travel, then the levels follow
  __device__ __forceinline__ int classify_leaves(uint32_t &nl) const {
    nl = (uint32_t)n_leaf_words / 2;
    int dual = MOBI_DUAL_NONE;
    if (nl == 2) {
      const uint32_t a = L->leaves[0] & 0xFFF, b = L->leaves[2] & 0xFFF;
      if (a == (0u | (1u << 10)) && b == ((4u << 4) | (1u << 10))) dual = MOBI_DUAL_TB;
      if (a == (0u | (1u << 8)) && b == (4u | (1u << 8))) dual = MOBI_DUAL_LR;
    }
    return dual;
  }
  bool dep_intra = false;
  __device__ __forceinline__ void add_dep(int o, uint32_t *deps, int &n_deps) {
    if (o < 0 || o >= cur_mb) return; // raster-later owners read as the fresh plane's zeros (the kernel masks them)
    #pragma nounroll
    for (int k = 0; k < n_deps; k++)
      if ((int)(deps[k] & 0x1FFF) == o) return;
    if (n_deps == MOBI_INTRA_DEPS) { fail(MOBI_E_UNSUPPORTED); return; }
    const uint32_t inter = (desc[o].w1 & 1) == MOBI_MB_INTRA ? 0u : MOBI_DEP_INTER; // this lane wrote desc[o] itself
    deps[n_deps++] = (uint32_t)o | inter;
    if (!inter) { desc[o].w3 |= 4u; dep_intra = true; } // w3 [1] has intra dependencies, [2] has intra dependents (mobi_recon_intra_cl)
  }
  __device__ __forceinline__ void end_mb() {
    if (r.err) return;
    MbDesc d;
    d.payload_off = pay_base + mb_pay;
    d.w2 = n_coefs;
    d.w3 = w3;
    d.w4 = d.w5 = d.w6 = d.w7 = 0;
    uint32_t nl = 0;
    int dual = MOBI_DUAL_NONE;
    if (mb_type == MOBI_MB_INTER) {
      dual = classify_leaves(nl);
      if (nl == 1 || dual) { // leaf records: positions and phases instead of motion vectors (MD.cs:400-416)
        const int S = stride;
        uint32_t pos[4] = {0, 0, 0, 0};
        #pragma nounroll
        for (uint32_t i = 0; i < nl; i++) {
          const uint32_t w0 = L->leaves[2 * i], w1 = L->leaves[2 * i + 1];
          const int ref = (w0 >> 12) & 7;
          const int dx = (int16_t)(w1 & 0xFFFF), dy = (int16_t)(w1 >> 16), cdx = dx >> 1, cdy = dy >> 1;
          pos[2 * i] = (uint32_t)(cur_off + (dy >> 1) * S + (dx >> 1));
          pos[2 * i + 1] = (uint32_t)(cur_off / 2 + (cdy >> 1) * S + (cdx >> 1));
          d.w2 |= (uint32_t)ref << (10 + 3 * i);
          d.w2 |= (uint32_t)((dx & 1) | ((dy & 1) << 1)) << (16 + 4 * i);
          d.w2 |= (uint32_t)((cdx & 1) | ((cdy & 1) << 1)) << (18 + 4 * i);
        }
        d.w3 = pos[0]; d.w4 = pos[1]; d.w5 = pos[2]; d.w6 = pos[3];
      } else { // the 64-entry MV cell map, in the 64 words p_residual() left free in front of the levels
        uint32_t *cells = pay + pay_base + mb_pay;
        #pragma nounroll
        for (int i = 0; i + 1 < n_leaf_words; i += 2) {
          const uint32_t w0 = L->leaves[i], mv = L->leaves[i + 1];
          const int x = (int)(w0 & 15) * 2, y = (int)((w0 >> 4) & 15) * 2, w = 16 >> ((w0 >> 8) & 3), h = 16 >> ((w0 >> 10) & 3);
          const uint32_t cell = mobi_cell((int16_t)(mv & 0xFFFF), (int16_t)(mv >> 16), (int)((w0 >> 12) & 7));
          #pragma nounroll
          for (int cy = y >> 1; cy < (y + h) >> 1; cy++)
            #pragma nounroll
            for (int cx = x >> 1; cx < (x + w) >> 1; cx++) cells[cy * 8 + cx] = cell;
        }
      }
    } else {
      uint32_t *rec_out = pay + pay_base + mb_pay;
      #pragma nounroll
      for (int i = 0; i < MOBI_INTRA_RECORDS; i++) rec_out[i] = L->recs[i];
      // the raster-earlier macroblocks this one's prediction halo touches (finish_levels of mobi_parse.cpp scans the whole
      // halo; these probes hit every distinct owner in the same order: the halo above is three 16-aligned runs, the columns
      // left and right change owner at most once, between the first row and the rest)
      uint32_t deps[MOBI_INTRA_DEPS];
      int n_deps = 0;
      dep_intra = false;
      const int S = stride, o = cur_off;
      add_dep(owner_luma(o - S - 1), deps, n_deps);
      add_dep(owner_luma(o - S), deps, n_deps);
      add_dep(owner_luma(o - S + 16), deps, n_deps);
      add_dep(owner_luma(o - 1), deps, n_deps);
      add_dep(owner_luma(o + 16), deps, n_deps);
      add_dep(owner_luma(o + S - 1), deps, n_deps);
      add_dep(owner_luma(o + S + 16), deps, n_deps);
      #pragma nounroll
      for (int v = 0; v < 2; v++) {
        const int b = o / 2 + v * (S / 2);
        add_dep(owner_chroma(b - S - 1), deps, n_deps);
        add_dep(owner_chroma(b - S), deps, n_deps);
        add_dep(owner_chroma(b - S + 8), deps, n_deps);
        add_dep(owner_chroma(b - 1), deps, n_deps);
        add_dep(owner_chroma(b + 8), deps, n_deps);
        add_dep(owner_chroma(b + S - 1), deps, n_deps);
        add_dep(owner_chroma(b + S + 8), deps, n_deps);
      }
      #pragma nounroll
      for (int k = n_deps; k < MOBI_INTRA_DEPS; k++) deps[k] = MOBI_DEP_NONE;
      d.w4 = deps[0] | (deps[1] << 16);
      d.w5 = deps[2] | (deps[3] << 16);
      d.w6 = deps[4] | (deps[5] << 16);
      d.w7 = deps[6] | (deps[7] << 16);
      if (dep_intra) d.w3 |= 2u;
      items[n_items++] = MOBI_ITEM(clip, cur_mb);
    }
    d.w1 = (uint32_t)mb_type | (nl << 1) | (cbp6 << 8) | (t8mask << 14) | ((quant & 63) << 20) | ((uint32_t)dual << 26);
    uint4 *dp = (uint4 *)(desc + cur_mb);
    dp[0] = uint4{d.payload_off, d.w1, d.w2, d.w3};
    dp[1] = uint4{d.w4, d.w5, d.w6, d.w7};
    pay_pos = mb_pay + hdr_words + n_coefs;
  }

  // ---------------------------------------------------------------- motion (MD.cs:400-456)
  __device__ __forceinline__ void check_window(long long pos, int w, int h, int phase, long long plane_len) {
    if (pos < 0) { fail(MOBI_E_INDEX); return; }
    const long long last = pos + (long long)(h - 1) * stride;
    long long hi = last + w - 1;           // phase 0: Array.Copy end is exclusive
    if (phase & 1) hi += 1;
    if (phase & 2) hi += stride;
    if (hi >= plane_len) fail(MOBI_E_INDEX);
  }
  __device__ __forceinline__ void mc_leaf(int wi, int hi, int x, int y, int ref, int dx, int dy, int mv_slot) {
    const int w = 16 >> wi, h = 16 >> hi;
    L->mvc[mv_slot] = dx; // every leaf overwrites the macroblock's exported MV (MD.cs:411-412)
    L->mvc[mv_slot + 1] = dy;
    if (ref > imin(5, frames_started - 1)) { fail(MOBI_E_NULLREF); return; } // Y[ref] == null
    const bool in_range = dx >= -MOBI_MV_LIMIT && dx <= MOBI_MV_LIMIT && dy >= -MOBI_MV_LIMIT && dy <= MOBI_MV_LIMIT;
    if (in_range) { // every position fits 32 bits: would CopyBlock throw?  (rows are visited top to bottom: first row / last row bound the rest)
      const int S = stride, o = cur_off + y * S + x, ylen = S * height;
      const int pos = o + (dy >> 1) * S + (dx >> 1);
      const int hi_y = pos + (h - 1) * S + w - 1 + (dx & 1) + ((dy & 1) ? S : 0); // phase 0: Array.Copy end is exclusive
      const int cdx = dx >> 1, cdy = dy >> 1;
      const int cpos = o / 2 + (cdy >> 1) * S + (cdx >> 1);
      const int hi_c = cpos + S / 2 + ((h >> 1) - 1) * S + (w >> 1) - 1 + (cdx & 1) + ((cdy & 1) ? S : 0); // the V window ends last
      if (pos < 0 || hi_y >= ylen || cpos < 0 || hi_c >= ylen / 2) { fail(MOBI_E_INDEX); return; }
    } else {
      const long long S = stride;
      const long long o = (long long)cur_off + (long long)y * S + x;
      check_window(o + (long long)(dy >> 1) * S + (dx >> 1), w, h, (dx & 1) | ((dy & 1) << 1), S * height);
      const int cdx = dx >> 1, cdy = dy >> 1;
      const long long cpos = o / 2 + (long long)(cdy >> 1) * S + (cdx >> 1);
      const int cph = (cdx & 1) | ((cdy & 1) << 1);
      check_window(cpos, w >> 1, h >> 1, cph, S * height / 2);
      check_window(cpos + S / 2, w >> 1, h >> 1, cph, S * height / 2);
      fail(MOBI_E_UNSUPPORTED); // (an index error above wins: the first error sticks)
      return;
    }
    L->leaves[n_leaf_words] = mobi_leaf_w0(x, y, wi, hi, ref);
    L->leaves[n_leaf_words + 1] = mobi_leaf_w1(dx, dy);
    n_leaf_words += 2;
  }

  // ---------------------------------------------------------------- residual (MD.cs:3330-3432)
  __device__ __forceinline__ void resid_block(int area, int sub, bool is8) {
    if (quant < 12) { fail(MOBI_E_UNSUPPORTED); return; } // see mobi_parse.cpp: below q=12 the reference depends on Internal[] aliasing
    const ResidOut o = resid_block_fn(r, n_coefs, T, pay + pay_base + mb_pay + hdr_words, is8 ? area * 64 : area * 64 + sub * 16,
                                      (is8 ? 1u : 0u) | (vlc == 1 ? 2u : 0u) | (tables_set ? 4u : 0u));
    r = o.r;
    n_coefs = o.n_coefs;
  }
  __device__ __forceinline__ void resid_area(int area) { // loc_11652C, MD.cs:2909-2929
    if (r.win >> 31) {
      r.win += r.win;
      r.nbr--;
      t8mask |= 1u << area;
      resid_block(area, 0, true);
    } else {
      const uint32_t u = ue();
      if (u >= 16) { fail(MOBI_E_INDEX); return; }
      const uint32_t m = T[MOBI_DT_CBP4_P + u];
      #pragma nounroll
      for (int sub = 0; sub < 4; sub++)
        if ((m >> sub) & 1) {
          resid_block(area, sub, false);
        }
    }
  }
  __device__ __forceinline__ void p_residual() { // loc_1161A0, MD.cs:1818-1833
    uint32_t nl;
    const int dual = classify_leaves(nl);
    hdr_words = (nl == 1 || dual) ? 0 : MOBI_MV_CELLS; // end_mb() fills the cell map
    const uint32_t u = ue();
    if (u >= 64) { fail(MOBI_E_INDEX); return; }
    cbp6 = T[MOBI_DT_CBP_P + u];
    #pragma nounroll
    for (int area = 0; area < 6; area++)
      if ((cbp6 >> area) & 1) {
        resid_area(area);
      }
  }

  // ---------------------------------------------------------------- intra syntax
  __device__ __forceinline__ void check_intra_reads(int mode, int o) { // see mobi_parse.cpp
    const uint32_t top = 0x1E5, left = 0x0F6; // modes 0,2,5,6,7,8 read the row above; 1,2,4,5,6,7 the column to the left
    if (((top >> mode) & 1) && o < stride) { fail(MOBI_E_INDEX); return; }
    if (((left >> mode) & 1) && o < 1) fail(MOBI_E_INDEX);
  }
  // predicted-mode code shared by loc_116220 / loc_116368 / sub_1163DC (MD.cs:1840-1859, 2785-2804, 2841-2858)
  __device__ __forceinline__ int pmode(int ci, bool four) {
    int pred = imin(L->mcache[ci - 8], L->mcache[ci - 1]);
    if (pred == 9) pred = 3;
    int v = (int)(r.win >> 28), nb = 1, mode = pred;
    if (v >= pred) v++;
    if (v < 9) { mode = v; nb = 4; }
    if (!r.err) { // the cache survives the frame: nothing may touch it after the (sticky) error = the reference's throw
      if (four) L->mcache[ci] = (uint8_t)mode;
      else L->mcache[ci] = L->mcache[ci + 1] = L->mcache[ci + 8] = L->mcache[ci + 9] = (uint8_t)mode;
    }
    take(nb);
    return mode;
  }
  // sub_116508 (MD.cs:2869-2896) or a bare PredictIntra: one 8x8 area whose mode is already known
  __device__ __forceinline__ void intra_area_fixed(int area, int mode, bool coded) {
    if (!coded) {
      check_intra_reads(mode, area_offset(area, 0));
      L->recs[area * 4] |= mobi_intra_rec(mode, 0, 0, 0, 0);
      return;
    }
    if (r.win >> 31) {
      r.win += r.win;
      r.nbr--;
      check_intra_reads(mode, area_offset(area, 0));
      L->recs[area * 4] |= mobi_intra_rec(mode, 1, 0, 0, 0);
      cbp6 |= 1u << area;
      t8mask |= 1u << area;
      resid_block(area, 0, true);
    } else {
      const uint32_t u = ue();
      if (u >= 20) { fail(MOBI_E_INDEX); return; }
      const uint32_t m4 = T[MOBI_DT_CBP4_I + u];
      #pragma nounroll
      for (int sub = 0; sub < 4; sub++) {
        check_intra_reads(mode, area_offset(area, sub));
        const int c = (m4 >> sub) & 1;
        L->recs[area * 4 + sub] |= mobi_intra_rec(mode, c, 1, 0, 0);
        if (c) {
          cbp6 |= 1u << area;
          resid_block(area, sub, false);
        }
      }
    }
  }
  __device__ __forceinline__ void intra_chroma(uint32_t cbp) { // loc_116290, MD.cs:1864-1880
    int m = (int)(r.win >> 29);
    take(3);
    if (m == 2) {
      m = 9;
      #pragma nounroll
      for (int area = 4; area < 6; area++) {
        const int p = se();
        check_intra_reads(2, area_offset(area, 0));
        if (p < -32768 || p > 32767) { fail(MOBI_E_UNSUPPORTED); return; }
        L->recs[area * 4] |= mobi_intra_rec(0, 0, 0, 1, (int16_t)p);
      }
    }
    #pragma nounroll
    for (int area = 4; area < 6; area++) {
      intra_area_fixed(area, m, (cbp >> area) & 1);
    }
  }
  __device__ __forceinline__ void intra_full_luma(uint32_t cbp) { // DecIntraFullBlockPMode, MD.cs:1759-1786
    int m = (int)(r.win >> 29);
    take(3);
    if (m == 2) {
      m = 9;
      const int p = se();
      check_intra_reads(2, cur_off);
      if (p < -32768 || p > 32767) { fail(MOBI_E_UNSUPPORTED); return; }
      w3 = 1u | ((uint32_t)(uint16_t)(int16_t)p << 16);
    }
    #pragma nounroll
    for (int k = 0; k < 4; k++) {
      intra_area_fixed(k, m, (cbp >> k) & 1);
    }
  }
  __device__ __forceinline__ void intra_sub_luma(uint32_t cbp) { // DecIntraSubBlockPMode, MD.cs:1789-1807
    #pragma nounroll
    for (int k = 0; k < 4; k++) {
      const int cik = 9 + (k & 1) * 2 + (k >> 1) * 0x10; // 9, 0xB, 0x19, 0x1B
      const bool coded = (cbp >> k) & 1;
      bool whole = true;
      if (coded) { // loc_116368, MD.cs:2776
        if (r.win >> 31) { r.win <<= 1; r.nbr--; }
        else whole = false;
      }
      if (whole) {
        const int m = pmode(cik, false);
        int p = 0;
        if (m == 2) { // the predictor itself reads its parameter (MD.cs:1915-1919)
          p = se();
          if (p < -32768 || p > 32767) { fail(MOBI_E_UNSUPPORTED); return; }
        }
        check_intra_reads(m, area_offset(k, 0));
        L->recs[k * 4] |= mobi_intra_rec(m, coded, 0, 0, (int16_t)p);
        if (coded) {
          cbp6 |= 1u << k;
          t8mask |= 1u << k;
          resid_block(k, 0, true);
        }
      } else {
        const uint32_t u4 = ue();
        if (u4 >= 20) { fail(MOBI_E_INDEX); return; }
        const uint32_t m4 = T[MOBI_DT_CBP4_I + u4];
        #pragma nounroll
        for (int sub = 0; sub < 4; sub++) {
          const int m = pmode(cik + (sub & 1) + (sub >> 1) * 8, true);
          int p = 0;
          if (m == 2) {
            p = se();
            if (p < -32768 || p > 32767) { fail(MOBI_E_UNSUPPORTED); return; }
          }
          check_intra_reads(m, area_offset(k, sub));
          const int c = (m4 >> sub) & 1;
          L->recs[k * 4 + sub] |= mobi_intra_rec(m, c, 1, 0, (int16_t)p);
          if (c) {
            cbp6 |= 1u << k;
            resid_block(k, sub, false);
          }
        }
      }
    }
  }
  // both intra macroblock kinds: CBP, luma, then the shared chroma part (MD.cs:1759-1807)
  __device__ __forceinline__ void intra_mb(bool sub) {
    begin_intra_payload();
    const uint32_t u = ue();
    if (u >= 64) { fail(MOBI_E_INDEX); return; }
    const uint32_t cbp = T[MOBI_DT_CBP_I + u];
    if (sub) intra_sub_luma(cbp); else intra_full_luma(cbp);
    intra_chroma(cbp);
  }

  // ---------------------------------------------------------------- partition tree (MD.cs:469-1746), iterative
  // work item: wi | hi<<2 | (x/2)<<4 | (y/2)<<8 ; children are pushed second-first so that the first is parsed first
  // returns 0: inter macroblock, leaves recorded; 1 / 2: the macroblock is intra (full / sub), still to be parsed
  __device__ __forceinline__ int pblock_tree(int mv_slot) {
    int sp = 0;
    L->stk[sp++] = 0;
    lds_u8 plut = T + MOBI_DT_PLUT, pbits = T + MOBI_DT_PBITS;
    while (sp > 0) {
      const uint32_t it = L->stk[--sp];
      const int wi = it & 3, hi = (it >> 2) & 3, x = ((it >> 4) & 15) * 2, y = ((it >> 8) & 15) * 2;
      const int s = wi * 4 + hi, w = 16 >> wi, h = 16 >> hi;
      const uint32_t code = plut[s * 64 + (r.win >> T[MOBI_DT_PSHIFT + s])];
      if (code >= T[MOBI_DT_PNB + s]) { fail(MOBI_E_INDEX); return 0; }
      take(pbits[s * 12 + code]);
      if (code == 0) {
        mc_leaf(wi, hi, x, y, 1, predx, predy, mv_slot);
      } else if (code <= 5) {
        const int dx = se();
        const int dy = se();
        mc_leaf(wi, hi, x, y, (int)code, dx + predx, dy + predy, mv_slot);
      } else if (code == 6 || code == 7) {
        if (s != 0) { fail(MOBI_E_PARTCODE); return 0; }
        mb_type = MOBI_MB_INTRA;
        return code == 6 ? 1 : 2; // no p_residual() for an intra macroblock
      } else if (code == 8) {
        if (h == 2) { fail(MOBI_E_PARTCODE); return 0; }
        L->stk[sp++] = (uint32_t)(wi | ((hi + 1) << 2) | ((x >> 1) << 4) | (((y + h / 2) >> 1) << 8));
        L->stk[sp++] = (uint32_t)(wi | ((hi + 1) << 2) | ((x >> 1) << 4) | ((y >> 1) << 8));
      } else if (code == 9) {
        if (w == 2) { fail(MOBI_E_PARTCODE); return 0; }
        L->stk[sp++] = (uint32_t)((wi + 1) | (hi << 2) | (((x + w / 2) >> 1) << 4) | ((y >> 1) << 8));
        L->stk[sp++] = (uint32_t)((wi + 1) | (hi << 2) | ((x >> 1) << 4) | ((y >> 1) << 8));
      }
    }
    return 0;
  }

  // ---------------------------------------------------------------- frames (MD.cs:97-259)
  __device__ __forceinline__ void parse_frame(bool iframe) {
    if (iframe) {
      yuvfmt = r.win >> 31;
      r.win += r.win;
      vlc = (int)(r.win >> 31);
      r.win += r.win;
      r.nbr -= 3;
      if (r.nbr < 0) fill_bits();
      if (r.err) return;
      const uint32_t q = r.win >> 26;
      take(6);
      if (r.err) return;
      if (quant != q) setup_quant(q);
    } else {
      if (--r.nbr < 0) fill_bits();
      if (r.err) return;
      const uint32_t q = quant;
      const int dq = se();
      if (r.err) return;
      if (version == MOBI_VERSION_MOFLEX3DS && q == 0) setup_quant(q);
      else if (dq != 0) setup_quant(q + (uint32_t)dq);
      vlc = 0; // a failed SetupQuantTables throws before these (MD.cs:131-139); nothing reads them afterwards
      #pragma nounroll
      for (int i = 0; i < 2 * (mbw + 2); i++) L->mvc[i] = 0;
    }
    if (r.err) return;
    int mx = 0, my = 0;
    #pragma nounroll
    for (int mb = 0; mb < mbw * mbh; mb++) {
      int kind;
      if (iframe) {
        kind = (r.win >> 31) == 1 ? 2 : 1;
        r.win += r.win;
        r.nbr--;
        if (r.nbr < 0) fill_bits();
        if (r.err) return;
        begin_mb(mb, mx, my, MOBI_MB_INTRA);
        if (r.err) return;
      } else {
        const int32_t *e = &L->mvc[2 * mx]; // entries: left, top, top-right (MD.cs:163-169)
        const int a0 = e[0], a1 = e[1], b0 = e[2], b1 = e[3], c0 = e[4], c1 = e[5];
        predx = imax(imin(a0, b0), imin(imax(a0, b0), c0));
        predy = imax(imin(a1, b1), imin(imax(a1, b1), c1));
        const int slot = 2 * (mx + 1);
        L->mvc[slot] = 0;
        L->mvc[slot + 1] = 0;
        begin_mb(mb, mx, my, MOBI_MB_INTER);
        if (r.err) return;
        kind = pblock_tree(slot);
        if (r.err) return;
      }
      if (kind) intra_mb(kind == 2);
      else p_residual();
      if (r.err) return;
      end_mb();
      if (r.err) return;
      if (++mx == mbw) { mx = 0; my++; }
    }
  }
};
} // namespace

#ifndef MOBI_PARSE_WAVES
#define MOBI_PARSE_WAVES 4 // waves per SIMD the register budget is cut for: 128 VGPRs; 5, 6 and 8 spill and are slower
#endif
extern "C" __global__ __launch_bounds__(64 * PWAVES) __attribute__((amdgpu_waves_per_eu(MOBI_PARSE_WAVES, MOBI_PARSE_WAVES))) void mobi_parse_frames(MobiDevParseArgs A) {
  __shared__ __attribute__((aligned(16))) uint8_t tab[MOBI_DT_BYTES];
  __shared__ WaveLds wl[PWAVES];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int clip = blockIdx.x * PWAVES + wave;
  bool finished = false; // by the lock-step parser in front (mobi_lsparse.hip): its new decoder state only has to move into place
  if (A.lockstep) {
    bool all = true;
#pragma unroll
    for (int w = 0; w < PWAVES; w++) {
      const int cw = blockIdx.x * PWAVES + w;
      const bool f = cw < A.n_clips && A.res[cw].pad == MOBI_LS_MAGIC;
      all = all && (f || cw >= A.n_clips);
      if (w == wave) finished = f;
    }
    if (finished && lane == 0) A.state_out[clip] = A.state_ls[clip];
    if (all) return; // (every thread of the workgroup sees the same four records)
  }
  for (int i = threadIdx.x; i < MOBI_DT_BYTES / 16; i += 64 * PWAVES) ((uint4 *)tab)[i] = ((const uint4 *)A.tables)[i];
  __syncthreads();
  if (clip >= A.n_clips || finished) return;
  if (A.bit_len[clip] == MOBI_DP_SKIP) return; // the host parser's clip (wave-uniform)
  const int n_mbs = A.mbw * A.mbh;
  WaveLds *L = &wl[wave];
  int rc_lane = 0, ft_lane = 0;
  if (lane == 0) {
    DP p;
    p.T = (lds_u8)tab;
    p.L = L;
    p.width = A.width; p.height = A.height; p.stride = A.stride; p.lg = A.lg; p.mbw = A.mbw; p.mbh = A.mbh;
    p.version = A.version;
    p.ver = A.version == MOBI_VERSION_MOFLEX3DS ? 0 : 1;
    const MobiDevState *st = A.state_in + clip;
    p.quant = st->quant; p.yuvfmt = st->yuvfmt; p.tables_set = st->tables_set;
    p.frames_started = st->frames_started + 1; // ring rotation + fresh planes happen before anything can throw (MD.cs:102-108)
    for (int i = 0; i < 40; i++) L->mcache[i] = st->mcache[i];
    p.vlc = 0; p.predx = st->predx; p.predy = st->predy; // (Internal[219], [220]: an I-frame leaves them as they are)
    p.desc = A.desc + (size_t)clip * n_mbs;
    p.pay = A.payload + (A.pay_local ? (size_t)clip * A.pay_cap : (size_t)0);
    p.pay_base = A.pay_local ? 0u : (uint32_t)clip * A.pay_cap;
    p.pay_pos = 0;
    p.pay_cap = A.pay_cap;
    p.items = A.items + (size_t)clip * n_mbs;
    p.n_items = 0;
    p.clip = (uint32_t)(A.clip_mod ? clip % A.clip_mod : clip);
    p.r.err = 0;
    p.cur_mb = 0; p.cur_x = p.cur_y = p.cur_off = 0; p.n_leaf_words = 0; p.mb_type = 0;
    p.cbp6 = p.t8mask = p.w3 = p.mb_pay = p.hdr_words = p.n_coefs = 0;
    // bit reader: DecodeFrame() reads the first 16-bit word itself (MD.cs:110-112)
    const uint8_t *base = A.bits + A.bit_off[clip];
    p.r.len = (int)A.bit_len[clip];
    p.r.lim = p.r.len;
    p.r.off = 0;
    p.r.nbr = 0;
    p.r.win = 0;
    p.r.cur = *(const uint64_t *)base;
    p.r.pf = base + 8;
    p.r.nxt = *(const uint64_t *)p.r.pf;
    p.r.cur_n = 4;
    uint32_t ftype = 0;
    if (p.r.len < 2) {
      p.fail(MOBI_E_INDEX);
    } else {
      p.r.nbr = -16;
      p.fill_bits(); // win = first word << 16, nbr = 0
      const bool iframe = (p.r.win >> 31) == 1;
      p.r.win += p.r.win;
      ftype = iframe;
      p.parse_frame(iframe);
    }
    MobiDevState *so = A.state_out + clip;
    so->quant = p.quant; so->yuvfmt = p.yuvfmt; so->tables_set = p.tables_set; so->frames_started = p.frames_started;
    for (int i = 0; i < 40; i++) so->mcache[i] = L->mcache[i];
    so->predx = p.predx; so->predy = p.predy;
    MobiDevResult r;
    r.rc = p.r.err;
    r.consumed = p.r.off;
    r.n_intra = p.r.err ? 0 : p.n_items;
    r.payload_words = p.pay_pos;
    r.quant = p.quant; r.yuvfmt = p.yuvfmt; r.frame_type = ftype; r.pad = 0;
    A.res[clip] = r;
    rc_lane = p.r.err;
    ft_lane = (int)ftype;
  }
  // the MV row cache a P-frame leaves (Internal[221..]): a later I-frame's walk through Internal[] may read it (mobi_state.h)
  const int rcl = __builtin_amdgcn_readfirstlane(rc_lane); // lane 0 is the first active lane
  if (rcl == 0 && __builtin_amdgcn_readfirstlane(ft_lane) == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    int32_t *mv = A.tail_out[clip].mvc;
    for (int i = lane; i < 2 * (A.mbw + 2); i += 64) mv[i] = L->mvc[i];
  }
  // a failed clip: descriptors typed "intra" that no launch list references, so nothing of it is written (mobi_abi.cpp step_write)
  if (rcl != 0) {
    uint4 *d = (uint4 *)(A.desc + (size_t)clip * n_mbs);
    for (int mb = lane; mb < n_mbs; mb += 64) {
      d[2 * mb] = uint4{0, MOBI_MB_INTRA, 0, 0};
      d[2 * mb + 1] = uint4{0, 0, 0, 0};
    }
  }
}

// What a clean frame leaves in Internal[90..217], from its command list (mobi_state.h): one lane per clip, behind the parse kernels.  A clip
// whose parse failed, or that is the host parser's anyway, is left alone: the host parser takes it over from the state the frame started with.
extern "C" __global__ __launch_bounds__(64) void mobi_parse_tail(MobiDevParseArgs A) {
  __shared__ uint8_t izz[80];
  if (threadIdx.x < 64) izz[A.tables[MOBI_DT_ZZ8 + threadIdx.x]] = (uint8_t)threadIdx.x;
  if (threadIdx.x < 16) izz[64 + A.tables[MOBI_DT_ZZ4 + threadIdx.x]] = (uint8_t)threadIdx.x;
  __syncthreads();
  const int clip = blockIdx.x * 64 + threadIdx.x;
  if (clip >= A.n_clips || A.bit_len[clip] == MOBI_DP_SKIP) return;
  const MobiDevResult r = A.res[clip];
  if (r.rc != 0) return;
  const int n_mbs = A.mbw * A.mbh;
  const MbDesc *desc = A.desc + (size_t)clip * n_mbs;
  const uint32_t *pay = A.payload + (A.pay_local ? (size_t)clip * A.pay_cap : (size_t)0);
  const uint32_t rel = A.pay_local ? 0u : (uint32_t)clip * A.pay_cap; // MbDesc.payload_off counts from the arena's start then; the scan's offsets from the clip's
  if (!A.pay_local) pay += rel;
  MobiTailScan sc;
  mobi_tail_scan_init(sc);
  for (int mb = n_mbs - 1; mb >= 0 && !sc.done; mb--) {
    const uint4 d = *(const uint4 *)(desc + mb);
    const int n = (int)(d.z & 0x3FF);
    if (!n) continue;
    const bool intra = (d.y & 1) == MOBI_MB_INTRA;
    const uint32_t nl = (d.y >> 1) & 0x7F, dual = (d.y >> 26) & 3;
    const uint32_t woff = d.x - rel + (intra ? MOBI_INTRA_RECORDS : (nl > 1 && !dual) ? MOBI_MV_CELLS : 0);
    mobi_tail_scan_mb(sc, pay + woff, n, woff, (d.y >> 14) & 0x3F, izz, izz + 64);
  }
  const MobiDevTail *in = A.tail_in + clip;
  MobiDevTail *out = A.tail_out + clip;
  mobi_tail_finish(sc, pay, A.scale + (size_t)(A.state_out[clip].quant & 63) * MOBI_SCALE_STRIDE, *in, *out);
  if (r.frame_type == 1) // an I-frame does not touch the MV row cache (MD.cs:224-249)
    for (int i = 0; i < 2 * (A.mbw + 2); i++) out->mvc[i] = in->mvc[i];
}

extern "C" int mobi_launch_parse(const MobiDevParseArgs *a, hipStream_t s) {
  if (a->n_clips <= 0) return 0;
  if (a->mbw > 64) return (int)hipErrorInvalidValue;
  if (a->lockstep)
    if (int e = mobi_launch_parse_ls(a, s)) return e;
  hipLaunchKernelGGL(mobi_parse_frames, dim3((unsigned)((a->n_clips + PWAVES - 1) / PWAVES)), dim3(64 * PWAVES), 0, s, *a);
  if (!a->skip_tail) hipLaunchKernelGGL(mobi_parse_tail, dim3((unsigned)((a->n_clips + 63) / 64)), dim3(64), 0, s, *a);
  return (int)hipGetLastError();
}
