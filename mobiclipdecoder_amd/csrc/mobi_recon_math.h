// mobi_recon_math.h -- per-pixel / per-block arithmetic of Mobiclip frame reconstruction.
//
// Host+device inline functions: the HIP kernels (mobi_kernels.hip) call them per lane, and the
// CPU command-list interpreter used by the CPU-only tests (tests/tools/mobi_cmd_interp.cpp) calls
// the very same code in loops, so kernel arithmetic is exercised without a GPU.
// Everything is integer; all forms were derived from MobiclipDecoder.cs ("MD.cs") and are cited.
#ifndef MOBI_RECON_MATH_H
#define MOBI_RECON_MATH_H
#include <stdint.h>

#if defined(__HIPCC__)
#define MOBI_HD __host__ __device__ __forceinline__
#else
#define MOBI_HD inline
#endif

// ---- inverse transforms -----------------------------------------------------------------------
// 8-point butterfly used by both passes of the 8x8 inverse transform (MD.cs:3452-3485, :3517-3550)
MOBI_HD void mobi_bfly8(const int in[8], int out[8]) {
  const int a0 = in[0] + in[4], a1 = in[0] - in[4];
  const int a2 = in[2] + (in[6] >> 1), a3 = (in[2] >> 1) - in[6];
  const int e0 = a0 + a2, e1 = a1 + a3, e2 = a1 - a3, e3 = a0 - a2;
  const int b0 = in[1] + in[7] - in[3] - (in[3] >> 1);
  const int b1 = in[7] - in[1] + in[5] + (in[5] >> 1);
  const int b2 = in[5] - (in[7] + (in[7] >> 1)) - in[3];
  const int b3 = in[3] + in[5] + in[1] + (in[1] >> 1);
  const int o0 = b2 + (b3 >> 2), o3 = b3 - (b2 >> 2);
  const int o1 = b0 + (b1 >> 2), o2 = (b0 >> 2) - b1;
  out[0] = e0 + o3; out[7] = e0 - o3;
  out[1] = e1 + o2; out[6] = e1 - o2;
  out[2] = e2 + o1; out[5] = e2 - o1;
  out[3] = e3 + o0; out[4] = e3 - o0;
}
// 4-point butterfly of the 4x4 inverse transform (MD.cs:3740-3747, :3768-3775)
MOBI_HD void mobi_bfly4(const int in[4], int out[4]) {
  const int a = in[0] + in[2], b = in[0] - in[2];
  const int c = (in[1] >> 1) - in[3], d = in[1] + (in[3] >> 1);
  out[0] = a + d; out[3] = a - d; out[1] = b + c; out[2] = b - c;
}
// residual add through the clamp table (MobiConst.cs:587: 64 zeros, 0..255, 64 x 255; MD.cs:3551).
// Returns the clamped pixel; *fault set when the table index leaves [0,384) (the reference throws).
MOBI_HD int mobi_add_clamp(int pred, int res, int *fault) {
  const int idx = 0x40 + pred + res;
  if ((unsigned)idx >= 384u) *fault = 1;
  const int v = pred + res;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// ---- motion compensation (CopyBlock, MD.cs:418-456), four pixels at a time -------------------
// a = bytes at pos..pos+3, b = pos+1.., c = pos+Stride.., d = pos+Stride+1..  (packed LE)
MOBI_HD uint32_t mobi_mc4(uint32_t a, uint32_t b, uint32_t c, uint32_t d, int phase) {
  const uint32_t M = 0x7F7F7F7Fu;
  switch (phase) {
    case 0: return a;
    case 1: return ((a >> 1) & M) + ((b >> 1) & M);              // (a>>1)+(b>>1), no carry across bytes
    case 2: return ((a >> 1) & M) + ((c >> 1) & M);
    default: {
      const uint32_t h0 = ((a >> 1) & M) + ((b >> 1) & M), h1 = ((c >> 1) & M) + ((d >> 1) & M);
      return ((h0 >> 1) & M) + ((h1 >> 1) & M);                  // truncating at every step
    }
  }
}

// ---- intra prediction ---------------------------------------------------------------------------
#define MOBI_F2(a, b) (((a) + (b) + 1) >> 1)
#define MOBI_F3(a, b, c) (((a) + 2 * (b) + (c) + 2) >> 2)

// One predicted sample of an n x n block (n = 8 or 4), modes 0,1,3..8 in the 8x8 numbering
// (the 4x4 twins 10,11,13..18 are the same formulas on a 4-wide block; MD.cs:1883-2774).
// nb(dy, dx) returns the already-reconstructed neighbour at block-relative (row dy, col dx):
// dy = -1 is the row above (dx may run to n+4 for mode 8), dx = -1 the column to the left.
template <class Nb>
MOBI_HD int mobi_pred_px(int mode, int n, int y, int x, int top_avail, int left_avail, Nb nb) {
#define T_(k) nb(-1, (k))
#define L_(i) nb((i), -1)
  switch (mode) {
    case 0: return T_(x);                                   // vertical           :1890 / :2475
    case 1: return L_(y);                                   // horizontal         :1903 / :2484
    case 3: {                                               // DC with availability :1920 / :2501
      if (!top_avail && !left_avail) return 0x80;
      int s = 0;
      if (top_avail) for (int k = 0; k < n; k++) s += T_(k);
      if (left_avail) for (int k = 0; k < n; k++) s += L_(k);
      return (top_avail && left_avail) ? (s + n) / (2 * n) : (s + n / 2) / n;
    }
    case 4: {                                               // horizontal-up      :2023 / :2581
      const int k = 2 * y + x;
      if (k >= 2 * n - 2) return L_(n - 1);
      const int j = k >> 1;
      if ((k & 1) == 0) return MOBI_F2(L_(j), L_(j + 1));
      return MOBI_F3(L_(j), L_(j + 1), L_(j + 2 < n ? j + 2 : n - 1));
    }
    case 5: {                                               // horizontal-down    :2091 / :2620
      const int k = x - 2 * y;
      if (k >= 3) return MOBI_F3(T_(k - 3), T_(k - 2), T_(k - 1));
      if (k == 2) return MOBI_F3(T_(-1), T_(0), T_(1));
      if (k == 1) return MOBI_F3(L_(0), T_(-1), T_(0));
      if (k == 0) return MOBI_F2(L_(0), T_(-1));
      if (k == -1) return MOBI_F3(T_(-1), L_(0), L_(1));
      const int j = (-k) >> 1;                              // k <= -2
      if (((-k) & 1) == 0) return MOBI_F2(L_(j - 1), L_(j));
      return MOBI_F3(L_(j - 1), L_(j), L_(j + 1));
    }
    case 6: {                                               // vertical-right     :2197 / :2656
      const int k = 2 * x - y;
      if (k >= 0 && (k & 1) == 0) { const int c = x - (y >> 1); return MOBI_F2(T_(c - 1), T_(c)); }
      if (k >= 0) { const int c = x - ((y + 1) >> 1); return MOBI_F3(T_(c - 1), T_(c), T_(c + 1)); }
      if (k == -1) return MOBI_F3(L_(0), T_(-1), T_(0));
      const int j = -k - 2;                                 // F3(L[j-1], L[j], L[j+1]) with L[-1] = top-left
      return MOBI_F3(j ? L_(j - 1) : T_(-1), L_(j), L_(j + 1));
    }
    case 7: {                                               // diagonal down-right :2291 / :2702
      const int k = x - y;
      if (k >= 2) return MOBI_F3(T_(k - 2), T_(k - 1), T_(k));
      if (k == 1) return MOBI_F3(T_(-1), T_(0), T_(1));
      if (k == 0) return MOBI_F3(L_(0), T_(-1), T_(0));
      if (k == -1) return MOBI_F3(T_(-1), L_(0), L_(1));
      return MOBI_F3(L_(-k - 2), L_(-k - 1), L_(-k));
    }
    case 8: {                                               // vertical-left      :2368 / :2734
      const int s = x + (y >> 1);
      return (y & 1) ? MOBI_F3(T_(s), T_(s + 1), T_(s + 2)) : MOBI_F2(T_(s), T_(s + 1));
    }
    default: return 0;
  }
#undef T_
#undef L_
}

// Plane predictors: 16x16 (sub_1167BC, MD.cs:3017), 8x8 (sub_116CCC, :3168), 4x4 (sub_117E98, :3253).
// Returns the packed word for samples (y, x0..x0+3).  The reference ORs un-clamped samples into the
// word, so a sample outside 0..255 bleeds into its neighbours; kept bit for bit.
template <class Nb>
MOBI_HD uint32_t mobi_plane_word(int n, int param, int y, int x0, Nb nb) {
  const int lg = (n == 4) ? 2 : 3, half = (n == 16);
  const int bl = nb(n - 1, -1), tr = nb(-1, n - 1);
  const int corner = ((bl + tr + 1) >> 1) + param * 2;
  const int cs = corner - bl + half, rs = corner - tr + half;
  const int l = nb(y, -1);
  const int r = (tr << lg) + (y + 1) * (half ? (rs >> 1) : rs);
  const int rstep0 = (r - (l << lg)) + half, rstep = half ? (rstep0 >> 1) : rstep0;
  const int rnd = (n == 4) ? 16 : 64, sh = (n == 4) ? 5 : 7;
  uint32_t word = 0;
  for (int k = 0; k < 4; k++) {
    const int x = x0 + k, t = nb(-1, x);
    const int b = (bl << lg) + (x + 1) * (half ? (cs >> 1) : cs);
    const int cstep0 = (b - (t << lg)) + half, cstep = half ? (cstep0 >> 1) : cstep0;
    const int acc = (t << (2 * lg)) + (y + 1) * cstep;
    const int v = (l << (2 * lg)) + (x + 1) * rstep;
    word |= (uint32_t)((acc + v + rnd) >> sh) << (8 * k);
  }
  return word;
}

#endif
