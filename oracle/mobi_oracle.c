/*
 * mobi_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see mobi_oracle.h).
 *
 * Plain-C restatement of LibMobiclip/Codec/Mobiclip/MobiclipDecoder.cs ("MD.cs").
 * Every function cites the MD.cs lines it follows.  C# semantics kept on purpose:
 *   - `uint` arithmetic wraps; shift counts are masked to 5 bits (SHL/SHR below) -- reachable
 *     in ReadVarInt* when the bit window is all zero (CLZ == 32), MD.cs:2972-2982;
 *   - `>>` on int is arithmetic, `/` truncates toward zero;
 *   - every managed array access is bounds checked and throws; the decoder swallows the
 *     exception (MD.cs:325) -> here a longjmp with an ORA_E_* code;
 *   - the `Internal[392]` word array (MD.cs:28) keeps the reference's exact layout, so every
 *     aliasing quirk (intra-mode byte cache in words 0..9, dequant words, coefficient block,
 *     IDCT scratch, VLC-table select, MV predictor and MV row cache) behaves identically.
 * PARITY UNPINNED by reference tests (there are none) -- see DESIGN.md.
 */
#include "mobi_oracle.h"
#include "mobi_tables.h"

#include <setjmp.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  uint8_t *p; /* NULL == C# null reference */
  long len;
} barr;

struct mobi_oracle {
  const uint8_t *Data; /* MD.cs:15 */
  long DataLen;
  int Offset; /* MD.cs:16 */
  uint32_t Width, Height;
  barr Y[6], UV[6]; /* MD.cs:19-20 */
  uint32_t Quantizer, YuvFormat;
  uint32_t Internal[392]; /* MD.cs:28 */
  int Stride;
  int Version;
  int nbr;     /* `ref int nrBitsRemaining` threaded through every C# call */
  uint32_t r3; /* `ref uint r3`: 32-bit MSB-first bit window */
  jmp_buf jb;
};
typedef struct mobi_oracle D;

#define THROW(d, code) longjmp((d)->jb, (code))
#define SHL(x, n) ((uint32_t)(x) << ((n) & 31))
#define SHR(x, n) ((uint32_t)(x) >> ((n) & 31))

static inline uint8_t RD(D *d, barr a, long i) {
  if (!a.p) THROW(d, ORA_E_NULLREF);
  if (i < 0 || i >= a.len) THROW(d, ORA_E_INDEX);
  return a.p[i];
}
static inline void WR(D *d, barr a, long i, uint32_t v) {
  if (!a.p) THROW(d, ORA_E_NULLREF);
  if (i < 0 || i >= a.len) THROW(d, ORA_E_INDEX);
  a.p[i] = (uint8_t)v;
}
/* IOUtil.ReadU32LE / WriteU32LE, Utils/IOUtil.cs:73-84 */
static inline uint32_t RD32(D *d, barr a, long i) {
  uint32_t b3 = RD(d, a, i + 3), b2 = RD(d, a, i + 2), b1 = RD(d, a, i + 1), b0 = RD(d, a, i);
  return (b3 << 24) | (b2 << 16) | (b1 << 8) | b0;
}
static inline void WR32(D *d, barr a, long i, uint32_t v) {
  WR(d, a, i, v);
  WR(d, a, i + 1, v >> 8);
  WR(d, a, i + 2, v >> 16);
  WR(d, a, i + 3, v >> 24);
}
static inline uint32_t *IN(D *d, long i) { /* Internal[i] with the managed bounds check */
  if (i < 0 || i >= 392) THROW(d, ORA_E_INDEX);
  return &d->Internal[i];
}
#define IBYTE(d) ((uint8_t *)(d)->Internal) /* `fixed (uint* ..) byte* InternalByte`, MD.cs:1837-1839 */

/* IOUtil.ReadU16LE, Utils/IOUtil.cs:39 */
static uint32_t data_u16(D *d, long off) {
  if (!d->Data) THROW(d, ORA_E_NULLREF);
  if (off + 1 < 0 || off + 1 >= d->DataLen) THROW(d, ORA_E_INDEX);
  if (off < 0) THROW(d, ORA_E_INDEX);
  return ((uint32_t)d->Data[off + 1] << 8) | d->Data[off];
}

/* ------------------------------------------------------------------ bit reader */
/* CLZ, MD.cs:3927 */
static int clz32(uint32_t v) {
  int n = 0;
  while (v) {
    v >>= 1;
    n++;
  }
  return 32 - n;
}
/* FillBits, MD.cs:2988-2996 */
static void fill_bits(D *d) {
  if (d->Offset >= d->DataLen) return;
  uint32_t w = data_u16(d, d->Offset);
  d->Offset += 2;
  d->nbr += 0x10;
  int sh = 0x10 - d->nbr;
  d->r3 |= SHL(w, sh);
}
#define TAKE(d, n)                   \
  do {                               \
    (d)->r3 = SHL((d)->r3, (n));     \
    (d)->nbr -= (n);                 \
    if ((d)->nbr < 0) fill_bits(d);  \
  } while (0)

/* ReadVarIntUnsigned, MD.cs:2970-2986 */
static uint32_t read_ue(D *d) {
  int z = clz32(d->r3);
  d->r3 = SHL(d->r3, z);
  d->r3 += d->r3;
  int sh = 0x20 - z;
  uint32_t v = (sh == 0x20) ? 0 : SHR(d->r3, sh);
  v += SHL(1u, z);
  v--;
  d->r3 = SHL(d->r3, z);
  d->nbr -= z << 1;
  if (--d->nbr < 0) fill_bits(d);
  return v;
}
/* ReadVarIntSigned, MD.cs:2998-3015 */
static int read_se(D *d) {
  int z = clz32(d->r3);
  d->r3 = SHL(d->r3, z);
  d->r3 += d->r3;
  int sh = 0x20 - z;
  int v = (sh == 0x20) ? 0 : (int)SHR(d->r3, sh);
  v = (int)((uint32_t)v + SHL(1u, z));
  if (v & 1) v = (int)(1u - (uint32_t)v);
  v >>= 1;
  d->r3 = SHL(d->r3, z);
  d->nbr -= z << 1;
  if (--d->nbr < 0) fill_bits(d);
  return v;
}

/* ------------------------------------------------------------------ dequant tables */
/* SetupQuantizationTables, MD.cs:3884-3925 */
static void setup_quant(D *d, uint32_t q) {
  if (d->Version == MOBI_VER_MOFLEX3DS) {
    if (q < 0xC) q = 0xC;
    if (q > 0x34) q = 0x34;
  }
  d->Quantizer = q;
  if (q >= sizeof(mobi_qdiv6)) THROW(d, ORA_E_INDEX); /* byte_119004[quantizer] */
  int sh = mobi_qdiv6[q] + 8;
  int m = mobi_qmod6[q];
  for (int i = 0; i < 16; i++) d->Internal[74 + i] = (uint32_t)mobi_zz4[i] | SHL((uint32_t)mobi_dq4[(m << 4) + i], sh);
  sh -= 2;
  for (int i = 0; i < 64; i++) d->Internal[10 + i] = (uint32_t)mobi_zz8[i] | SHL((uint32_t)mobi_dq8[(m << 6) + i], sh);
  uint8_t *ib = IBYTE(d);
  ib[1] = ib[2] = ib[3] = ib[4] = 9;
  ib[8] = ib[0x10] = ib[0x18] = ib[0x20] = 9;
}

/* ------------------------------------------------------------------ motion compensation */
/* CopyBlock, MD.cs:418-456 (row buffer `pixels`, then Array.Copy into Dst) */
static void copy_block(D *d, barr Src, int Dx, int Dy, uint32_t W, uint32_t H, barr Dst, int Offset) {
  const int S = d->Stride;
  uint8_t pixels[16];
  for (uint32_t i = 0; i < H; i++) {
    long pos = (long)Offset + (long)((Dy >> 1) + (int)i) * S + (Dx >> 1);
    switch ((Dx & 1) | ((Dy & 1) << 1)) {
      case 0:
        if (!Src.p) THROW(d, ORA_E_NULLREF);
        if (pos < 0 || pos + (long)W > Src.len) THROW(d, ORA_E_INDEX); /* Array.Copy range check */
        memcpy(pixels, Src.p + pos, W);
        break;
      case 1:
        for (uint32_t j = 0; j < W; j++) pixels[j] = (uint8_t)((RD(d, Src, pos + j) >> 1) + (RD(d, Src, pos + j + 1) >> 1));
        break;
      case 2:
        for (uint32_t j = 0; j < W; j++) pixels[j] = (uint8_t)((RD(d, Src, pos + j) >> 1) + (RD(d, Src, pos + j + S) >> 1));
        break;
      case 3:
        for (uint32_t j = 0; j < W; j++)
          pixels[j] = (uint8_t)((((RD(d, Src, pos + j) >> 1) + (RD(d, Src, pos + j + 1) >> 1)) >> 1) +
                                (((RD(d, Src, pos + j + S) >> 1) + (RD(d, Src, pos + j + 1 + S) >> 1)) >> 1));
        break;
    }
    long o = (long)Offset + (long)i * S;
    if (!Dst.p) THROW(d, ORA_E_NULLREF);
    if (o < 0 || o + (long)W > Dst.len) THROW(d, ORA_E_INDEX);
    memcpy(Dst.p + o, pixels, W);
  }
}

/* ---- syntax coverage counters (tests/test_coverage.py): which parts of the syntax has this THREAD decoded so far (thread-local: shared counters made 256 decoder threads fight over one cache line)?  Test-side
 * accounting only: they change no result.  Layout in mobi_oracle.h (MOBI_COV_*). ---- */
static __thread uint64_t g_cov[MOBI_COV_WORDS];
#define COV(i) (g_cov[(i)]++)
void mobi_oracle_coverage(uint64_t *out, int reset) {
  if (out) memcpy(out, g_cov, sizeof(g_cov));
  if (reset) memset(g_cov, 0, sizeof(g_cov));
}

/* loc_1147B0 / loc_114A64 / loc_114CAC / loc_114ED4 (MD.cs:409,592,692,795): one MC leaf, width w */
static void mc_leaf(D *d, int io, uint32_t srcFrame, uint32_t w, uint32_t h, int dx, int dy, int Offset) {
  *IN(d, io) = (uint32_t)dx;
  *IN(d, io + 1) = (uint32_t)dy;
  uint32_t f = srcFrame / 4;
  if (f >= 1 && f <= 5) COV(MOBI_COV_REF + f - 1);
  COV(MOBI_COV_PHASE + ((dx & 1) | ((dy & 1) << 1)));
  copy_block(d, d->Y[f], dx, dy, w, h, d->Y[0], Offset);
  copy_block(d, d->UV[f], dx >> 1, dy >> 1, w >> 1, h >> 1, d->UV[0], Offset / 2);
  copy_block(d, d->UV[f], dx >> 1, dy >> 1, w >> 1, h >> 1, d->UV[0], Offset / 2 + d->Stride / 2);
}
/* sub_114790 / sub_114A44 / sub_114C8C / sub_114EB4 (MD.cs:400,583,683,786) */
static void mc_leaf_mvd(D *d, int io, uint32_t srcFrame, uint32_t w, uint32_t h, int Offset) {
  int dx = read_se(d);
  int dy = read_se(d);
  dx += (int)d->Internal[219];
  dy += (int)d->Internal[220];
  mc_leaf(d, io, srcFrame, w, h, dx, dy, Offset);
}

static void dec_intra_full(D *d, int Offset);
static void dec_intra_sub(D *d, int Offset);
static void p_residual(D *d, int Offset);

/* ReadPBlock{W}x{H} + SwitchPBlock{W}x{H}, MD.cs:469-1746.  The sixteen C# functions are
 * isomorphic: LUT on the top bits -> code; bit-count table; then
 *   0: leaf with the predicted MV, ref slot 1;  1..5: se,se MV delta, ref slot = code;
 *   6,7: intra MB (16x16 only, else throw);  8: top/bottom halves (throw when h==2);
 *   9: left/right halves (throw when w==2).  wi/hi = log2(16/w), log2(16/h). */
static void pblock(D *d, int wi, int hi, int io, int Offset) {
  int ver;
  if (d->Version == MOBI_VER_MOFLEX3DS) ver = 0;
  else if (d->Version == MOBI_VER_MODSDS) ver = 1;
  else return;
  const int s = wi * 4 + hi;
  const uint32_t w = 16u >> wi, h = 16u >> hi;
  uint32_t code = mobi_part_lut[ver][s][d->r3 >> mobi_part_shift[ver][s]];
  if (code >= mobi_part_nbits_len[ver][s]) THROW(d, ORA_E_INDEX); /* bit-count table shorter than code */
  int nb = mobi_part_bits[ver][s][code];
  TAKE(d, nb);
  if (code < 10) COV(MOBI_COV_PART + (ver * 16 + s) * 10 + code);
  switch (code) {
    case 0:
      mc_leaf(d, io, 4, w, h, (int)d->Internal[219], (int)d->Internal[220], Offset);
      break;
    case 1: case 2: case 3: case 4: case 5:
      mc_leaf_mvd(d, io, code * 4, w, h, Offset);
      break;
    case 6:
      if (s != 0) THROW(d, ORA_E_PARTCODE);
      dec_intra_full(d, Offset);
      return; /* MD.cs:509-518: no loc_1161A0 after intra MBs */
    case 7:
      if (s != 0) THROW(d, ORA_E_PARTCODE);
      dec_intra_sub(d, Offset);
      return;
    case 8:
      if (h == 2) THROW(d, ORA_E_PARTCODE);
      pblock(d, wi, hi + 1, io, Offset);
      pblock(d, wi, hi + 1, io, Offset + d->Stride * (int)(h / 2));
      break;
    case 9:
      if (w == 2) THROW(d, ORA_E_PARTCODE);
      pblock(d, wi + 1, hi, io, Offset);
      pblock(d, wi + 1, hi, io, Offset + (int)(w / 2));
      break;
    default:
      break;
  }
  if (s == 0) p_residual(d, Offset); /* MD.cs:476,482,...,525,534 */
}

/* ------------------------------------------------------------------ residual VLC + IDCT */
/* ReadDCTMatrix, MD.cs:3330-3432.  r12 walks the dequant words inside Internal[]. */
static void read_dct(D *d, uint32_t *pr12) {
  const uint16_t *A = (d->Internal[218] == 1) ? mobi_vx2table1_a : mobi_vx2table0_a;
  const uint8_t *B = (d->Internal[218] == 1) ? mobi_vx2table1_b : mobi_vx2table0_b;
  uint32_t r12 = *pr12;
  COV(MOBI_COV_VLCTAB + (d->Internal[218] == 1));
  for (;;) {
    int skip, value, nb, t;
    uint32_t e = d->r3 >> 25, r8;
    if (e == 3) {
      d->r3 <<= 7;
      int c = (d->r3 >> 31) == 1;
      d->r3 <<= 1;
      COV(MOBI_COV_ESCAPE + (!c ? 0 : ((d->r3 >> 31) == 1 ? 2 : 1)));
      if (!c) { /* escape 0: level offset */
        d->nbr -= 8;
        if (d->nbr < 0) fill_bits(d);
        e = A[d->r3 >> 20];
        t = B[e >> 9]; /* index < 128 */
        nb = (int)(e & 0xF);
        e >>= 4;
        value = (int)(e & 0x1F) + t;
        e >>= 5;
        d->r3 = SHL(d->r3, nb - 1);
        if (((d->r3 >> 31) & 1) == 1) value = -value;
        d->r3 <<= 1;
        d->nbr -= nb;
        if (d->nbr < 0) fill_bits(d);
        skip = (int)(e & 0x3F);
        e >>= 6;
      } else {
        c = (d->r3 >> 31) == 1;
        d->r3 <<= 1;
        if (!c) { /* escape 10: run offset */
          d->nbr -= 9;
          if (d->nbr < 0) fill_bits(d);
          e = A[d->r3 >> 20];
          nb = (int)(e & 0xF);
          e >>= 4;
          value = (int)(e & 0x1F);
          e >>= 5;
          r8 = e & 0x3F;
          e >>= 6;
          t = B[0x80 + value + (e << 6)]; /* <= 0x80+31+64 < 256 */
          d->r3 = SHL(d->r3, nb - 1);
          if (((d->r3 >> 31) & 1) == 1) value = -value;
          d->r3 <<= 1;
          d->nbr -= nb;
          if (d->nbr < 0) fill_bits(d);
          skip = (int)r8 + t;
        } else { /* escape 11: raw last/run/level */
          d->nbr -= 9;
          if (d->nbr < 0) fill_bits(d);
          e = d->r3 >> 31;
          d->r3 <<= 1;
          skip = (int)(d->r3 >> 26);
          d->r3 <<= 6;
          d->nbr -= 7;
          if (d->nbr < 0) fill_bits(d);
          value = (int32_t)d->r3 >> 20;
          d->r3 <<= 12;
          d->nbr -= 12;
          if (d->nbr < 0) fill_bits(d);
        }
      }
    } else {
      e = A[d->r3 >> 20];
      nb = (int)(e & 0xF);
      e >>= 4;
      value = (int)(e & 0x1F);
      e >>= 5;
      d->r3 = SHL(d->r3, nb - 1);
      if (((d->r3 >> 31) & 1) == 1) value = -value;
      d->r3 <<= 1;
      d->nbr -= nb;
      if (d->nbr < 0) fill_bits(d);
      skip = (int)(e & 0x3F);
      e >>= 6;
    }
    r12 = (uint32_t)(r12 + (uint32_t)skip);
    r8 = *IN(d, (long)r12);
    r12++;
    int zz = (int)(r8 & 0xFF);
    int sc = (int)(r8 >> 8);
    *IN(d, 90 + zz) = (uint32_t)(sc * value);
    if (e & 1) break;
  }
  *pr12 = r12;
}

static inline void add_clamp(D *d, barr Dst, long o, int v) { /* MinMaxTable[0x40 + Dst[o] + v] */
  int idx = 0x40 + (int)RD(d, Dst, o) + v;
  if (idx < 0 || idx >= 384) THROW(d, ORA_E_INDEX);
  WR(d, Dst, o, mobi_vx2minmaxtable[idx]);
}

/* 8-point butterfly shared by both passes of IDCT64Px8, MD.cs:3452-3485 / :3517-3550 */
static void bfly8(const int in[8], int out[8]) {
  int r0 = in[0], r1 = in[1], r2 = in[2], r3 = in[3], r4 = in[4], r5 = in[5], r6 = in[6], r7 = in[7], r8, r9;
  r8 = r0 + r4;
  r9 = r0 - r4;
  r0 = r2 + (r6 >> 1);
  r4 = (r2 >> 1) - r6;
  r2 = r9 + r4;
  r4 = r9 - r4;
  r6 = r8 - r0;
  r0 = r8 + r0;
  r8 = r1 + r7;
  r8 -= r3;
  r8 -= (r3 >> 1);
  r9 = r7 - r1;
  r9 += r5;
  r9 += (r5 >> 1);
  r7 += (r7 >> 1);
  r7 = r5 - r7;
  r7 -= r3;
  r3 += r5;
  r3 += r1;
  r3 += (r1 >> 1);
  r1 = r7 + (r3 >> 2);
  r7 = r3 - (r7 >> 2);
  r3 = r8 + (r9 >> 2);
  r5 = (r8 >> 2) - r9;
  r0 += r7;
  r7 = r0 - r7 * 2;
  r8 = r2 + r5;
  r9 = r2 - r5;
  r2 = r4 + r3;
  r5 = r4 - r3;
  r3 = r6 + r1;
  r4 = r6 - r1;
  out[0] = r0; out[1] = r8; out[2] = r2; out[3] = r3; out[4] = r4; out[5] = r5; out[6] = r9; out[7] = r7;
}
/* IDCT64Px8, MD.cs:3435-3561 */
static void idct64p8(D *d, barr Dst, int Offset) {
  uint32_t *I = d->Internal;
  int in[8], out[8];
  for (int k = 0; k < 8; k++) {
    for (int m = 0; m < 8; m++) in[m] = (int)I[90 + 8 * k + m];
    if (k == 0) in[0] += 0x20;
    bfly8(in, out);
    for (int m = 0; m < 8; m++) I[154 + 8 * m + k] = (uint32_t)out[m];
  }
  for (int i = 0; i < 8; i++) {
    for (int m = 0; m < 8; m++) in[m] = (int)I[154 + 8 * i + m];
    bfly8(in, out);
    for (int j = 0; j < 8; j++) add_clamp(d, Dst, (long)Offset + j, out[j] >> 6);
    Offset += d->Stride;
  }
}
/* 4-input reduced butterfly of IDCT16Px8, MD.cs:3577-3600 / :3624-3647 */
static void bfly8_4(int r0, int r1, int r2, int r3, int out[8]) {
  int r4 = r0 - (r2 >> 1);
  int r6 = r0 - r2;
  int r9 = r0 + (r2 >> 1);
  r0 += r2;
  int r8 = r1 - r3;
  r8 -= (r3 >> 1);
  int r7 = r3 + r1;
  r7 += (r1 >> 1);
  r2 = -r3;
  int r5 = r1 + (r8 >> 2);
  r3 = -r1;
  r3 = r8 + (r3 >> 2);
  r1 = r2 + (r7 >> 2);
  r7 -= (r2 >> 2);
  r0 += r7;
  r7 = r0 - r7 * 2;
  r8 = r9 + r5;
  r9 -= r5;
  r2 = r4 + r3;
  r5 = r4 - r3;
  r3 = r6 + r1;
  r4 = r6 - r1;
  out[0] = r0; out[1] = r8; out[2] = r2; out[3] = r3; out[4] = r4; out[5] = r5; out[6] = r9; out[7] = r7;
}
/* IDCT16Px8, MD.cs:3564-3658: only the top-left 4x4 of the coefficient block is read */
static void idct16p8(D *d, barr Dst, int Offset) {
  uint32_t *I = d->Internal;
  int out[8];
  for (int k = 0; k < 4; k++) {
    int r0 = (int)I[90 + 8 * k], r1 = (int)I[91 + 8 * k], r2 = (int)I[92 + 8 * k], r3 = (int)I[93 + 8 * k];
    if (k == 0) r0 += 0x20;
    bfly8_4(r0, r1, r2, r3, out);
    for (int m = 0; m < 8; m++) I[154 + 4 * m + k] = (uint32_t)out[m];
  }
  for (int i = 0; i < 8; i++) {
    bfly8_4((int)I[154 + 4 * i], (int)I[155 + 4 * i], (int)I[156 + 4 * i], (int)I[157 + 4 * i], out);
    for (int j = 0; j < 8; j++) add_clamp(d, Dst, (long)Offset + j, out[j] >> 6);
    Offset += d->Stride;
  }
}
/* IDCT3Px8, MD.cs:3661-3707: coefficients 0, 1 and 8 only (and it overwrites Internal[90..97]) */
static void idct3p8(D *d, barr Dst, int Offset) {
  uint32_t *I = d->Internal;
  int r8 = (int)I[90], r9 = (int)I[91], r10 = (int)I[98];
  r8 += 32;
  int r7 = r9 + (r9 >> 1);
  int r11 = r7 >> 2;
  int r3 = -r9;
  r3 = r9 + (r3 >> 2);
  int r5 = r9 + (r9 >> 2);
  I[90] = (uint32_t)(r8 + r7);
  I[97] = (uint32_t)(r8 - r7);
  I[91] = (uint32_t)(r8 + r5);
  I[96] = (uint32_t)(r8 - r5);
  I[92] = (uint32_t)(r8 + r3);
  I[95] = (uint32_t)(r8 - r3);
  I[93] = (uint32_t)(r8 + r11);
  I[94] = (uint32_t)(r8 - r11);
  r7 = r10 + (r10 >> 1);
  int r1 = r7 >> 2;
  r3 = -r10;
  r3 = r10 + (r3 >> 2);
  r5 = r10 + (r10 >> 2);
  for (int i = 0; i < 8; i++) {
    int r0 = (int)I[90 + i];
    add_clamp(d, Dst, (long)Offset + 0, (r0 + r7) >> 6);
    add_clamp(d, Dst, (long)Offset + 1, (r0 + r5) >> 6);
    add_clamp(d, Dst, (long)Offset + 2, (r0 + r3) >> 6);
    add_clamp(d, Dst, (long)Offset + 3, (r0 + r1) >> 6);
    add_clamp(d, Dst, (long)Offset + 4, (r0 - r1) >> 6);
    add_clamp(d, Dst, (long)Offset + 5, (r0 - r3) >> 6);
    add_clamp(d, Dst, (long)Offset + 6, (r0 - r5) >> 6);
    add_clamp(d, Dst, (long)Offset + 7, (r0 - r7) >> 6);
    Offset += d->Stride;
  }
}
/* IDCT1Px8 / IDCT1Px4, MD.cs:3710-3725 / :3787-3798 */
static void idct1p(D *d, barr Dst, int Offset, int n) {
  int r9 = ((int)d->Internal[90] + 32) >> 6;
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) add_clamp(d, Dst, (long)Offset + j, r9);
    Offset += d->Stride;
  }
}
static void bfly4(int r0, int r1, int r2, int r3, int out[4]) { /* MD.cs:3740-3747 / :3768-3775 */
  r0 += r2;
  r2 = r0 - r2 * 2;
  int r8 = (r1 >> 1) - r3;
  int r9 = r1 + (r3 >> 1);
  r3 = r0 - r9;
  r0 += r9;
  r1 = r2 + r8;
  r2 -= r8;
  out[0] = r0; out[1] = r1; out[2] = r2; out[3] = r3;
}
/* IDCT16Px4, MD.cs:3728-3784 */
static void idct16p4(D *d, barr Dst, int Offset) {
  uint32_t *I = d->Internal;
  int out[4];
  for (int k = 0; k < 4; k++) {
    int r0 = (int)I[90 + 4 * k];
    if (k == 0) r0 += 0x20;
    bfly4(r0, (int)I[91 + 4 * k], (int)I[92 + 4 * k], (int)I[93 + 4 * k], out);
    for (int m = 0; m < 4; m++) I[106 + 4 * m + k] = (uint32_t)out[m];
  }
  for (int i = 0; i < 4; i++) {
    bfly4((int)I[106 + 4 * i], (int)I[107 + 4 * i], (int)I[108 + 4 * i], (int)I[109 + 4 * i], out);
    for (int j = 0; j < 4; j++) add_clamp(d, Dst, (long)Offset + j, out[j] >> 6);
    Offset += d->Stride;
  }
}

/* loc_116540, MD.cs:2931-2943: one 8x8 residual block */
static void resid8(D *d, barr Dst, int Offset) {
  for (int i = 0; i < 64; i++) d->Internal[90 + i] = 0;
  uint32_t r12 = 10;
  read_dct(d, &r12);
  COV(MOBI_COV_IDCT + (r12 <= 11 ? 0 : r12 <= 13 ? 1 : r12 <= 20 ? 2 : 3));
  if (r12 <= 11) idct1p(d, Dst, Offset, 8);
  else if (r12 <= 13) idct3p8(d, Dst, Offset);
  else if (r12 <= 20) idct16p8(d, Dst, Offset);
  else idct64p8(d, Dst, Offset);
}
/* sub_1166E8, MD.cs:2958-2968 (also the tail of loc_116628 :2948-2955): one 4x4 residual block */
static void resid4(D *d, barr Dst, int Offset) {
  for (int i = 0; i < 16; i++) d->Internal[90 + i] = 0;
  uint32_t r12 = 74;
  read_dct(d, &r12);
  COV(MOBI_COV_IDCT + (r12 <= 75 ? 4 : 5));
  if (r12 <= 75) idct1p(d, Dst, Offset, 4);
  else idct16p4(d, Dst, Offset);
}
/* loc_11652C, MD.cs:2909-2929 */
static void resid8_or_4x4(D *d, barr Dst, int Offset) {
  if (((d->r3 >> 31) & 1) == 1) {
    d->r3 += d->r3;
    d->nbr--;
    resid8(d, Dst, Offset);
  } else {
    uint32_t ue = read_ue(d);
    if (ue >= sizeof(mobi_cbp4_inter)) THROW(d, ORA_E_INDEX);
    uint32_t m = mobi_cbp4_inter[ue];
    if (m & 1) resid4(d, Dst, Offset);
    Offset += 4;
    if ((m >> 1) & 1) resid4(d, Dst, Offset);
    Offset += d->Stride * 4;
    Offset -= 4;
    if ((m >> 2) & 1) resid4(d, Dst, Offset);
    Offset += 4;
    if ((m >> 3) & 1) resid4(d, Dst, Offset);
  }
}
/* loc_1161A0, MD.cs:1818-1833 */
static void p_residual(D *d, int Offset) {
  uint32_t ue = read_ue(d);
  if (ue >= sizeof(mobi_cbp_inter)) THROW(d, ORA_E_INDEX);
  uint32_t m = mobi_cbp_inter[ue];
  const int S = d->Stride;
  if (m & 1) resid8_or_4x4(d, d->Y[0], Offset);
  Offset += 8;
  if ((m >> 1) & 1) resid8_or_4x4(d, d->Y[0], Offset);
  Offset += S * 8;
  Offset -= 8;
  if ((m >> 2) & 1) resid8_or_4x4(d, d->Y[0], Offset);
  Offset += 8;
  if ((m >> 3) & 1) resid8_or_4x4(d, d->Y[0], Offset);
  Offset -= S * 8;
  Offset -= 8;
  if ((m >> 4) & 1) resid8_or_4x4(d, d->UV[0], Offset / 2);
  if ((m >> 5) & 1) resid8_or_4x4(d, d->UV[0], Offset / 2 + S / 2);
}

/* ------------------------------------------------------------------ intra prediction */
#define F2(a, b) (((a) + (b) + 1) >> 1)
#define F3(a, b, c) (((a) + 2 * (b) + (c) + 2) >> 2)

/* plane predictors: sub_1167BC (16x16, MD.cs:3017), sub_116CCC (8x8, :3168), sub_117E98 (4x4, :3253).
 * The packed-word stores are kept: a sample outside 0..255 bleeds into the neighbouring bytes
 * of its 4-pixel word exactly like `r5 |= (r12 << 8)` does. */
static void plane_pred(D *d, barr Dst, int Offset, int n, int param) {
  const int S = d->Stride;
  COV(MOBI_COV_PLANE + (n == 16 ? 0 : n == 8 ? 1 : 2));
  int T[16], acc[16], step[16];
  if (n == 4) {
    uint32_t w = RD32(d, Dst, (long)Offset - S);
    for (int i = 0; i < 4; i++) T[i] = (int)((w >> (8 * i)) & 0xFF);
  } else {
    if (!Dst.p) THROW(d, ORA_E_NULLREF);
    if ((long)Offset - S < 0 || (long)Offset - S + n > Dst.len) THROW(d, ORA_E_INDEX); /* Array.Copy */
    for (int i = 0; i < n; i++) T[i] = Dst.p[Offset - S + i];
  }
  int bl = RD(d, Dst, (long)Offset + (long)S * (n - 1) - 1); /* bottom of the left column */
  int tr = T[n - 1];
  int corner = ((bl + tr + 1) >> 1) + param * 2;
  const int half = (n == 16); /* 16x16 uses half steps with a +1 bias, MD.cs:3026-3037 */
  const int lg = (n == 4) ? 2 : 3;   /* 4x4 works at <<2 / <<4, the others at <<3 / <<6 */
  int cs = corner - bl + half, b = bl << lg;
  for (int i = 0; i < n; i++) {
    b += half ? (cs >> 1) : cs;
    acc[i] = T[i] << (2 * lg);
    step[i] = (b - (T[i] << lg)) + half;
  }
  int rs = corner - tr + half, r = tr << lg;
  const int rnd = (n == 4) ? 16 : 64, sh = (n == 4) ? 5 : 7;
  for (int y = 0; y < n; y++) {
    r += half ? (rs >> 1) : rs;
    int l = RD(d, Dst, (long)Offset - 1);
    int rstep = (r - (l << lg)) + half;
    int v = l << (2 * lg);
    for (int x0 = 0; x0 < n; x0 += 4) {
      uint32_t word = 0;
      for (int k = 0; k < 4; k++) {
        int x = x0 + k;
        /* column accumulator advances once per row (MD.cs:3055-3062), row accumulator once per pixel */
        int a = acc[x] + (half ? (step[x] >> 1) : step[x]);
        acc[x] = a;
        v += half ? (rstep >> 1) : rstep;
        word |= (uint32_t)((a + v + rnd) >> sh) << (8 * k);
      }
      WR32(d, Dst, (long)Offset + x0, word);
    }
    Offset += S;
  }
}

/* PredictIntra, MD.cs:1883-2774 (modes 2 and 12 read `se` and are dispatched by the caller) */
static void predict_intra(D *d, uint32_t mode, barr Dst, int Offset) {
  const int S = d->Stride;
  if (mode < 20) COV(MOBI_COV_INTRA + mode);
  int vfix = (Dst.p == d->UV[0].p && (Offset % S) >= S / 2); /* MD.cs:1886 */
  int T[16], L[8], TL, px[8][8];
#define RT(k) ((int)RD(d, Dst, (long)Offset - S + (k)))
#define RL(i) ((int)RD(d, Dst, (long)Offset + (long)(i) * S - 1))
  int n = 8;
  switch (mode) {
    case 0: { /* vertical, :1890 */
      if (!Dst.p) THROW(d, ORA_E_NULLREF);
      if ((long)Offset - S < 0 || (long)Offset - S + 8 > Dst.len) THROW(d, ORA_E_INDEX);
      for (int j = 0; j < 8; j++) T[j] = Dst.p[Offset - S + j];
      for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) px[i][j] = T[j];
      break;
    }
    case 1: /* horizontal, :1903 -- reads and writes are interleaved row by row */
      for (int i = 0; i < 8; i++) {
        int l = RL(i);
        for (int j = 0; j < 8; j++) WR(d, Dst, (long)Offset + (long)i * S + j, (uint32_t)l);
      }
      return;
    case 2: {
      int p = read_se(d); /* sub_116CCC reads its parameter first, :3170 */
      plane_pred(d, Dst, Offset, 8, p);
      return;
    }
    case 3:    /* DC 8x8, :1920 */
    case 13: { /* DC 4x4, :2501 */
      n = (mode == 3) ? 8 : 4;
      int r8 = 0;
      if (((Offset - (vfix ? (S / 2) : 0)) % S) != 0) r8 += 8;
      if (Offset >= S) r8 += 4;
      uint32_t dc = 0x80;
      if (r8 / 4 == 1 || r8 / 4 == 3) {
        uint32_t s = 0;
        for (int q = 0; q < n; q += 4) {
          uint32_t w = RD32(d, Dst, (long)Offset - S + q);
          s += (w >> 24) + ((w >> 16) & 0xFF) + ((w >> 8) & 0xFF) + (w & 0xFF);
        }
        dc = s;
      }
      if (r8 / 4 == 2) dc = 0;
      if (r8 / 4 >= 2)
        for (int i = 0; i < n; i++) dc += (uint32_t)RL(i);
      if (r8 / 4 == 1 || r8 / 4 == 2) dc = (dc + n / 2) / (uint32_t)n;
      if (r8 / 4 == 3) dc = (dc + n) / (uint32_t)(2 * n);
      for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) WR(d, Dst, (long)Offset + (long)i * S + j, dc);
      return;
    }
    case 4: { /* horizontal-up, :2023: pixel(y,x) = Z[2y+x] */
      for (int i = 0; i < 8; i++) L[i] = RL(i);
      int Z[24];
      for (int j = 0; j < 7; j++) {
        Z[2 * j] = F2(L[j], L[j + 1]);
        Z[2 * j + 1] = F3(L[j], L[j + 1], (j + 2 < 8) ? L[j + 2] : L[7]);
      }
      for (int k = 14; k < 24; k++) Z[k] = L[7];
      for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) px[y][x] = Z[2 * y + x];
      break;
    }
    case 5: { /* horizontal-down, :2091: pixel(y,x) = W[x-2y] */
      uint32_t a = RD32(d, Dst, (long)Offset - S), b = RD32(d, Dst, (long)Offset - S + 4);
      for (int k = 0; k < 4; k++) { T[k] = (a >> (8 * k)) & 0xFF; T[4 + k] = (b >> (8 * k)) & 0xFF; }
      L[0] = RL(0);
      TL = RT(-1);
      for (int i = 1; i < 8; i++) L[i] = RL(i);
      int Wb[24], *W = Wb + 15; /* W[-14..7] */
      W[0] = F2(L[0], TL);
      W[1] = F3(L[0], TL, T[0]);
      W[2] = F3(TL, T[0], T[1]);
      for (int k = 3; k <= 7; k++) W[k] = F3(T[k - 3], T[k - 2], T[k - 1]);
      W[-1] = F3(TL, L[0], L[1]);
      for (int j = 1; j <= 7; j++) {
        W[-2 * j] = F2(L[j - 1], L[j]);
        if (j < 7) W[-(2 * j + 1)] = F3(L[j - 1], L[j], L[j + 1]);
      }
      for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) px[y][x] = W[x - 2 * y];
      break;
    }
    case 6: { /* vertical-right, :2197 (left column read down to row 6 only) */
      uint32_t a = RD32(d, Dst, (long)Offset - S), b = RD32(d, Dst, (long)Offset - S + 4);
      for (int k = 0; k < 4; k++) { T[k] = (a >> (8 * k)) & 0xFF; T[4 + k] = (b >> (8 * k)) & 0xFF; }
      TL = RT(-1);
      for (int i = 0; i < 7; i++) L[i] = RL(i);
      /* E[k] = value on even diagonals, O[k] on odd ones; index k = 2x - y */
      for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) {
          int k = 2 * x - y, v;
#define TT(i) ((i) == -1 ? TL : T[(i)])
          if (k >= 0 && (k & 1) == 0) v = F2(TT(x - (y >> 1) - 1), TT(x - (y >> 1)));
          else if (k >= 0) {
            int c = x - ((y + 1) >> 1); /* c >= 0 here */
            v = F3(TT(c - 1), TT(c), TT(c + 1));
          } else if (k == -1) v = F3(L[0], TL, T[0]);
          else { /* k <= -2: F3(L[j-1], L[j], L[j+1]) with j = -k-2 and L[-1] := TL */
            int j = -k - 2;
            v = F3(j ? L[j - 1] : TL, L[j], L[j + 1]);
          }
#undef TT
          px[y][x] = v;
        }
      break;
    }
    case 7: { /* diagonal down-right, :2291: pixel(y,x) = Dg[x-y] */
      uint32_t a = RD32(d, Dst, (long)Offset - S);
      L[0] = RL(0);
      TL = RT(-1);
      uint32_t b = RD32(d, Dst, (long)Offset - S + 4);
      for (int k = 0; k < 4; k++) { T[k] = (a >> (8 * k)) & 0xFF; T[4 + k] = (b >> (8 * k)) & 0xFF; }
      for (int i = 1; i < 8; i++) L[i] = RL(i);
      int Db[16], *Dg = Db + 7; /* Dg[-7..7] */
      Dg[0] = F3(L[0], TL, T[0]);
      Dg[1] = F3(TL, T[0], T[1]);
      for (int k = 2; k <= 7; k++) Dg[k] = F3(T[k - 2], T[k - 1], T[k]);
      Dg[-1] = F3(TL, L[0], L[1]);
      for (int j = 2; j <= 7; j++) Dg[-j] = F3(L[j - 2], L[j - 1], L[j]);
      for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) px[y][x] = Dg[x - y];
      break;
    }
    case 8: { /* vertical-left, :2368: reads the top row out to +12 */
      uint32_t a = RD32(d, Dst, (long)Offset - S), b = RD32(d, Dst, (long)Offset - S + 4), c = RD32(d, Dst, (long)Offset - S + 8);
      int TT[13];
      for (int k = 0; k < 4; k++) { TT[k] = (a >> (8 * k)) & 0xFF; TT[4 + k] = (b >> (8 * k)) & 0xFF; TT[8 + k] = (c >> (8 * k)) & 0xFF; }
      TT[12] = RT(12);
      for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) {
          int s = x + (y >> 1);
          px[y][x] = (y & 1) ? F3(TT[s], TT[s + 1], TT[s + 2]) : F2(TT[s], TT[s + 1]);
        }
      break;
    }
    case 9: return;
    case 10: { /* 4x4 vertical, :2475 */
      uint32_t w = RD32(d, Dst, (long)Offset - S);
      for (int i = 0; i < 4; i++) WR32(d, Dst, (long)Offset + (long)i * S, w);
      return;
    }
    case 11: /* 4x4 horizontal, :2484 */
      for (int i = 0; i < 4; i++) {
        uint32_t l = (uint32_t)RL(i);
        WR32(d, Dst, (long)Offset + (long)i * S, l | (l << 8) | (l << 16) | (l << 24));
      }
      return;
    case 12: {
      int p = read_se(d); /* sub_117E98, :3255 */
      plane_pred(d, Dst, Offset, 4, p);
      return;
    }
    case 14: { /* 4x4 horizontal-up, :2581 */
      n = 4;
      for (int i = 0; i < 4; i++) L[i] = RL(i);
      int Z[12];
      for (int j = 0; j < 3; j++) {
        Z[2 * j] = F2(L[j], L[j + 1]);
        Z[2 * j + 1] = F3(L[j], L[j + 1], (j + 2 < 4) ? L[j + 2] : L[3]);
      }
      for (int k = 6; k < 12; k++) Z[k] = L[3];
      for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) px[y][x] = Z[2 * y + x];
      break;
    }
    case 15: { /* 4x4 horizontal-down, :2620 */
      n = 4;
      TL = RT(-1);
      L[0] = RL(0);
      uint32_t a = RD32(d, Dst, (long)Offset - S);
      for (int k = 0; k < 4; k++) T[k] = (a >> (8 * k)) & 0xFF;
      for (int i = 1; i < 4; i++) L[i] = RL(i);
      int Wb[12], *W = Wb + 7; /* W[-6..3] */
      W[0] = F2(L[0], TL);
      W[1] = F3(L[0], TL, T[0]);
      W[2] = F3(TL, T[0], T[1]);
      W[3] = F3(T[0], T[1], T[2]);
      W[-1] = F3(TL, L[0], L[1]);
      for (int j = 1; j <= 3; j++) {
        W[-2 * j] = F2(L[j - 1], L[j]);
        if (j < 3) W[-(2 * j + 1)] = F3(L[j - 1], L[j], L[j + 1]);
      }
      for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) px[y][x] = W[x - 2 * y];
      break;
    }
    case 16: { /* 4x4 vertical-right, :2656 (left column down to row 2) */
      n = 4;
      uint32_t a = RD32(d, Dst, (long)Offset - S);
      TL = RT(-1);
      for (int k = 0; k < 4; k++) T[k] = (a >> (8 * k)) & 0xFF;
      for (int i = 0; i < 3; i++) L[i] = RL(i);
      for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) {
          int k = 2 * x - y, v;
#define TT(i) ((i) == -1 ? TL : T[(i)])
          if (k >= 0 && (k & 1) == 0) v = F2(TT(x - (y >> 1) - 1), TT(x - (y >> 1)));
          else if (k >= 0) {
            int c = x - ((y + 1) >> 1);
            v = F3(TT(c - 1), TT(c), TT(c + 1));
          } else if (k == -1) v = F3(L[0], TL, T[0]);
          else {
            int j = -k - 2;
            v = F3(j ? L[j - 1] : TL, L[j], L[j + 1]);
          }
#undef TT
          px[y][x] = v;
        }
      break;
    }
    case 17: { /* 4x4 diagonal down-right, :2702 */
      n = 4;
      uint32_t a = RD32(d, Dst, (long)Offset - S);
      TL = RT(-1);
      L[0] = RL(0);
      for (int k = 0; k < 4; k++) T[k] = (a >> (8 * k)) & 0xFF;
      for (int i = 1; i < 4; i++) L[i] = RL(i);
      int Db[8], *Dg = Db + 3; /* Dg[-3..3] */
      Dg[0] = F3(L[0], TL, T[0]);
      Dg[1] = F3(TL, T[0], T[1]);
      Dg[2] = F3(T[0], T[1], T[2]);
      Dg[3] = F3(T[1], T[2], T[3]);
      Dg[-1] = F3(TL, L[0], L[1]);
      Dg[-2] = F3(L[0], L[1], L[2]);
      Dg[-3] = F3(L[1], L[2], L[3]);
      for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) px[y][x] = Dg[x - y];
      break;
    }
    case 18: { /* 4x4 vertical-left, :2734: reads the top row out to +7 */
      n = 4;
      uint32_t a = RD32(d, Dst, (long)Offset - S), b = RD32(d, Dst, (long)Offset - S + 4);
      for (int k = 0; k < 4; k++) { T[k] = (a >> (8 * k)) & 0xFF; T[4 + k] = (b >> (8 * k)) & 0xFF; }
      for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) {
          int s = x + (y >> 1);
          px[y][x] = (y & 1) ? F3(T[s], T[s + 1], T[s + 2]) : F2(T[s], T[s + 1]);
        }
      break;
    }
    case 19: return;
    default: return;
  }
  for (int y = 0; y < n; y++)
    for (int x = 0; x < n; x++) WR(d, Dst, (long)Offset + (long)y * S + x, (uint32_t)px[y][x]);
#undef RT
#undef RL
}

/* ------------------------------------------------------------------ intra MB syntax */
/* the shared "predicted mode" decode of loc_116220 / loc_116368 / sub_1163DC (MD.cs:1840-1852 etc.) */
static uint32_t pmode_peek(D *d, int r5, int *nbits) {
  uint8_t *ib = IBYTE(d);
  uint32_t r12 = ib[r5 - 8], r6 = ib[r5 - 1];
  if (r12 > r6) r12 = r6;
  if (r12 == 9) r12 = 3;
  r6 = d->r3 >> 28;
  if (r6 >= r12) r6++;
  if (r6 < 9) {
    r12 = r6;
    *nbits = 4;
  } else *nbits = 1;
  return r12;
}
static void set4(D *d, int r5, uint32_t m) {
  uint8_t *ib = IBYTE(d);
  ib[r5] = ib[r5 + 1] = ib[r5 + 8] = ib[r5 + 9] = (uint8_t)m;
}
/* loc_116518, MD.cs:2898 */
static void pred_then_resid8(D *d, barr Dst, int Offset, uint32_t mode) {
  predict_intra(d, mode, Dst, Offset);
  resid8(d, Dst, Offset);
}
/* loc_116628, MD.cs:2945 */
static void pred_then_resid4(D *d, barr Dst, int Offset, uint32_t mode) {
  predict_intra(d, mode, Dst, Offset);
  resid4(d, Dst, Offset);
}
/* loc_116220, MD.cs:1835-1862: uncoded 8x8 block of a sub-mode intra MB */
static void intra_sub_uncoded(D *d, int r5, barr Dst, int Offset) {
  int nb;
  uint32_t m = pmode_peek(d, r5, &nb);
  set4(d, r5, m);
  TAKE(d, nb);
  predict_intra(d, m, Dst, Offset);
}
/* sub_1163DC, MD.cs:2836-2861 */
static uint32_t pmode4(D *d, int r5) {
  int nb;
  uint32_t m = pmode_peek(d, r5, &nb);
  IBYTE(d)[r5] = (uint8_t)m;
  m += 0xA;
  TAKE(d, nb);
  return m;
}
/* loc_116368, MD.cs:2776-2834: coded 8x8 block of a sub-mode intra MB */
static void intra_sub_coded(D *d, int r5, barr Dst, int Offset) {
  if (((d->r3 >> 31) & 1) == 1) {
    d->r3 <<= 1;
    d->nbr--;
    int nb;
    uint32_t m = pmode_peek(d, r5, &nb);
    TAKE(d, nb);
    set4(d, r5, m);
    pred_then_resid8(d, Dst, Offset, m);
  } else {
    uint32_t ue = read_ue(d);
    if (ue >= sizeof(mobi_cbp4_intra)) THROW(d, ORA_E_INDEX);
    uint32_t r4 = mobi_cbp4_intra[ue];
    static const int dr5[4] = {0, 1, 8, 9};
    for (int k = 0; k < 4; k++) {
      uint32_t m = pmode4(d, r5 + dr5[k]);
      if ((r4 >> k) & 1) pred_then_resid4(d, Dst, Offset, m);
      else predict_intra(d, m, Dst, Offset);
      if (k == 0) Offset += 4;
      else if (k == 1) Offset += d->Stride * 4 - 4;
      else if (k == 2) Offset += 4;
    }
  }
}
/* sub_116508, MD.cs:2869-2896: coded 8x8 block with a fixed mode */
static void intra_full_coded(D *d, barr Dst, int Offset, uint32_t mode) {
  if (((d->r3 >> 31) & 1) == 1) {
    d->r3 += d->r3;
    d->nbr--;
    pred_then_resid8(d, Dst, Offset, mode);
  } else {
    mode += 0xA;
    uint32_t ue = read_ue(d);
    if (ue >= sizeof(mobi_cbp4_intra)) THROW(d, ORA_E_INDEX);
    int r4 = mobi_cbp4_intra[ue];
    for (int k = 0; k < 4; k++) {
      if ((r4 >> k) & 1) pred_then_resid4(d, Dst, Offset, mode);
      else predict_intra(d, mode, Dst, Offset);
      if (k == 0) Offset += 4;
      else if (k == 1) Offset += d->Stride * 4 - 4;
      else if (k == 2) Offset += 4;
    }
  }
}
/* loc_116290, MD.cs:1864-1880: chroma of an intra MB */
static void intra_chroma(D *d, uint32_t cbp, int Offset) {
  const int S = d->Stride;
  uint32_t m = d->r3 >> 29;
  TAKE(d, 3);
  if (m == 2) {
    m = 9;
    int p = read_se(d);
    plane_pred(d, d->UV[0], Offset / 2, 8, p);
    p = read_se(d);
    plane_pred(d, d->UV[0], Offset / 2 + S / 2, 8, p);
  }
  if (((cbp >> 4) & 1) == 1) intra_full_coded(d, d->UV[0], Offset / 2, m);
  else predict_intra(d, m, d->UV[0], Offset / 2);
  if (((cbp >> 5) & 1) == 1) intra_full_coded(d, d->UV[0], Offset / 2 + S / 2, m);
  else predict_intra(d, m, d->UV[0], Offset / 2 + S / 2);
}
/* DecIntraFullBlockPMode, MD.cs:1759-1786 */
static void dec_intra_full(D *d, int Offset) {
  const int S = d->Stride;
  uint32_t ue = read_ue(d);
  if (ue >= sizeof(mobi_cbp_intra)) THROW(d, ORA_E_INDEX);
  uint32_t cbp = mobi_cbp_intra[ue];
  uint32_t m = d->r3 >> 29;
  TAKE(d, 3);
  if (m == 2) {
    m = 9;
    int p = read_se(d); /* sub_1167BC reads its parameter first, :3019 */
    plane_pred(d, d->Y[0], Offset, 16, p);
  }
  static const int bx[4] = {0, 8, 0, 8}, by[4] = {0, 0, 8, 8};
  for (int k = 0; k < 4; k++) {
    int o = Offset + by[k] * S + bx[k];
    if ((cbp >> k) & 1) intra_full_coded(d, d->Y[0], o, m);
    else predict_intra(d, m, d->Y[0], o);
  }
  intra_chroma(d, cbp, Offset);
}
/* DecIntraSubBlockPMode, MD.cs:1789-1807 */
static void dec_intra_sub(D *d, int Offset) {
  const int S = d->Stride;
  uint32_t ue = read_ue(d);
  if (ue >= sizeof(mobi_cbp_intra)) THROW(d, ORA_E_INDEX);
  uint32_t cbp = mobi_cbp_intra[ue];
  static const int bx[4] = {0, 8, 0, 8}, by[4] = {0, 0, 8, 8}, ci[4] = {9, 0xB, 0x19, 0x1B};
  for (int k = 0; k < 4; k++) {
    int o = Offset + by[k] * S + bx[k];
    if (((cbp >> k) & 1) == 0) intra_sub_uncoded(d, ci[k], d->Y[0], o);
    else intra_sub_coded(d, ci[k], d->Y[0], o);
  }
  intra_chroma(d, cbp, Offset);
}

/* ------------------------------------------------------------------ frame */
static barr new_bytes(long n) {
  barr a;
  a.len = n < 0 ? 0 : n;
  a.p = (uint8_t *)calloc(a.len ? a.len : 1, 1);
  return a;
}
/* DecodeVXS2, MD.cs:97-259 (up to, not including, the Bitmap block) */
static void decode_vxs2(D *d) {
  free(d->Y[5].p);
  free(d->UV[5].p);
  for (int i = 5; i > 0; i--) {
    d->Y[i] = d->Y[i - 1];
    d->UV[i] = d->UV[i - 1];
  }
  d->Y[0] = new_bytes((long)d->Stride * d->Height);
  d->UV[0] = new_bytes((long)d->Stride * d->Height / 2);
  d->nbr = 0;
  d->r3 = data_u16(d, d->Offset);
  d->Offset += 2;
  d->r3 <<= 16;
  int iframe = (d->r3 >> 31) == 1;
  d->r3 += d->r3;
  const int S = d->Stride;
  if (!iframe) {
    if (--d->nbr < 0) fill_bits(d);
    if (d->Version == MOBI_VER_MOFLEX3DS) {
      uint32_t q = d->Quantizer;
      int dq = read_se(d);
      if (q == 0) setup_quant(d, q);
      else if (dq != 0) setup_quant(d, (uint32_t)(q + (uint32_t)dq));
    } else if (d->Version == MOBI_VER_MODSDS) {
      int dq = read_se(d);
      if (dq != 0) setup_quant(d, (uint32_t)(d->Quantizer + (uint32_t)dq));
    }
    d->Internal[218] = 0;
    int io = 221;
    int w = (int)d->Width + 0x20;
    for (;;) {
      *IN(d, io) = 0;
      *IN(d, io + 1) = 0;
      io += 2;
      w -= 0x10;
      if (w <= 0) break;
    }
    int r11 = 0;
    int h = (int)d->Height;
    for (;;) {
      w = (int)d->Width;
      io = 221;
      for (;;) {
        int v[6];
        for (int k = 0; k < 6; k++) v[k] = (int)*IN(d, io + k);
        io += 2;
#define CSWAP(a, b) do { if (v[a] > v[b]) { int t = v[a]; v[a] = v[b]; v[b] = t; } } while (0)
        CSWAP(0, 2); CSWAP(2, 4); CSWAP(0, 2);
        CSWAP(1, 3); CSWAP(3, 5); CSWAP(1, 3);
#undef CSWAP
        d->Internal[219] = (uint32_t)v[2];
        d->Internal[220] = (uint32_t)v[3];
        *IN(d, io) = 0;
        *IN(d, io + 1) = 0;
        pblock(d, 0, 0, io, r11);
        r11 += 0x10;
        w -= 0x10;
        if (w <= 0) break;
      }
      r11 += S * 16;
      r11 -= (int)d->Width;
      h -= 0x10;
      if (h <= 0) break;
    }
  } else {
    d->YuvFormat = (d->r3 >> 31) & 1;
    d->r3 += d->r3;
    d->Internal[218] = (d->r3 >> 31) & 1;
    d->r3 += d->r3;
    d->nbr -= 3;
    if (d->nbr < 0) fill_bits(d);
    uint32_t q = d->r3 >> 26;
    TAKE(d, 6);
    if (d->Quantizer != q) setup_quant(d, q);
    int r11 = 0;
    int h = (int)d->Height;
    for (;;) {
      int w = (int)d->Width;
      for (;;) {
        int sub = (d->r3 >> 31) == 1;
        d->r3 += d->r3;
        d->nbr--;
        if (d->nbr < 0) fill_bits(d);
        if (sub) dec_intra_sub(d, r11);
        else dec_intra_full(d, r11);
        r11 += 0x10;
        w -= 0x10;
        if (w <= 0) break;
      }
      r11 += S * 16;
      r11 -= (int)d->Width;
      h -= 0x10;
      if (h <= 0) break;
    }
  }
}

/* ------------------------------------------------------------------ public API */
mobi_oracle *mobi_oracle_create(uint32_t width, uint32_t height, int version) {
  D *d = (D *)calloc(1, sizeof(D));
  if (!d) return NULL;
  d->Width = width;
  d->Height = height;
  d->Version = version;
  d->Stride = (width <= 256) ? 256 : (width <= 512) ? 512 : 1024; /* MD.cs:50-52 */
  return d;
}
void mobi_oracle_destroy(mobi_oracle *d) {
  if (!d) return;
  for (int i = 0; i < 6; i++) {
    free(d->Y[i].p);
    free(d->UV[i].p);
  }
  free(d);
}
int mobi_oracle_decode(mobi_oracle *d, const uint8_t *data, size_t len, int32_t *offset) {
  d->Data = data;
  d->DataLen = (long)len;
  d->Offset = *offset;
  int rc = setjmp(d->jb);
  if (rc == 0) {
    if (d->Version == MOBI_VER_MODSDS || d->Version == MOBI_VER_MOFLEX3DS) decode_vxs2(d);
    else rc = ORA_E_VERSION;
  }
  *offset = d->Offset;
  return rc;
}
/* timing aid for bench.py's cpu_baseline: DecodeFrame() over a whole clip in one call (Data = the bytes up to the end of the
 * frame, Offset = its start, as the parity tests do), optionally followed by the Bitmap conversion of every frame */
int mobi_oracle_decode_clip(mobi_oracle *d, const uint8_t *data, const uint32_t *frame_off, int n_frames, uint32_t *argb_or_null) {
  for (int f = 0; f < n_frames; f++) {
    int32_t off = (int32_t)frame_off[f];
    const int rc = mobi_oracle_decode(d, data, frame_off[f + 1], &off);
    if (rc != 0) return rc;
    if (argb_or_null) mobi_oracle_argb(d, argb_or_null);
  }
  return n_frames;
}
int mobi_oracle_stride(const mobi_oracle *d) { return d->Stride; }
uint32_t mobi_oracle_quantizer(const mobi_oracle *d) { return d->Quantizer; }
uint32_t mobi_oracle_yuvformat(const mobi_oracle *d) { return d->YuvFormat; }
const uint8_t *mobi_oracle_y(const mobi_oracle *d, int idx) { return (idx >= 0 && idx < 6) ? d->Y[idx].p : NULL; }
const uint8_t *mobi_oracle_uv(const mobi_oracle *d, int idx) { return (idx >= 0 && idx < 6) ? d->UV[idx].p : NULL; }
uint32_t *mobi_oracle_internal(mobi_oracle *d) { return d->Internal; }

/* ---- Analyzer.InterPredict2x2, Analyzer.cs:608-681, over every 2x2 block of every macroblock (:683-693) ---- */
void mobi_oracle_motion_search(const mobi_oracle *d, const uint8_t *src, uint32_t *out) {
  const int W = (int)d->Width, H = (int)d->Height, S = d->Stride;
  for (int mby = 0; mby < H / 16; mby++)
    for (int mbx = 0; mbx < W / 16; mbx++)
      for (int Y = 0; Y < 8; Y++)
        for (int X = 0; X < 8; X++) {
          const int BX = mbx * 16, BY = mby * 16;                 /* Block.X, Block.Y */
          const uint8_t *c0 = src + (BY + Y * 2) * W + BX + X * 2; /* cmp[0..3], Encoder/MacroBlock.cs:80-83 */
          const int cmp[4] = {c0[0], c0[1], c0[W], c0[W + 1]};
          int rdx = 0, rdy = 0, rframe = 0, resultscore = INT32_MAX; /* InterPredict2x2Result defaults, :613-615 */
          for (int i = 0; i < 5; i++) {                            /* :616 */
            const uint8_t *past = d->Y[i].p;
            if (!past) break;                                      /* :618 */
            int St = 6, centerx = 0, centery = 0, centerscore = 0; /* :619-622 */
            while (St >= 1) {                                      /* :623 */
              int bestscore = INT32_MAX, bestx = 0, besty = 0;
              for (int y = -St; y <= St; y += St) {                /* :628 */
                if (BY + y + centery + Y * 2 < 0 || BY + 2 + y + centery + Y * 2 > H) continue; /* :630 */
                for (int x = -St; x <= St; x += St) {
                  if (BX + x + centerx + X * 2 < 0 || BX + 2 + x + centerx + X * 2 > W) continue; /* :633 */
                  const uint8_t *ps = past + (BY + Y * 2) * S + (BX + X * 2) + x + centerx + (y + centery) * S; /* :635 */
                  int a = cmp[0] - ps[0], b = cmp[1] - ps[1], c = cmp[2] - ps[S], e = cmp[3] - ps[S + 1];
                  if (a < 0) a = -a;
                  if (b < 0) b = -b;
                  if (c < 0) c = -c;
                  if (e < 0) e = -e;
                  const int score = a + b + c + e;                 /* :645 */
                  const int nx = x + centerx, ny = y + centery;
                  if (score < bestscore ||
                      (score == bestscore && (nx < 0 ? -nx : nx) + (ny < 0 ? -ny : ny) < (bestx < 0 ? -bestx : bestx) + (besty < 0 ? -besty : besty))) { /* :646-651 */
                    bestx = nx;
                    besty = ny;
                    bestscore = score;
                  }
                }
              }
              St /= 2;                                             /* :661-664 */
              centerx = bestx;
              centery = besty;
              centerscore = bestscore;
            }
            const int cx2 = centerx * 2, cy2 = centery * 2;
            if (centerscore < resultscore ||
                (centerscore == resultscore && (cx2 < 0 ? -cx2 : cx2) + (cy2 < 0 ? -cy2 : cy2) < (rdx < 0 ? -rdx : rdx) + (rdy < 0 ? -rdy : rdy))) { /* :666-671 */
              rdx = cx2;
              rdy = cy2;
              rframe = i;
              resultscore = centerscore;
            }
          }
          const uint32_t sc = resultscore == INT32_MAX ? 0xFFFu : (uint32_t)resultscore;
          out[((size_t)(mby * (W / 16) + mbx) * 64) + Y * 8 + X] = ((uint32_t)rdx & 0xFFu) | (((uint32_t)rdy & 0xFFu) << 8) | ((uint32_t)rframe << 16) | (sc << 20);
        }
}

/* ---- YUV -> ARGB, MD.cs:260-323 (compile with -ffp-contract=off: every operator rounds once, as in the CLR) ---- */
int mobi_oracle_argb(const mobi_oracle *d, uint32_t *out) {
  const uint8_t *Y = d->Y[0].p, *UV = d->UV[0].p;
  if (!Y || !UV) return ORA_E_NULLREF;
  const int S = d->Stride, W = (int)d->Width, H = (int)d->Height;
  for (int y = 0; y < H; y++) {
    for (int x = 0; x < W; x++) {
      volatile float Y2 = (float)Y[y * S + x];                    /* :266 */
      const int c = y / 2 * S + x / 2;
      volatile float U = (float)UV[c] - 128.0f;                   /* :267 */
      volatile float V = (float)UV[c + S / 2] - 128.0f;           /* :268 */
      if (x != W - 1 && y != H - 1) {                             /* :269 */
        switch ((x & 1) | ((y & 1) << 1)) {                       /* :271 */
          case 1: /* :273-278 */
            U += (float)UV[c + 1] - 128.0f; V += (float)UV[c + 1 + S / 2] - 128.0f;
            U /= 2.0f; V /= 2.0f;
            break;
          case 2: /* :279-284 */
            U += (float)UV[c + S] - 128.0f; V += (float)UV[c + S + S / 2] - 128.0f;
            U /= 2.0f; V /= 2.0f;
            break;
          case 3: /* :285-294 */
            U += (float)UV[c + 1] - 128.0f; V += (float)UV[c + 1 + S / 2] - 128.0f;
            U += (float)UV[c + S] - 128.0f; V += (float)UV[c + S + S / 2] - 128.0f;
            U += (float)UV[c + 1 + S] - 128.0f; V += (float)UV[c + 1 + S + S / 2] - 128.0f;
            U /= 4.0f; V /= 4.0f;
            break;
        }
      }
      volatile float R, G, B, t;
      if (d->Version == MOBI_VER_MOFLEX3DS) {                     /* :297-305 */
        t = 1.420f * V; R = Y2 + t;
        t = 0.344f * U; G = Y2 - t; t = 0.714f * V; G = G - t;
        t = 1.772f * U; B = Y2 + t;
        t = R - 16.0f; t = t * 255.0f; R = t / 239.0f;            /* (255f - 16f) folds to 239f */
        t = G - 16.0f; t = t * 255.0f; G = t / 239.0f;
        t = B - 16.0f; t = t * 255.0f; B = t / 239.0f;
      } else if (d->Version == MOBI_VER_MODSDS) {                 /* :306-311: integer arithmetic on truncated values */
        R = (float)((int)Y2 + (int)U - (int)V);
        G = (float)((int)Y2 + (int)V);
        B = (float)((int)Y2 - (int)U - (int)V);
      } else {
        R = G = B = 0.0f;
      }
      if (R < 0) R = 0;                                           /* :313-318 */
      if (R > 255) R = 255;
      if (G < 0) G = 0;
      if (G > 255) G = 255;
      if (B < 0) B = 0;
      if (B > 255) B = 255;
      out[y * W + x] = 0xFF000000u | ((uint32_t)(int)R << 16) | ((uint32_t)(int)G << 8) | (uint32_t)(int)B; /* Color.FromArgb(r,g,b).ToArgb(), :319 */
    }
  }
  return ORA_OK;
}

/* ---- unit-level hooks: run one primitive on caller memory through a scratch decoder ---- */
static D *scratch(int stride) {
  D *d = (D *)calloc(1, sizeof(D));
  d->Stride = stride;
  d->Version = MOBI_VER_MOFLEX3DS;
  return d;
}
int mobi_oracle_idct8(const int32_t *coef, int variant, uint8_t *dst, int dst_len, int offset, int stride) {
  D *d = scratch(stride);
  barr a = {dst, dst_len};
  for (int i = 0; i < 64; i++) d->Internal[90 + i] = (uint32_t)coef[i];
  int rc = setjmp(d->jb);
  if (rc == 0) {
    if (variant == 64) idct64p8(d, a, offset);
    else if (variant == 16) idct16p8(d, a, offset);
    else if (variant == 3) idct3p8(d, a, offset);
    else idct1p(d, a, offset, 8);
  }
  free(d);
  return rc;
}
int mobi_oracle_idct4(const int32_t *coef, int variant, uint8_t *dst, int dst_len, int offset, int stride) {
  D *d = scratch(stride);
  barr a = {dst, dst_len};
  for (int i = 0; i < 16; i++) d->Internal[90 + i] = (uint32_t)coef[i];
  int rc = setjmp(d->jb);
  if (rc == 0) {
    if (variant == 16) idct16p4(d, a, offset);
    else idct1p(d, a, offset, 4);
  }
  free(d);
  return rc;
}
int mobi_oracle_copyblock(const uint8_t *src, int src_len, int dx, int dy, uint32_t w, uint32_t h,
                          uint8_t *dst, int dst_len, int offset, int stride) {
  D *d = scratch(stride);
  barr s = {(uint8_t *)src, src_len}, t = {dst, dst_len};
  int rc = setjmp(d->jb);
  if (rc == 0) copy_block(d, s, dx, dy, w, h, t, offset);
  free(d);
  return rc;
}
int mobi_oracle_predict(int mode, uint8_t *dst, int dst_len, int offset, int stride, int is_uv) {
  if (mode == 2 || mode == 12) return ORA_E_VERSION;
  D *d = scratch(stride);
  barr a = {dst, dst_len};
  if (is_uv) d->UV[0] = a;
  int rc = setjmp(d->jb);
  if (rc == 0) predict_intra(d, (uint32_t)mode, a, offset);
  d->UV[0].p = NULL;
  free(d);
  return rc;
}
int mobi_oracle_plane(int size, int param, uint8_t *dst, int dst_len, int offset, int stride) {
  D *d = scratch(stride);
  barr a = {dst, dst_len};
  int rc = setjmp(d->jb);
  if (rc == 0) plane_pred(d, a, offset, size, param);
  free(d);
  return rc;
}

/* ---- encoder-side forward transforms: MobiEncoder.DCT64 (Encoder/MobiEncoder.cs:962-1010) and DCT16 (:1146-1178).
 * Integer arithmetic throughout: the input (a block of residuals, Block - CompVals, Encoder/MacroBlock.cs:584-588) is scaled by 64, each
 * pass is a fixed integer matrix followed by a C# integer division (truncation toward zero, like C's), rows first, then columns;
 * note that the second pass writes its results transposed (tmp2[i * 8 + k] is coefficient k of COLUMN i), exactly as the reference does. */
static void ora_dct8_pass(int p, int q, int r, int s, int t, int u, int v, int w, int32_t *o) { /* o[0..7] */
  o[0] = (w + v + u + t + s + r + q + p) / 8;
  o[1] = (-40 * w + 40 * v - 12 * u + 12 * t - 24 * s + 24 * r - 48 * q + 48 * p) / 289;
  o[2] = (w + v - 2 * u - 2 * t - s - r + 2 * q + 2 * p) / 10;
  o[3] = (12 * w - 12 * v + 24 * u - 24 * t + 48 * s - 48 * r - 40 * q + 40 * p) / 289;
  o[4] = (-w - v + u + t - s - r + q + p) / 8;
  o[5] = (48 * w - 48 * v - 40 * u + 40 * t - 12 * s + 12 * r - 24 * q + 24 * p) / 289;
  o[6] = (-2 * w - 2 * v - u - t + 2 * s + 2 * r + q + p) / 10;
  o[7] = (24 * w - 24 * v + 48 * u - 48 * t - 40 * s + 40 * r - 12 * q + 12 * p) / 289;
}
void mobi_oracle_dct8(const int32_t *in, int32_t *out) {
  int32_t px[64], tmp[64];
  for (int i = 0; i < 64; i++) px[i] = in[i] * 64;                                       /* :964-968 */
  for (int i = 0; i < 8; i++)                                                            /* :970-988: p,q,r,s,t,u,v,w = columns 0,7,2,5,3,4,1,6 */
    ora_dct8_pass(px[i * 8 + 0], px[i * 8 + 7], px[i * 8 + 2], px[i * 8 + 5], px[i * 8 + 3], px[i * 8 + 4], px[i * 8 + 1], px[i * 8 + 6], tmp + i * 8);
  for (int i = 0; i < 8; i++)                                                            /* :990-1008 */
    ora_dct8_pass(tmp[0 * 8 + i], tmp[7 * 8 + i], tmp[2 * 8 + i], tmp[5 * 8 + i], tmp[3 * 8 + i], tmp[4 * 8 + i], tmp[1 * 8 + i], tmp[6 * 8 + i], out + i * 8);
}
static void ora_dct4_pass(int q, int r, int s, int t, int32_t *o) {
  o[0] = (t + s + r + q) / 4;
  o[1] = (-2 * t - s + r + 2 * q) / 5;
  o[2] = (t - s - r + q) / 4;
  o[3] = (-t + 2 * s - 2 * r + q) / 5;
}
void mobi_oracle_dct4(const int32_t *in, int32_t *out) {
  int32_t px[16], tmp[16];
  for (int i = 0; i < 16; i++) px[i] = in[i] * 64;                                       /* :1148-1152 */
  for (int i = 0; i < 4; i++) ora_dct4_pass(px[i * 4 + 0], px[i * 4 + 1], px[i * 4 + 2], px[i * 4 + 3], tmp + i * 4);          /* :1154-1164 */
  for (int i = 0; i < 4; i++) ora_dct4_pass(tmp[0 * 4 + i], tmp[1 * 4 + i], tmp[2 * 4 + i], tmp[3 * 4 + i], out + i * 4);      /* :1166-1176 */
}
