/* abi_caller.c -- a plain C caller of libmobiclip_hip.so, nothing but include/mobiclip_hip.h and libc: what a host written in
 * any language with a C FFI does (the reference's would be C# P/Invoke).
 *
 *   abi_caller <stream.bin> <width> <height> <version> <n_frames> <off_0> ... <off_n>
 *
 * Decodes the frames the way the reference's callers drive MobiclipDecoder (Program.cs:69-71: d.Data = frame; d.Offset = o;
 * d.DecodeFrame()), and prints for every frame one line
 *   <frame> <rc> <offset_after> <quantizer> <sha256 of Y[0]> <sha256 of UV[0]>
 *
 *   abi_caller --batch-async <stream.bin> <width> <height> <version> <n_frames> <off_0> ... <off_n>
 *
 * The same stream as three clips of a batch that parses on the GPU, through mobi_batch_submit / mobi_batch_wait with two frame steps
 * in flight; prints the same lines (for clip 2).
 *   abi_caller --batch-gop <the same arguments>
 *
 * ... through mobi_batch_decode_gop, groups of 4, 1, 6, 2, ... frames per call (parsed side by side on the GPU); the same lines (Quantizer only
 * on a group's last frame, 0 elsewhere).
 * tests/test_abi_c_caller.py compares the lines with tests/golden/golden.json.  Test tool: not part of the product. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mobiclip_hip.h"

/* HIP ordinal: MOBI_DEVICE (default 0), so that a node with several GPUs can run this caller on any of them */
static int test_device(void) { const char *e = getenv("MOBI_DEVICE"); return e ? atoi(e) : 0; }

/* ---- SHA-256 (FIPS 180-4), small and slow ---- */
static const uint32_t K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha_block(uint32_t h[8], const uint8_t *p) {
  uint32_t w[64], a[8];
  int i;
  for (i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
  for (i = 16; i < 64; i++) {
    const uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  memcpy(a, h, sizeof(a));
  for (i = 0; i < 64; i++) {
    const uint32_t S1 = ROR(a[4], 6) ^ ROR(a[4], 11) ^ ROR(a[4], 25), ch = (a[4] & a[5]) ^ (~a[4] & a[6]);
    const uint32_t t1 = a[7] + S1 + ch + K[i] + w[i];
    const uint32_t S0 = ROR(a[0], 2) ^ ROR(a[0], 13) ^ ROR(a[0], 22), mj = (a[0] & a[1]) ^ (a[0] & a[2]) ^ (a[1] & a[2]);
    const uint32_t t2 = S0 + mj;
    a[7] = a[6]; a[6] = a[5]; a[5] = a[4]; a[4] = a[3] + t1; a[3] = a[2]; a[2] = a[1]; a[1] = a[0]; a[0] = t1 + t2;
  }
  for (i = 0; i < 8; i++) h[i] += a[i];
}
static void sha256_hex(const uint8_t *data, size_t len, char out[65]) {
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  uint8_t tail[128];
  size_t i, full = len / 64 * 64, rem = len - full, tl;
  for (i = 0; i < full; i += 64) sha_block(h, data + i);
  memset(tail, 0, sizeof(tail));
  memcpy(tail, data + full, rem);
  tail[rem] = 0x80;
  tl = rem + 9 <= 64 ? 64 : 128;
  for (i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(((uint64_t)len * 8) >> (8 * i));
  for (i = 0; i < tl; i += 64) sha_block(h, tail + i);
  for (i = 0; i < 8; i++) sprintf(out + 8 * i, "%08x", h[i]);
}

static int run_batch_async(const uint8_t *data, uint32_t w, uint32_t hgt, int version, int nf, char **offs) {
  enum { N = 3 };
  mobi_batch *b = mobi_batch_create(N, w, hgt, version, test_device());
  if (!b) { fprintf(stderr, "mobi_batch_create failed\n"); return 3; }
  if (mobi_batch_set_parse_mode(b, 1) != MOBI_OK) return 3;             /* the bitstreams are parsed on the GPU */
  const size_t ysz = (size_t)mobi_batch_stride(b) * hgt;
  uint8_t *y = (uint8_t *)malloc(ysz), *uv = (uint8_t *)malloc(ysz / 2);
  const uint8_t *ptrs[N] = {data, data, data};
  char hy[65], huv[65];
  int i, bad = 0;
  for (i = 0; i <= nf; i++) {
    if (i < nf) {                                                        /* frame i goes in behind frame i - 1 ... */
      const size_t l = (size_t)atol(offs[i + 1]);
      const size_t lens[N] = {l, l, l};
      const int32_t o = atoi(offs[i]), off[N] = {o, o, o};
      if (mobi_batch_submit(b, ptrs, lens, off) != MOBI_OK) { fprintf(stderr, "submit %d failed\n", i); return 3; }
    }
    if (i > 0) {                                                         /* ... whose results are collected now */
      int32_t after[N];
      int rc[N];
      if (mobi_batch_wait(b, after, rc) != MOBI_OK) { fprintf(stderr, "wait %d failed\n", i - 1); return 3; }
      hy[0] = huv[0] = '-'; hy[1] = huv[1] = 0;
      /* frame i - 1 is ring position 1 once frame i has been submitted (Y[1], MD.cs:102-106), position 0 after the last frame */
      if (rc[2] == MOBI_OK && mobi_batch_get_planes(b, 2, i < nf ? 1 : 0, y, uv) == MOBI_OK) {
        sha256_hex(y, ysz, hy);
        sha256_hex(uv, ysz / 2, huv);
      } else bad = 1;
      printf("%d %d %d %u %s %s\n", i - 1, rc[2], (int)after[2], mobi_batch_quantizer(b, 2), hy, huv);
    }
  }
  mobi_batch_destroy(b);
  free(y); free(uv);
  return bad;
}

/* --batch-gop: the same stream as three clips through mobi_batch_decode_gop, up to MOBI_GOP frames per call (r06: the frames of a group
 * are parsed side by side on the GPU); every frame of a group is read back from the ring afterwards (frame k of K at ring index K - 1 - k) */
static int run_batch_gop(const uint8_t *data, uint32_t w, uint32_t hgt, int version, int nf, char **offs) {
  enum { N = 3, MAXK = 6 };
  mobi_batch *b = mobi_batch_create(N, w, hgt, version, test_device());
  if (!b) { fprintf(stderr, "mobi_batch_create failed\n"); return 3; }
  if (mobi_batch_set_parse_mode(b, 3) != MOBI_OK) return 3;             /* parse on the GPU, the lock-step parser in front */
  const size_t ysz = (size_t)mobi_batch_stride(b) * hgt;
  uint8_t *y = (uint8_t *)malloc(ysz), *uv = (uint8_t *)malloc(ysz / 2);
  char hy[65], huv[65];
  int f0 = 0, bad = 0, turn = 0;
  while (f0 < nf) {
    static const int sizes[4] = {4, 1, 6, 2};
    int K = sizes[turn++ & 3], k, c;
    if (K > nf - f0) K = nf - f0;
    const uint8_t *ptrs[N * MAXK];
    size_t lens[N * MAXK];
    int32_t off[N * MAXK];
    int rc[N * MAXK];
    for (k = 0; k < K; k++)
      for (c = 0; c < N; c++) {                                         /* [k * n_clips + c]: frame k of the group, clip c */
        ptrs[k * N + c] = data;
        lens[k * N + c] = (size_t)atol(offs[f0 + k + 1]);
        off[k * N + c] = atoi(offs[f0 + k]);
      }
    if (mobi_batch_decode_gop(b, K, ptrs, lens, off, rc) != MOBI_OK) { fprintf(stderr, "decode_gop at frame %d failed\n", f0); return 3; }
    for (k = 0; k < K; k++) {
      hy[0] = huv[0] = '-'; hy[1] = huv[1] = 0;
      if (rc[k * N + 2] == MOBI_OK && mobi_batch_get_planes(b, 2, K - 1 - k, y, uv) == MOBI_OK) {
        sha256_hex(y, ysz, hy);
        sha256_hex(uv, ysz / 2, huv);
      } else bad = 1;
      /* (Quantizer is the decoder's field: it describes the group's last frame) */
      printf("%d %d %d %u %s %s\n", f0 + k, rc[k * N + 2], (int)off[k * N + 2], k == K - 1 ? mobi_batch_quantizer(b, 2) : 0u, hy, huv);
    }
    f0 += K;
  }
  mobi_batch_destroy(b);
  free(y); free(uv);
  return bad;
}

int main(int argc, char **argv) {
  const int batch_async = argc > 1 && strcmp(argv[1], "--batch-async") == 0;
  const int batch_gop = argc > 1 && strcmp(argv[1], "--batch-gop") == 0;
  if (batch_async || batch_gop) { argv++; argc--; }
  if (argc < 7) { fprintf(stderr, "usage: abi_caller stream.bin width height version n_frames off_0 .. off_n\n"); return 2; }
  const uint32_t w = (uint32_t)atoi(argv[2]), hgt = (uint32_t)atoi(argv[3]);
  const int version = atoi(argv[4]), nf = atoi(argv[5]);
  if (argc != 6 + nf + 1) { fprintf(stderr, "expected %d frame offsets\n", nf + 1); return 2; }
  FILE *f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  fseek(f, 0, SEEK_END);
  const long flen = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t *data = (uint8_t *)malloc((size_t)flen + 1);
  if (fread(data, 1, (size_t)flen, f) != (size_t)flen) { fprintf(stderr, "short read\n"); return 2; }
  fclose(f);

  if (batch_async) { const int e = run_batch_async(data, w, hgt, version, nf, argv + 6); free(data); return e; }
  if (batch_gop) { const int e = run_batch_gop(data, w, hgt, version, nf, argv + 6); free(data); return e; }
  mobi_dec *d = mobi_create(w, hgt, version, test_device());                    /* new MobiclipDecoder(Width, Height, Version) */
  if (!d) { fprintf(stderr, "mobi_create failed: %s\n", mobi_error_string(MOBI_E_DEVICE)); return 3; }
  const int stride = mobi_stride(d);
  const size_t ysz = (size_t)stride * hgt;
  uint8_t *y = (uint8_t *)malloc(ysz), *uv = (uint8_t *)malloc(ysz / 2);
  char hy[65], huv[65];
  int i, bad = 0;
  for (i = 0; i < nf; i++) {
    int32_t off = atoi(argv[6 + i]);                                  /* d.Offset = ... */
    const size_t len = (size_t)atol(argv[6 + i + 1]);                 /* d.Data = the stream up to the end of this frame */
    const int rc = mobi_decode(d, data, len, &off);                   /* d.DecodeFrame() */
    hy[0] = huv[0] = '-'; hy[1] = huv[1] = 0;
    if (rc == MOBI_OK && mobi_get_planes(d, 0, y, uv) == MOBI_OK) {   /* d.Y[0], d.UV[0] */
      sha256_hex(y, ysz, hy);
      sha256_hex(uv, ysz / 2, huv);
    } else bad = 1;
    printf("%d %d %d %u %s %s\n", i, rc, (int)off, mobi_quantizer(d), hy, huv);
  }
  mobi_destroy(d);
  free(y); free(uv); free(data);
  return bad;
}
