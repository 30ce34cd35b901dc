#!/usr/bin/env python3
"""Generates tests/golden/unit_vectors.npz: unit-level inputs and the outputs the REFERENCE's statements give for them.

Runs only in the build container (needs /root/reference): the outputs come from oracle/_ref/libmobi_csref.so, the mechanical
C#->C++ transliteration of the reference (oracle/tools/cs2cpp.py), i.e. from the reference's own lines, not from this repository's
oracle or kernels.  For every unit there are TWO reference statements, the decoder's and the encoder's copy (SURVEY.md 8(c),
in-source redundancies 1-3):

  intra predictors   MobiclipDecoder.PredictIntra (MD.cs:1883-2774)          Encoder/MacroBlock.cs GetCompvals8x8 :630, GetCompvals4x4 :1184
  plane predictors   (decoder: reads its parameter from the bitstream, only   Encoder/MacroBlock.cs PredictIntraPlane16x16 :1477, 8x8 :1630, 4x4 :1716
                      reachable through whole-stream decodes)
  inverse transforms (decoder: through whole-stream decodes)                  Encoder/MobiEncoder.cs IDCT64 :1012, IDCT16 :1180
  CopyBlock          (decoder: MD.cs:418-456, through whole-stream decodes)   Utils/FrameUtil.cs GetPBlock :96-143
  forward transforms --                                                       Encoder/MobiEncoder.cs DCT64 :962, DCT16 :1146

Both statements are run; where both exist for the same inputs they must agree (asserted here), and what is stored is that result.
tests/test_unit_vectors.py checks the oracle's unit entry points (and, on the GPU, the kernels) against the stored vectors.
This file is a fixture generator: it contains no reference text, only calls into the generated library.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SO = os.path.join(ROOT, "oracle", "_ref", "libmobi_csref.so")
OUT = os.path.join(ROOT, "tests", "golden", "unit_vectors.npz")
S, ROWS = 256, 48  # a 256-wide plane (Stride 256, MD.cs:50-52), 48 rows
MODES8 = [0, 1, 3, 4, 5, 6, 7, 8]
MODES4 = [10, 11, 13, 14, 15, 16, 17, 18]


def lib():
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(ROOT, "oracle", "tools", "cs2cpp.py")):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "oracle", "tools", "cs2cpp.py")])
    L = C.CDLL(SO)
    L.csref_create.restype = C.c_void_p
    L.csref_create.argtypes = [C.c_uint, C.c_uint, C.c_int]
    L.csref_dec_predict.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_int]
    L.csref_enc_compvals.argtypes = [C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.csref_enc_plane.argtypes = [C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.csref_enc_idct.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.csref_enc_dct.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    L.csref_enc_getpblock.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_void_p]
    return L


def intra_cases():
    """(mode, x, y, is_uv): block positions that exercise every availability the decoder distinguishes (MD.cs:1886-1887, 1923-1924):
    interior, first row, first column, corner, and the V plane's first column (column Stride/2 of the UV array)."""
    cases = []
    for m in MODES8:
        for (x, y, uv) in [(16, 16, 0), (40, 8, 0), (0, 16, 0), (24, 0, 0), (0, 0, 0), (128, 8, 1), (128 + 8, 16, 1), (128, 0, 1), (8, 8, 1)]:
            cases.append((m, x, y, uv))
    for m in MODES4:
        for (x, y, uv) in [(20, 12, 0), (4, 4, 0), (0, 8, 0), (12, 0, 0), (0, 0, 0), (128, 4, 1), (128 + 4, 12, 1), (128, 0, 1)]:
            cases.append((m, x, y, uv))
    return cases


def main():
    if not os.path.isdir("/root/reference"):
        sys.exit("needs /root/reference (build container only)")
    L = lib()
    rng = np.random.default_rng(0x4D4F4249)
    dec = L.csref_create(256, 192, 2)
    out = {}
    # ---- intra predictors: decoder statement and encoder statement on the same random plane
    plane = rng.integers(0, 256, S * ROWS, dtype=np.uint8)
    cases = intra_cases()
    pred = np.zeros((len(cases), 64), np.uint8)
    rc = np.zeros(len(cases), np.int32)
    n_both = 0
    for i, (m, x, y, uv) in enumerate(cases):
        n = 8 if m < 10 else 4
        off = y * S + x
        a = plane.copy()
        r = L.csref_dec_predict(dec, m, a.ctypes.data, a.size, off, uv)
        rc[i] = r
        if r == 0:
            blk = a.reshape(ROWS, S)[y:y + n, x:x + n]
            pred[i, :n * n] = blk.ravel()
            # nothing outside the block may change
            b = a.copy().reshape(ROWS, S)
            b[y:y + n, x:x + n] = plane.reshape(ROWS, S)[y:y + n, x:x + n]
            assert np.array_equal(b.ravel(), plane), (m, x, y)
        # the encoder's copy: block (X, Y) of the plane that starts at Offset.  Its availability rule is X > 0 / Y > 0 inside that plane
        # (MacroBlock.cs:669-670), so the V plane is Offset = Stride / 2 with X counted from there
        eoff = S // 2 if (uv and x >= S // 2) else 0
        ex = x - eoff
        e = plane.copy()
        eo = np.zeros(64, np.uint8)
        er = L.csref_enc_compvals(m, e.ctypes.data, e.size, ex, y, S, eoff, eo.ctypes.data)
        if er == 0 and r == 0:
            assert np.array_equal(eo[:n * n], pred[i, :n * n]), ("decoder and encoder statements disagree", m, x, y, uv)
            n_both += 1
        elif r == 0 and er != 0:
            # the encoder fetches whole neighbour blocks (FrameUtil.GetBlockPixels*), so at the plane's first rows / columns it throws
            # where the decoder, which reads single samples, does not; the decoder's result stands
            pass
    out.update(intra_plane=plane, intra_cases=np.array(cases, np.int32), intra_pred=pred, intra_rc=rc)
    # ---- plane predictors (encoder statement only at unit level): size, block position, parameter
    pcases, pout = [], []
    for size in (16, 8, 4):
        for (x, y) in [(16, 16), (32, 8), (size, size)]:
            for param in (-9, -1, 0, 3, 11):
                d = plane.copy()
                o = np.zeros(256, np.uint8)
                r = L.csref_enc_plane(size, d.ctypes.data, d.size, y * S + x, S, param, o.ctypes.data)
                assert r == 0
                pcases.append((size, x, y, param))
                pout.append(o.copy())
    out.update(plane_cases=np.array(pcases, np.int32), plane_out=np.array(pout, np.uint8))
    # ---- inverse transforms: dequantised-looking coefficient blocks, sparse and dense, and a random prediction
    icoef8, ipred8, iout8, irc8 = [], [], [], []
    for k in range(48):
        c = np.zeros(64, np.int32)
        nnz = [1, 1, 3, 6, 16, 64][k % 6]
        pos = rng.choice(64 if nnz > 16 else 16, size=min(nnz, 16 if nnz <= 16 else 64), replace=False) if nnz < 64 else np.arange(64)
        if k % 6 == 0:
            pos = np.array([0])
        c[pos] = rng.integers(-40, 41, len(pos)) * rng.choice([16, 20, 26, 40, 64, 104], len(pos))
        p = rng.integers(0, 256, 64, dtype=np.uint8)
        o = np.zeros(64, np.uint8)
        r = L.csref_enc_idct(64, c.ctypes.data, p.ctypes.data, o.ctypes.data)
        icoef8.append(c); ipred8.append(p); iout8.append(o); irc8.append(r)
    icoef4, ipred4, iout4, irc4 = [], [], [], []
    for k in range(32):
        c = np.zeros(16, np.int32)
        nnz = [1, 2, 5, 16][k % 4]
        pos = rng.choice(16, size=nnz, replace=False) if k % 4 else np.array([0])
        c[pos] = rng.integers(-40, 41, len(pos)) * rng.choice([40, 52, 64, 80, 104, 160], len(pos))
        p = rng.integers(0, 256, 16, dtype=np.uint8)
        o = np.zeros(16, np.uint8)
        r = L.csref_enc_idct(16, c.ctypes.data, p.ctypes.data, o.ctypes.data)
        icoef4.append(c); ipred4.append(p); iout4.append(o); irc4.append(r)
    out.update(idct8_coef=np.array(icoef8), idct8_pred=np.array(ipred8), idct8_out=np.array(iout8), idct8_rc=np.array(irc8, np.int32),
               idct4_coef=np.array(icoef4), idct4_pred=np.array(ipred4), idct4_out=np.array(iout4), idct4_rc=np.array(irc4, np.int32))
    # ---- CopyBlock: every phase, every leaf size of the partition tree
    src = rng.integers(0, 256, S * ROWS, dtype=np.uint8)
    ccases, cout = [], []
    for (w, h) in [(16, 16), (16, 8), (8, 16), (8, 8), (4, 8), (8, 4), (4, 4), (2, 4), (4, 2), (2, 2), (16, 2), (2, 16)]:
        for (dx, dy) in [(0, 0), (5, 0), (0, 7), (-3, -5), (9, 11), (-8, 6)]:
            o = np.zeros(256, np.uint8)
            r = L.csref_enc_getpblock(src.ctypes.data, src.size, dx, dy, w, h, 16 * S + 32, S, o.ctypes.data)
            assert r == 0
            ccases.append((w, h, dx, dy))
            cout.append(o.copy())
    out.update(copy_src=src, copy_cases=np.array(ccases, np.int32), copy_out=np.array(cout, np.uint8), copy_offset=np.int32(16 * S + 32))
    # ---- forward transforms (Encoder/MobiEncoder.cs DCT64, DCT16): residual-like inputs
    f8 = rng.integers(-255, 256, (24, 64)).astype(np.int32)
    f8[0] = 0; f8[1] = 255; f8[2] = -255
    f4 = rng.integers(-255, 256, (24, 16)).astype(np.int32)
    f4[0] = 0; f4[1] = 255; f4[2] = -255
    o8 = np.zeros_like(f8); o4 = np.zeros_like(f4)
    for k in range(len(f8)):
        assert L.csref_enc_dct(64, f8[k].ctypes.data, o8[k].ctypes.data) == 0
        assert L.csref_enc_dct(16, f4[k].ctypes.data, o4[k].ctypes.data) == 0
    out.update(dct8_in=f8, dct8_out=o8, dct4_in=f4, dct4_out=o4)
    np.savez_compressed(OUT, **out)
    print(OUT, os.path.getsize(OUT), "bytes;", len(cases), "intra cases,", n_both, "of them stated twice (decoder + encoder) and equal;",
          int((rc != 0).sum()), "throw in the decoder")


def write_text(path=os.path.join(ROOT, "tests", "golden", "unit_vectors.txt")):
    """unit_vectors.npz as plain lines, for tests/golden/verify/VerifyUnits.cs (a C# program against the unmodified reference): byte arrays as
    hex, int arrays comma-separated.  Needs nothing but the committed .npz."""
    d = np.load(OUT)
    hx = lambda a: np.asarray(a, np.uint8).tobytes().hex()
    cs = lambda a: ",".join(str(int(v)) for v in np.asarray(a).ravel())
    with open(path, "w") as f:
        f.write("# written by tests/golden/make_unit_vectors.py --text from unit_vectors.npz; read by tests/golden/verify/VerifyUnits.cs\n")
        f.write(f"stride {S} rows {ROWS}\n")
        for k in range(len(d["idct8_rc"])):
            f.write(f"idct8 {int(d['idct8_rc'][k])} {cs(d['idct8_coef'][k])} {hx(d['idct8_pred'][k])} {hx(d['idct8_out'][k])}\n")
        for k in range(len(d["idct4_rc"])):
            f.write(f"idct4 {int(d['idct4_rc'][k])} {cs(d['idct4_coef'][k])} {hx(d['idct4_pred'][k])} {hx(d['idct4_out'][k])}\n")
        for k in range(len(d["dct8_in"])):
            f.write(f"dct8 {cs(d['dct8_in'][k])} {cs(d['dct8_out'][k])}\n")
        for k in range(len(d["dct4_in"])):
            f.write(f"dct4 {cs(d['dct4_in'][k])} {cs(d['dct4_out'][k])}\n")
        f.write(f"copysrc {int(d['copy_offset'])} {hx(d['copy_src'])}\n")
        for (w, h, dx, dy), o in zip(d["copy_cases"], d["copy_out"]):
            f.write(f"copy {w} {h} {dx} {dy} {hx(o[: w * h])}\n")
        f.write(f"intraplane {hx(d['intra_plane'])}\n")
        for (m, x, y, uv), pr, rc in zip(d["intra_cases"], d["intra_pred"], d["intra_rc"]):
            n = 8 if m < 10 else 4
            f.write(f"intra {m} {x} {y} {uv} {int(rc)} {hx(pr[: n * n])}\n")
        for (size, x, y, param), o in zip(d["plane_cases"], d["plane_out"]):
            f.write(f"plane {size} {x} {y} {param} {hx(o[: size * size])}\n")


if __name__ == "__main__":
    if "--text" in sys.argv:
        write_text()
    else:
        main()
        write_text()
