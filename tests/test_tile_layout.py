"""The private plane layout (mobiclipdecoder_amd/csrc/mobi_tile.h): a bijection of the reference's linear plane offsets that makes a
macroblock's samples contiguous.  CPU only: the same inline functions the kernels use, compiled into the test tool."""
import ctypes as C

import numpy as np
import pytest

from tests.interp_binding import lib


@pytest.mark.parametrize("S,H", [(256, 192), (256, 16), (512, 48), (1024, 480), (1024, 32)])
def test_tile_maps_are_bijections_with_contiguous_macroblocks(S, H):
    L = lib()
    L.mobi_test_ty.restype = C.c_uint32
    L.mobi_test_ty.argtypes = [C.c_uint32, C.c_int]
    L.mobi_test_tc.restype = C.c_uint32
    L.mobi_test_tc.argtypes = [C.c_uint32, C.c_int]
    lg = S.bit_length() - 1
    ysz = S * H
    a = np.arange(ysz, dtype=np.uint32)
    row, col = a >> lg, a & (S - 1)
    # luma: the documented formula, vectorised; spot-checked against the compiled function
    ty = ((((row >> 4) * (S >> 4) + (col >> 4)) << 8) + (((row >> 3) & 1) << 7) + (((col >> 3) & 1) << 6) + ((row & 7) << 3) + (col & 7)).astype(np.int64)
    for k in np.random.default_rng(1).integers(0, ysz, 2000):
        assert L.mobi_test_ty(int(k), lg) == ty[k]
    assert np.array_equal(np.sort(ty), np.arange(ysz))  # a bijection of [0, Stride * Height): padding columns included
    # a macroblock = 256 contiguous bytes, a quadrant = 64, two rows of a quadrant = one 16-byte chunk
    t = ty.reshape(H, S)
    mb = t[16:32, 32:48] if H >= 32 else t[0:16, 32:48]
    assert mb.max() - mb.min() == 255 and mb.min() % 256 == 0
    q = mb[8:16, 0:8]
    assert q.max() - q.min() == 63 and q.min() % 64 == 0
    assert np.array_equal(mb[2:4, 8:16].ravel(), mb[2, 8] + np.arange(16))
    # eight macroblocks of an octet are one run
    assert t[0:16, 0:128].max() - t[0:16, 0:128].min() == 2047
    # chroma: U in columns [0, S/2), V in [S/2, S); a macroblock = 128 bytes = 8 rows of [U 8 | V 8]
    csz = ysz // 2
    c = np.arange(csz, dtype=np.uint32)
    crow, ccol = c >> lg, c & (S - 1)
    v, x = ccol >> (lg - 1), ccol & (S // 2 - 1)
    tc = ((((crow >> 3) * (S >> 4) + (x >> 3)) << 7) + ((crow & 7) << 4) + (v << 3) + (x & 7)).astype(np.int64)
    for k in np.random.default_rng(2).integers(0, csz, 2000):
        assert L.mobi_test_tc(int(k), lg) == tc[k]
    assert np.array_equal(np.sort(tc), np.arange(csz))
    tt = tc.reshape(H // 2, S)
    u_mb, v_mb = tt[0:8, 8:16], tt[0:8, S // 2 + 8: S // 2 + 16]
    both = np.concatenate([u_mb.ravel(), v_mb.ravel()])
    assert both.max() - both.min() == 127 and both.min() % 128 == 0
    assert np.array_equal(v_mb - u_mb, np.full((8, 8), 8))  # the V sample sits 8 bytes behind the U sample of the same place
