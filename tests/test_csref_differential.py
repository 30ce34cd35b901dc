"""CPU, build container only: the hand-written oracle against the mechanical C#->C++ transliteration of the
reference (oracle/tools/cs2cpp.py -> oracle/_ref/libmobi_csref.so).  Skipped wherever that library cannot be
produced (no /root/reference, e.g. on the GPU box).  Bit-exact planes, Offset, Quantizer, and agreement on
which frames throw -- on valid streams of every configuration and on randomly corrupted ones."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from mobiclipdecoder_amd import default_params, generate_clip
from mobiclipdecoder_amd.streamgen import BASE_SEED
from tests.oracle_binding import OracleDecoder

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_ref", "libmobi_csref.so")


def _lib():
    gen = os.path.join(ROOT, "oracle", "tools", "cs2cpp.py")
    if not os.path.exists(SO) or (os.path.isdir("/root/reference") and os.path.getmtime(SO) < os.path.getmtime(gen)):
        if not os.path.isdir("/root/reference"):
            pytest.skip("reference tree absent: the transliteration can only be generated in the build container")
        subprocess.check_call([sys.executable, os.path.join(ROOT, "oracle", "tools", "cs2cpp.py")])
    L = C.CDLL(SO)
    L.csref_create.restype = C.c_void_p
    L.csref_create.argtypes = [C.c_uint, C.c_uint, C.c_int]
    L.csref_destroy.argtypes = [C.c_void_p]
    L.csref_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.POINTER(C.c_int)]
    L.csref_stride.argtypes = [C.c_void_p]
    L.csref_quantizer.argtypes = [C.c_void_p]
    L.csref_quantizer.restype = C.c_uint
    for n in ("csref_y", "csref_uv"):
        getattr(L, n).restype = C.POINTER(C.c_uint8)
        getattr(L, n).argtypes = [C.c_void_p, C.c_int]
    return L


def _diff(L, params, data, fo, whole=False):
    """decode every frame with both; return (#frames ok, #frames thrown)"""
    h = L.csref_create(params.width, params.height, params.version)
    o = OracleDecoder(params.width, params.height, params.version)
    S, H = o.Stride, params.height
    ok = thrown = 0
    for f in range(params.n_frames):
        buf = np.ascontiguousarray(data if whole else data[: fo[f + 1]])
        off = C.c_int(int(fo[f]))
        rc = L.csref_decode(h, buf.ctypes.data, buf.size, C.byref(off))
        o.Data, o.Offset = buf, int(fo[f])
        ro = o.DecodeFrame()
        assert (rc != 0) == (ro is None), (f, rc, o.last_error)
        assert off.value == o.Offset, (f, off.value, o.Offset)
        assert L.csref_quantizer(h) == o.Quantizer, f
        # the frame slot is compared even after a throw: partial frames stay in the ring (MD.cs:325)
        for r in range(min(6, f + 1)):
            py, oy = L.csref_y(h, r), o.y(r)
            assert bool(py) == (oy is not None)
            if oy is not None:
                assert np.array_equal(np.ctypeslib.as_array(py, (H, S)), oy), (f, r, "Y")
                assert np.array_equal(np.ctypeslib.as_array(L.csref_uv(h, r), (H // 2, S)), o.uv(r)), (f, r, "UV")
        if rc == 0:
            ok += 1
        else:
            thrown += 1
    L.csref_destroy(h)
    return ok, thrown


@pytest.mark.parametrize("cfg", ["A", "B", "C"])
def test_valid_streams(cfg):
    L = _lib()
    for seed in range(4):
        p = default_params(cfg, BASE_SEED + 500 + seed, n_frames=9, pm_intra=150, pm_deep=150, pm_multiref=300,
                           qdelta_prob=300, table1_prob=500, escape_prob=100, edge_mode=seed & 1,
                           iframe_interval=5 if seed == 3 else 0)
        data, fo = generate_clip(p)
        ok, thrown = _diff(L, p, data, fo, whole=bool(seed & 2))
        assert ok == 9 and thrown == 0


def test_small_geometries_and_quantizers():
    L = _lib()
    for i, (w, h, ver, q) in enumerate([(16, 16, 2, 12), (32, 16, 1, 52), (256, 32, 1, 30), (512, 32, 2, 40), (528, 48, 2, 25), (1024, 32, 2, 18)]):
        p = default_params("A", BASE_SEED + 900 + i, n_frames=8, width=w, height=h, version=ver, quantizer=q, pm_intra=200, mv_range=12)
        data, fo = generate_clip(p)
        assert _diff(L, p, data, fo) == (8, 0), (w, h)


def test_corrupted_streams_including_partial_frames():
    """Bit flips: decode garbage or throw -- both executables must do the same thing, down to the partial
    frame a throw leaves behind and the decoder state it leaks into the following frames."""
    L = _lib()
    rng = np.random.default_rng(11)
    tot_ok = tot_thrown = 0
    for trial in range(150):
        p = default_params("AB"[trial & 1], BASE_SEED + 2000 + trial, n_frames=5, pm_intra=100, width=64, height=48,
                           version=1 + (trial & 1), pm_deep=200)
        data, fo = generate_clip(p)
        data = data.copy()
        for _ in range(int(rng.integers(1, 5))):
            data[int(rng.integers(0, data.size))] ^= 1 << int(rng.integers(0, 8))
        ok, thrown = _diff(L, p, data, fo, whole=bool(trial & 2))
        tot_ok += ok
        tot_thrown += thrown
    assert tot_ok > 100 and tot_thrown > 30, (tot_ok, tot_thrown)


def test_low_quantizer_aliasing_domain():
    """ModsDS quantizers below 12 make the dequant word leak into its zigzag byte (MD.cs:3907-3911): the product
    refuses those streams, but the oracle must still follow the reference through the Internal[] aliasing."""
    L = _lib()
    for q in (0, 3, 7, 11):
        p = default_params("A", BASE_SEED + 3000 + q, n_frames=4, width=64, height=48, quantizer=12)
        data, fo = generate_clip(p)
        data = data.copy()
        # I-frame header: bit0=1, yuv, table, then 6-bit quantizer in bits 12..7 of the first 16-bit LE word
        w = int(data[0]) | (int(data[1]) << 8)
        w = (w & ~(0x3F << 7)) | (q << 7)
        data[0], data[1] = w & 0xFF, w >> 8
        _diff(L, p, data, fo)


@pytest.mark.parametrize("cfg", ["A", "B", "C"])
def test_bench_streams(cfg):
    """The exact streams bench.py times (mobiclipdecoder_amd.sharding.stream_seed(config, rank 0, index 0..3), 1 I + 32 P
    frames, SURVEY 8(d) mix) and the streams of the GPU coverage suite: oracle and transliteration agree on every frame."""
    from mobiclipdecoder_amd import sharding
    L = _lib()
    for i in range(4 if cfg == "B" else 2):
        p = default_params(cfg, sharding.stream_seed(cfg, 0, i), n_frames=33)
        data, fo = generate_clip(p)
        ok, thrown = _diff(L, p, data, fo)
        assert ok == 33 and thrown == 0


def test_coverage_suite_streams():
    from tests.gpu_streams import suite_params
    L = _lib()
    for p in suite_params():
        data, fo = generate_clip(p)
        ok, thrown = _diff(L, p, data, fo)
        assert ok == p.n_frames and thrown == 0


def test_unit_statements_random_sweep():
    """The oracle's unit entry points against BOTH reference statements of each unit (decoder PredictIntra; the encoder's GetCompvals /
    PredictIntraPlane / IDCT / FrameUtil.GetPBlock copies) on thousands of random inputs -- the committed
    tests/golden/unit_vectors.npz is a small sample of this sweep that travels to the GPU box."""
    from tests.oracle_binding import lib as oracle_lib
    L = _lib()
    OL = oracle_lib()
    L.csref_dec_predict.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_int]
    L.csref_enc_compvals.argtypes = [C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.csref_enc_plane.argtypes = [C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.csref_enc_idct.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.csref_enc_getpblock.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_void_p]
    rng = np.random.default_rng(2024)
    S, ROWS = 256, 40
    dec = L.csref_create(256, 192, 2)
    stated_twice = 0
    for trial in range(3000):
        plane = rng.integers(0, 256, S * ROWS, dtype=np.uint8)
        four = bool(trial & 1)
        m = int(rng.choice([0, 1, 3, 4, 5, 6, 7, 8])) + (10 if four else 0)
        n = 4 if four else 8
        x = int(rng.choice([0, n, 2 * n, 128, 128 + n, 64])); y = int(rng.choice([0, n, 2 * n, 24]))
        uv = int(x >= 128 or (trial % 5 == 0))
        off = y * S + x
        a, b = plane.copy(), plane.copy()
        ra = OL.mobi_oracle_predict(m, a.ctypes.data, a.size, off, S, uv)
        rb = L.csref_dec_predict(dec, m, b.ctypes.data, b.size, off, uv)
        assert (ra != 0) == (rb != 0), (m, x, y, uv)
        if rb == 0:
            assert np.array_equal(a, b), (m, x, y, uv)
            eoff = S // 2 if (uv and x >= S // 2) else 0
            e, eo = plane.copy(), np.zeros(64, np.uint8)
            if L.csref_enc_compvals(m, e.ctypes.data, e.size, x - eoff, y, S, eoff, eo.ctypes.data) == 0:
                assert np.array_equal(eo[: n * n], a.reshape(ROWS, S)[y:y + n, x:x + n].ravel()), ("encoder statement", m, x, y, uv)
                stated_twice += 1
        # planes
        size = int(rng.choice([16, 8, 4])); px = int(rng.choice([size, 2 * size, 48])); py = int(rng.choice([size, 16])); param = int(rng.integers(-40, 41))
        a, e, eo = plane.copy(), plane.copy(), np.zeros(256, np.uint8)
        assert OL.mobi_oracle_plane(size, param, a.ctypes.data, a.size, py * S + px, S) == 0
        assert L.csref_enc_plane(size, e.ctypes.data, e.size, py * S + px, S, param, eo.ctypes.data) == 0
        assert np.array_equal(a.reshape(ROWS, S)[py:py + size, px:px + size].ravel(), eo[: size * size]), (size, px, py, param)
        # CopyBlock
        w, h = int(rng.choice([2, 4, 8, 16])), int(rng.choice([2, 4, 8, 16]))
        dx, dy = int(rng.integers(-20, 21)), int(rng.integers(-20, 21))
        dst, eo = np.zeros(S * ROWS, np.uint8), np.zeros(256, np.uint8)
        base = 12 * S + 32
        assert OL.mobi_oracle_copyblock(plane.ctypes.data, plane.size, dx, dy, w, h, dst.ctypes.data, dst.size, base, S) == 0
        assert L.csref_enc_getpblock(plane.ctypes.data, plane.size, dx, dy, w, h, base, S, eo.ctypes.data) == 0
        assert np.array_equal(dst.reshape(ROWS, S)[12:12 + h, 32:32 + w].ravel(), eo[: w * h]), (w, h, dx, dy)
        # inverse transforms (encoder statement = the full transform)
        nn = 16 if four else 64
        c = np.zeros(nn, np.int32)
        k = int(rng.integers(1, nn + 1))
        pos = rng.choice(nn, size=k, replace=False)
        c[pos] = rng.integers(-60, 61, k) * int(rng.choice([16, 40, 64, 104, 160]))
        p = rng.integers(0, 256, nn, dtype=np.uint8)
        eo = np.zeros(nn, np.uint8)
        re_ = L.csref_enc_idct(nn, c.ctypes.data, p.ctypes.data, eo.ctypes.data)
        side = 4 if four else 8
        d = np.zeros(side * 16, np.uint8)
        d.reshape(side, 16)[:, :side] = p.reshape(side, side)
        fn = OL.mobi_oracle_idct4 if four else OL.mobi_oracle_idct8
        ro = fn(c.ctypes.data, nn, d.ctypes.data, d.size, 0, 16)
        assert (ro != 0) == (re_ != 0)
        if re_ == 0:
            assert np.array_equal(d.reshape(side, 16)[:, :side].ravel(), eo)
    assert stated_twice > 1500
