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
    if not os.path.exists(SO):
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
