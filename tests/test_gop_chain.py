"""The frame-parallel chain (mobiclipdecoder_amd/csrc/mobi_gop.h) on the CPU: the start state of every frame of a group predicted from the
frame HEADERS alone, every frame parsed from its predicted state by the lock-step parser's lane functions, the chain verified and merged by
the functions mobi_gop_prepare / mobi_gop_chain call on the device (tests/tools/mobi_lsparse_host.cpp: mobi_gop_host_check) -- against the
host parser, which parses in stream order.  No difference is allowed anywhere; on intact streams no prediction may fail (a failed
prediction only costs time -- the host parser takes the clip -- but it would mean the header chain is not understood)."""
import ctypes as C

import numpy as np
import pytest

from mobiclipdecoder_amd import build, default_params, generate_clip
from mobiclipdecoder_amd.streamgen import BASE_SEED
from tests.gpu_streams import COVERAGE_SUITE


@pytest.fixture(scope="module")
def lib():
    L = C.CDLL(build.build_lshost())
    L.mobi_gop_host_check.argtypes = [C.c_uint, C.c_uint, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_long * 5)]
    return L


def check(L, p, data, fo, K):
    fo = np.ascontiguousarray(fo, dtype=np.uint32)
    data = np.ascontiguousarray(data, dtype=np.uint8)
    st = (C.c_long * 5)()
    assert L.mobi_gop_host_check(p.width, p.height, p.version, data.ctypes.data, fo.ctypes.data, len(fo) - 1, K, C.byref(st)) == 0
    return list(st)


@pytest.mark.parametrize("K", [2, 6])
@pytest.mark.parametrize("idx", range(len(COVERAGE_SUITE)))
def test_coverage_suite_streams_chain(lib, idx, K):
    cfg, seed, kw = COVERAGE_SUITE[idx]
    p = default_params(cfg, BASE_SEED + seed, **dict(kw, n_frames=13))
    ok, wrong_guess, unfinished, host, diff = check(lib, p, *generate_clip(p), K)
    assert (ok, wrong_guess, unfinished, host, diff) == (13, 0, 0, 0, 0)


def test_header_chain_with_quantiser_deltas_and_iframes_everywhere(lib):
    """quantiser deltas in most P-frames (clamped at 12 and 52 on Moflex3DS, MD.cs:3886-3889), I-frames at every interval, both versions,
    several geometries; ModsDS streams that start below quantiser 12 or run into it are the host parser's for those frames"""
    rng = np.random.default_rng(0x474F50)
    total = np.zeros(5, np.int64)
    for trial in range(60):
        w, h = [(64, 48), (128, 96), (256, 192), (320, 240), (512, 32), (16, 144)][trial % 6]
        p = default_params("A", BASE_SEED + 8800 + trial, width=w, height=h, version=1 + trial % 2, n_frames=int(rng.integers(6, 20)), quantizer=int(rng.integers(12, 53)),
                           qdelta_prob=int(rng.integers(300, 1000)), iframe_interval=int(rng.integers(0, 6)), pm_intra=int(rng.integers(0, 400)),
                           intra_sub_prob=int(rng.integers(0, 1000)), table1_prob=int(rng.integers(0, 1000)), pm_multiref=int(rng.integers(0, 400)))
        st = np.array(check(lib, p, *generate_clip(p), 1 + trial % 6))
        assert st[4] == 0 and st[1] == 0, (trial, st)
        total += st
    assert total[0] > 500 and total[4] == 0


def test_damaged_streams_never_differ(lib):
    """bit flips: predictions may fail and lanes may bail out (the host parser takes over), but what the chain accepts is the truth"""
    rng = np.random.default_rng(0x474F51)
    total = np.zeros(5, np.int64)
    for trial in range(300):
        p = default_params("AB"[trial % 2], BASE_SEED + 9000 + trial, n_frames=8, width=96, height=64, version=1 + trial % 2, pm_intra=120, pm_deep=150,
                           pm_multiref=250, qdelta_prob=250, escape_prob=80, table1_prob=400, iframe_interval=4)
        data, fo = generate_clip(p)
        d = data.copy()
        for _ in range(int(rng.integers(1, 6))):
            d[int(rng.integers(0, d.size))] ^= 1 << int(rng.integers(0, 8))
        total += np.array(check(lib, p, d, fo, 2 + trial % 5))
    assert total[4] == 0 and total[0] > 800 and total[2] > 50, total  # both outcomes were exercised
