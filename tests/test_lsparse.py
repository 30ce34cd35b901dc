"""The lock-step parser (mobiclipdecoder_amd/csrc/mobi_lsparse.h: one clip per lane, mobi_parse_frames_ls on the GPU) run lane by lane on the
CPU (tests/tools/mobi_lsparse_host.cpp) against the host parser on the same frames.

Whenever it does not bail out, everything it leaves must be what the host parser leaves: every descriptor word, every payload word, the
intra list, the dependency sets, the consumed bytes, the quantiser and the state that survives the frame.  Whenever the host parser
reports anything but MOBI_OK it must have bailed out (mobi_parse_frames then parses that clip); on intact streams it must never bail out."""
import ctypes as C
import os

import numpy as np
import pytest

from mobiclipdecoder_amd import default_params, generate_clip
from mobiclipdecoder_amd.streamgen import BASE_SEED

from tests.gpu_streams import COVERAGE_SUITE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from mobiclipdecoder_amd import build
    L = C.CDLL(build.build_lshost())
    L.mobi_lshost_compare.argtypes = [C.c_uint, C.c_uint, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    return L


def compare(L, p, data, fo, allow_bail):
    fo = np.ascontiguousarray(fo, dtype=np.uint32)
    data = np.ascontiguousarray(data, dtype=np.uint8)
    bails = C.c_int(0)
    n = len(fo) - 1
    r = L.mobi_lshost_compare(p.width, p.height, p.version, data.ctypes.data, fo.ctypes.data, n, allow_bail, C.byref(bails))
    return r, bails.value, n


@pytest.mark.parametrize("idx", range(len(COVERAGE_SUITE)))
def test_coverage_suite_streams_parse_identically(lib, idx):
    cfg, seed, kw = COVERAGE_SUITE[idx]
    p = default_params(cfg, BASE_SEED + seed, **kw)
    d, fo = generate_clip(p)
    r, bails, n = compare(lib, p, d, fo, 0)
    assert r == n and bails == 0, (r, bails)


def test_geometries_versions_quantisers_and_edge_vectors(lib):
    for i, (w, h, ver, q) in enumerate([(16, 16, 1, 12), (64, 48, 1, 25), (64, 48, 2, 52), (256, 192, 2, 40), (512, 64, 2, 30), (1024, 32, 2, 20), (272, 160, 1, 16)]):
        p = default_params("A", BASE_SEED + 900 + i, n_frames=8, width=w, height=h, version=ver, quantizer=q, pm_intra=200, mv_range=12, iframe_interval=5)
        d, fo = generate_clip(p)
        r, bails, n = compare(lib, p, d, fo, 0)
        assert r == n and bails == 0, (w, h, ver, q, r, bails)
    for i in range(4):  # vectors that reach the padding and wrap rows (MD.cs:414-423)
        p = default_params("A", BASE_SEED + 340 + i, n_frames=6, edge_mode=1, mv_range=40)
        d, fo = generate_clip(p)
        r, bails, n = compare(lib, p, d, fo, 0)
        assert r == n and bails == 0, (i, r, bails)


def test_long_gop_uses_every_reference_slot(lib):
    p = default_params("B", BASE_SEED + 330, n_frames=33, pm_multiref=300, pm_intra=60)
    d, fo = generate_clip(p)
    r, bails, n = compare(lib, p, d, fo, 0)
    assert r == n and bails == 0, (r, bails)


@pytest.mark.parametrize("cfg", ["A", "B"])
def test_damaged_streams_bail_out_or_parse_identically(lib, cfg):
    """Bit flips, byte garbage, noise frames, truncations: the lock-step parser may leave a frame to mobi_parse_frames whenever it likes, but
    what it does finish must be the host parser's result, and it must never finish a frame the host parser rejects."""
    nfr = 8
    rng = np.random.default_rng(20260928)
    p = default_params(cfg, BASE_SEED + 500, n_frames=nfr, pm_intra=120, pm_deep=120, pm_multiref=200, qdelta_prob=200, escape_prob=60, iframe_interval=3)
    if cfg == "B":
        p = default_params(cfg, BASE_SEED + 501, n_frames=nfr, width=256, height=192, pm_intra=120, pm_deep=120, pm_multiref=200, qdelta_prob=200, escape_prob=60, iframe_interval=3)
    d0, fo = generate_clip(p)
    total_bails = finished = 0
    for trial in range(160):
        d = np.array(d0, copy=True)
        kind = trial % 6
        if kind == 0:
            for pos in rng.integers(0, d.size, 3):
                d[pos] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            for pos in rng.integers(0, d.size, 40):
                d[pos] ^= 1 << int(rng.integers(0, 8))
        elif kind == 2:
            f = int(rng.integers(0, nfr))
            a = int(rng.integers(fo[f], fo[f + 1]))
            d[a:a + 64] = rng.integers(0, 256, d[a:a + 64].size, dtype=np.uint8)
        elif kind == 3:
            f = int(rng.integers(0, nfr))
            d[fo[f] + 2:fo[f + 1]] = rng.integers(0, 256, int(fo[f + 1] - fo[f] - 2), dtype=np.uint8)
        elif kind == 4:  # every frame cut short by a few bytes (odd and even lengths): the data ends inside the frame
            cut = int(rng.integers(1, 9))
            parts, offs = [], [0]
            for f in range(nfr):
                fr = d[fo[f]:max(int(fo[f]) + 2, int(fo[f + 1]) - cut)]
                parts.append(fr)
                offs.append(offs[-1] + fr.size)
            r, bails, n = compare(lib, p, np.concatenate(parts), offs, 1)
            assert r == n, (trial, r)
            total_bails += bails
            finished += n - bails
            continue
        r, bails, n = compare(lib, p, d, fo, 1)
        assert r == n, (trial, kind, r)
        total_bails += bails
        finished += n - bails
    assert total_bails > 50 and finished > 300, (total_bails, finished)  # both outcomes were exercised


def test_random_generator_mixes(lib):
    """48 random points of the generator's parameter space (macroblock mix, motion range, residual density and shape, escapes, both VLC tables,
    quantiser deltas, I-frame intervals, both versions, several geometries): intact streams are never handed over and always identical."""
    rng = np.random.default_rng(0x4C53)
    for trial in range(48):
        w, h = [(64, 48), (128, 96), (256, 192), (320, 240), (512, 32), (16, 144)][trial % 6]
        kw = dict(width=w, height=h, version=1 + trial % 2, n_frames=int(rng.integers(3, 9)), quantizer=int(rng.integers(12, 53)),
                  pm_skip=int(rng.integers(0, 400)), pm_split1=int(rng.integers(0, 300)), pm_deep=int(rng.integers(0, 250)), pm_intra=int(rng.integers(0, 300)),
                  pm_multiref=int(rng.integers(0, 500)), mv_range=int(rng.integers(0, 40)), cbp_prob=int(rng.integers(0, 1000)), t8_prob=int(rng.integers(0, 1000)),
                  dense_prob=int(rng.integers(0, 400)), max_coefs=int(rng.integers(1, 12)), scan_span=int(rng.integers(1, 64)),
                  intra_sub_prob=int(rng.integers(0, 1000)), plane_prob=int(rng.integers(0, 700)), escape_prob=int(rng.integers(0, 400)),
                  qdelta_prob=int(rng.integers(0, 600)), table1_prob=int(rng.integers(0, 1000)), iframe_interval=int(rng.integers(0, 5)))
        if kw["pm_skip"] + kw["pm_split1"] + kw["pm_deep"] + kw["pm_intra"] > 1000:
            kw["pm_skip"] = 0
        p = default_params("A", BASE_SEED + 8000 + trial, **kw)
        d, fo = generate_clip(p)
        r, bails, n = compare(lib, p, d, fo, 0)
        assert r == n and bails == 0, (trial, kw, r, bails)
