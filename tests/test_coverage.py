"""Syntax coverage accounting (CPU): every stream in this repository comes from our own generator, so something has to say which
parts of the reference's syntax the GPU-tested streams actually reach.  The oracle counts (test-side counters,
oracle/mobi_oracle.h MOBI_COV_*); this test decodes the coverage suite of tests/gpu_streams.py -- the streams
tests/test_gpu_parity.py::test_coverage_suite_streams runs through the HIP path -- and asserts that every item was hit."""
import ctypes as C

import numpy as np

from mobiclipdecoder_amd import generate_clip
from tests import oracle_binding
from tests.gpu_streams import suite_params

PART, INTRA, PLANE, ESCAPE, VLCTAB, REF, PHASE, IDCT, WORDS = 0, 320, 340, 343, 346, 348, 353, 357, 363


def _coverage():
    L = oracle_binding.lib()
    L.mobi_oracle_coverage.argtypes = [C.c_void_p, C.c_int]
    L.mobi_oracle_coverage.restype = None
    L.mobi_oracle_coverage(None, 1)
    for p in suite_params():
        data, fo = generate_clip(p)
        d = oracle_binding.OracleDecoder(p.width, p.height, p.version)
        for f in range(p.n_frames):
            d.Data, d.Offset = data[fo[f]:fo[f + 1]], 0
            assert d.DecodeFrame() is not None, (p.seed, f)
        d.close()
    cov = np.zeros(WORDS, np.uint64)
    L.mobi_oracle_coverage(cov.ctypes.data, 1)
    return cov


def test_gpu_suite_streams_reach_the_whole_syntax():
    cov = _coverage()
    missing = []
    for ver in (0, 1):  # 0 = Moflex3DS tables, 1 = ModsDS tables
        for wi in range(4):
            for hi in range(4):
                s = wi * 4 + hi
                legal = list(range(6)) + ([6, 7] if s == 0 else []) + ([8] if hi < 3 else []) + ([9] if wi < 3 else [])
                for code in legal:  # 8 / 9 = split top-bottom / left-right: not below 2 rows / 2 columns (MD.cs:1683-1746)
                    if cov[PART + (ver * 16 + s) * 10 + code] == 0:
                        missing.append(f"partition {16 >> wi}x{16 >> hi} ver {ver} code {code}")
    for mode in [0, 1, 3, 4, 5, 6, 7, 8, 10, 11, 13, 14, 15, 16, 17, 18]:  # 2 / 12 are the planes, 9 / 19 "already predicted"
        if cov[INTRA + mode] == 0:
            missing.append(f"intra mode {mode}")
    for k, name in enumerate(["plane 16x16", "plane 8x8", "plane 4x4"]):
        if cov[PLANE + k] == 0:
            missing.append(name)
    for k, name in enumerate(["escape: level offset", "escape: run offset", "escape: raw"]):
        if cov[ESCAPE + k] == 0:
            missing.append(name)
    for k in range(2):
        if cov[VLCTAB + k] == 0:
            missing.append(f"VLC table {k}")
    for k in range(5):
        if cov[REF + k] == 0:
            missing.append(f"reference slot {k + 1}")
    for k in range(4):
        if cov[PHASE + k] == 0:
            missing.append(f"CopyBlock phase {k}")
    for k, name in enumerate(["IDCT1Px8", "IDCT3Px8", "IDCT16Px8", "IDCT64Px8", "IDCT1Px4", "IDCT16Px4"]):
        if cov[IDCT + k] == 0:
            missing.append(name)
    assert not missing, missing
