"""Golden fixtures (tests/golden/): committed bitstreams + SHA-256 of every output plane.
CPU: the oracle reproduces them.  GPU: the HIP path (through the C ABI) reproduces them."""
import hashlib
import json
import os

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAN = json.load(open(os.path.join(HERE, "golden.json")))


def _run(case, dec):
    data = np.fromfile(os.path.join(HERE, case["name"] + ".bin"), dtype=np.uint8)
    assert data.size == case["bytes"]
    fo = case["frame_off"]
    for f, exp in enumerate(case["frames"]):
        dec.Data, dec.Offset = data[: fo[f + 1]], fo[f]
        r = dec.DecodeFrame()
        assert r is not None, (case["name"], f, dec.last_error)
        assert dec.Offset == exp["offset_after"] and dec.Quantizer == exp["quantizer"], (case["name"], f)
        assert hashlib.sha256(np.ascontiguousarray(r[0]).tobytes()).hexdigest() == exp["y_sha256"], (case["name"], f, "Y")
        assert hashlib.sha256(np.ascontiguousarray(r[1]).tobytes()).hexdigest() == exp["uv_sha256"], (case["name"], f, "UV")


@pytest.mark.parametrize("case", MAN["cases"], ids=[c["name"] for c in MAN["cases"]])
def test_oracle_reproduces_golden(case):
    from tests.oracle_binding import OracleDecoder
    _run(case, OracleDecoder(case["width"], case["height"], case["version"]))


@pytest.mark.parametrize("case", MAN["cases"], ids=[c["name"] for c in MAN["cases"]])
def test_interpreter_reproduces_golden(case):
    from tests.interp_binding import InterpDecoder
    _run(case, InterpDecoder(case["width"], case["height"], case["version"]))


@pytest.mark.gpu
@pytest.mark.parametrize("case", MAN["cases"], ids=[c["name"] for c in MAN["cases"]])
def test_hip_reproduces_golden(case):
    from mobiclipdecoder_amd import MobiclipDecoder
    d = MobiclipDecoder(case["width"], case["height"], case["version"])
    assert d.Stride == case["stride"]
    _run(case, d)
    d.close()


def test_text_manifest_for_the_reference_side_check_matches_golden_json():
    """tests/golden/verify/VerifyGolden.cs (a C# program against the unmodified reference; it cannot run here) reads golden_manifest.txt:
    the same content as golden.json, line by line."""
    g = MAN
    lines = [ln.split() for ln in open(os.path.join(HERE, "golden_manifest.txt")) if not ln.startswith("#")]
    it = iter(lines)
    for c in g["cases"]:
        assert next(it) == ["case", c["name"], str(c["width"]), str(c["height"]), str(c["version"]), str(len(c["frames"]))]
        assert next(it) == ["covers"] + c["covers"].split()  # what the fixture exercises (the oracle's coverage counters): what a green run pins
        for i, fr in enumerate(c["frames"]):
            assert next(it) == ["frame", str(c["frame_off"][i]), str(c["frame_off"][i + 1]), fr["y_sha256"], fr["uv_sha256"], str(fr["offset_after"]), str(fr["quantizer"])]
    assert next(it, None) is None
    src = open(os.path.join(HERE, "verify", "VerifyGolden.cs")).read()
    assert "new MobiclipDecoder(" in src and "DecodeFrame()" in src and "golden_manifest.txt" in src
    # one stream per class of input r01-r04 refused and r05 decodes (walks through Internal[], the transforms' scratch, quantisers below 12,
    # plane parameters and motion vectors beyond the command list's old fields) is part of what the reference-side check pins
    names = {c["name"] for c in g["cases"]}
    assert {"r05_walk_mods_64x48", "r05_scratch_moflex_64x48", "r05_lowq_mods_64x48", "r05_wide_plane_mods_32x32", "r05_far_mv_moflex_32x32"} <= names
