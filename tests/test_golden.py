"""Golden fixtures (tests/golden/): committed bitstreams + SHA-256 of every output plane.
CPU: the oracle reproduces them.  GPU: the HIP path (through the C ABI) reproduces them."""
import hashlib
import json
import os

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAN = json.load(open(os.path.join(HERE, "golden.json")))


def _run(case, dec, partial_picture=False):
    data = np.fromfile(os.path.join(HERE, case["name"] + ".bin"), dtype=np.uint8)
    assert data.size == case["bytes"]
    fo = case["frame_off"]
    for f, exp in enumerate(case["frames"]):
        dec.Data, dec.Offset = data[: fo[f + 1]], fo[f]
        r = dec.DecodeFrame()
        if exp.get("rejected"):
            # a frame the reference returns null for (MD.cs:325-328).  The fixture records the PARTIAL picture the reference keeps (the oracle
            # restates it: that is what pins where the throw happened); this library keeps the slot's old picture instead (mobiclip_hip.h), so
            # for it the check is the rejection itself, Offset at the throw and Quantizer.  (An I-frame follows every such frame in the fixtures.)
            assert r is None, (case["name"], f, "the reference rejects this frame")
            assert dec.Offset == exp["offset_after"] and dec.Quantizer == exp["quantizer"], (case["name"], f, dec.Offset, exp["offset_after"])
            if partial_picture:
                assert hashlib.sha256(np.ascontiguousarray(dec.y(0)).tobytes()).hexdigest() == exp["y_sha256"], (case["name"], f, "partial Y")
                assert hashlib.sha256(np.ascontiguousarray(dec.uv(0)).tobytes()).hexdigest() == exp["uv_sha256"], (case["name"], f, "partial UV")
            continue
        assert r is not None, (case["name"], f, dec.last_error)
        assert dec.Offset == exp["offset_after"] and dec.Quantizer == exp["quantizer"], (case["name"], f)
        assert hashlib.sha256(np.ascontiguousarray(r[0]).tobytes()).hexdigest() == exp["y_sha256"], (case["name"], f, "Y")
        assert hashlib.sha256(np.ascontiguousarray(r[1]).tobytes()).hexdigest() == exp["uv_sha256"], (case["name"], f, "UV")


@pytest.mark.parametrize("case", MAN["cases"], ids=[c["name"] for c in MAN["cases"]])
def test_oracle_reproduces_golden(case):
    from tests.oracle_binding import OracleDecoder
    _run(case, OracleDecoder(case["width"], case["height"], case["version"]), partial_picture=True)


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
            assert next(it) == ["reject" if fr.get("rejected") else "frame", str(c["frame_off"][i]), str(c["frame_off"][i + 1]), fr["y_sha256"], fr["uv_sha256"], str(fr["offset_after"]), str(fr["quantizer"])]
    assert next(it, None) is None
    src = open(os.path.join(HERE, "verify", "VerifyGolden.cs")).read()
    assert "new MobiclipDecoder(" in src and "DecodeFrame()" in src and "golden_manifest.txt" in src
    # one stream per class of input r01-r04 refused and r05 decodes (walks through Internal[], the transforms' scratch, quantisers below 12,
    # plane parameters and motion vectors beyond the command list's old fields) is part of what the reference-side check pins
    names = {c["name"] for c in g["cases"]}
    assert {"r05_walk_mods_64x48", "r05_scratch_moflex_64x48", "r05_lowq_mods_64x48", "r05_wide_plane_mods_32x32", "r05_far_mv_moflex_32x32"} <= names
    # r06: so are frames the reference REJECTS ("reject" lines: the partial picture it keeps, Offset at the throw, Quantizer)
    assert sum(bool(fr.get("rejected")) for c in g["cases"] for fr in c["frames"]) >= 4 and '"reject"' in src


def test_make_golden_selftest():
    """python tests/golden/make_golden.py --selftest: manifest == golden.json, the oracle still gives every recorded frame, and every `covers`
    line is what the oracle's coverage counters say today (what a green run of the pinning kit pins is never stale)"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(HERE, "make_golden.py"), "--selftest"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and " 0 problems" in r.stdout, (r.stdout, r.stderr)
