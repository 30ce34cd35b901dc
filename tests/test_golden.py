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
