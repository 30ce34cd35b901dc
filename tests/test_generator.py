"""CPU: the synthetic stream generator is deterministic, and everything it writes parses back
(oracle decodes every frame without an exception and lands exactly on the frame boundary)."""
import numpy as np
import pytest

from mobiclipdecoder_amd import default_params, generate_clip
from mobiclipdecoder_amd.streamgen import BASE_SEED
from tests.oracle_binding import OracleDecoder


def test_deterministic():
    a = generate_clip(default_params("A", BASE_SEED, n_frames=5))
    b = generate_clip(default_params("A", BASE_SEED, n_frames=5))
    c = generate_clip(default_params("A", BASE_SEED + 1, n_frames=5))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[0].size != c[0].size or not np.array_equal(a[0], c[0])


@pytest.mark.parametrize("cfg", ["A", "B", "C"])
def test_round_trip_lands_on_frame_boundaries(cfg):
    p = default_params(cfg, BASE_SEED + 11, n_frames=12, escape_prob=200, pm_deep=200, pm_intra=100, qdelta_prob=400, table1_prob=500)
    data, fo = generate_clip(p)
    assert np.all(np.diff(fo.astype(np.int64)) > 0) and fo[-1] == data.size and data.size % 2 == 0
    o = OracleDecoder(p.width, p.height, p.version)
    w = OracleDecoder(p.width, p.height, p.version)
    for f in range(p.n_frames):
        o.Data, o.Offset = data[: fo[f + 1]], int(fo[f])       # Data ends with the frame: no refill past it (MD.cs:2990)
        assert o.DecodeFrame() is not None, (f, o.last_error)
        assert o.Offset == fo[f + 1], f
        w.Data, w.Offset = data, int(fo[f])                    # MOC5 style: whole file; FillBits may read one word ahead
        assert w.DecodeFrame() is not None
        assert fo[f + 1] <= w.Offset <= fo[f + 1] + 2, f
        assert np.array_equal(o.y(0), w.y(0))
    y = o.y(0)[:, : p.width]
    assert y.std() > 10  # textured, not flat


def test_bad_parameters_rejected():
    for kw in ({"width": 100}, {"height": 8}, {"quantizer": 5}, {"version": 0}, {"n_frames": 0}):
        with pytest.raises(ValueError):
            generate_clip(default_params("A", BASE_SEED, **kw))
