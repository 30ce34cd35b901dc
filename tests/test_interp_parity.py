"""CPU: product parser + command list + kernel arithmetic (executed by the CPU interpreter in the GPU's
scheduling order) against the oracle: bit-exact planes, Offset, Quantizer; and error-class agreement on
corrupted streams."""
import numpy as np
import pytest

from mobiclipdecoder_amd import default_params, generate_clip
from mobiclipdecoder_amd.streamgen import BASE_SEED
from tests.interp_binding import InterpDecoder
from tests.oracle_binding import OracleDecoder


def _compare(params, frames=None, mutate=None):
    data, fo = generate_clip(params)
    if mutate is not None:
        data = mutate(data.copy(), fo)
    a = InterpDecoder(params.width, params.height, params.version)
    o = OracleDecoder(params.width, params.height, params.version)
    stats = {"ok": 0, "err": 0, "unsupported": 0}
    for f in range(frames or params.n_frames):
        a.Data = o.Data = data[: fo[f + 1]]
        a.Offset = o.Offset = int(fo[f])
        ra, ro = a.DecodeFrame(), o.DecodeFrame()
        if a.last_error == -6:  # MOBI_E_UNSUPPORTED: documented divergence (Internal[] aliasing domain)
            stats["unsupported"] += 1
            break
        if a.last_error == -5:  # clamp-table domain: found after the parse; oracle throws mid-frame
            assert o.last_error == -1
            stats["err"] += 1
            break
        assert a.last_error == o.last_error, (f, a.last_error, o.last_error)
        if a.last_error != 0:
            stats["err"] += 1
            break  # frame content after a throw is unspecified
        assert a.Offset == o.Offset and a.Quantizer == o.Quantizer, f
        assert np.array_equal(ra[0], ro[0]), (f, np.argwhere(ra[0] != ro[0])[:4].tolist())
        assert np.array_equal(ra[1], ro[1]), (f, np.argwhere(ra[1] != ro[1])[:4].tolist())
        stats["ok"] += 1
    return stats


@pytest.mark.parametrize("cfg", ["A", "B", "C"])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_streams_bit_exact(cfg, seed):
    p = default_params(cfg, BASE_SEED + seed, n_frames=8, qdelta_prob=300, table1_prob=500, pm_intra=120,
                       pm_multiref=200, escape_prob=60, edge_mode=1 if seed == 2 else 0, iframe_interval=5 if seed == 1 else 0)
    assert _compare(p)["ok"] == 8


@pytest.mark.parametrize("q", [12, 25, 40, 52])
def test_quantizer_range(q):
    assert _compare(default_params("A", BASE_SEED + q, n_frames=5, quantizer=q))["ok"] == 5


def test_small_and_odd_geometries():
    for (w, h, ver) in [(16, 16, 2), (32, 16, 1), (256, 16, 1), (512, 32, 2), (528, 48, 2), (1024, 32, 2), (272, 32, 1)]:
        p = default_params("A", BASE_SEED + w + h, n_frames=7, width=w, height=h, version=ver, pm_intra=150, mv_range=12)
        assert _compare(p)["ok"] == 7, (w, h, ver)


def test_corrupted_streams_agree_on_error_class():
    """Bit flips: whatever the reference would do (decode garbage or throw), the parser must do the same
    thing -- same planes when it decodes, same exception class when it throws."""
    rng = np.random.default_rng(7)
    tally = {"ok": 0, "err": 0, "unsupported": 0}
    for trial in range(60):
        p = default_params("A", BASE_SEED + 300 + trial, n_frames=4, pm_intra=100, width=64, height=48)

        def flip(d, fo, rng=rng):
            for _ in range(int(rng.integers(1, 4))):
                i = int(rng.integers(0, d.size))
                d[i] ^= 1 << int(rng.integers(0, 8))
            return d
        s = _compare(p, mutate=flip)
        for k in tally:
            tally[k] += s[k]
    assert tally["ok"] > 20 and tally["err"] > 5, tally
    assert tally["unsupported"] < tally["err"] + tally["ok"], tally


@pytest.mark.parametrize("version,cfg", [(2, "B"), (1, "A")])
def test_heavier_corruption_both_versions(version, cfg):
    """The same differential check with the richer syntax mix (deep trees, several references, quantiser deltas, escapes,
    intra macroblocks in P-frames), both codec versions, byte garbage and many flips per stream."""
    rng = np.random.default_rng(11 + version)
    tally = {"ok": 0, "err": 0, "unsupported": 0}
    for trial in range(50):
        p = default_params(cfg, BASE_SEED + 700 + trial, n_frames=5, width=96, height=64, version=version, pm_intra=120, pm_deep=150,
                           pm_multiref=250, qdelta_prob=250, escape_prob=80, table1_prob=400, iframe_interval=3)

        def mangle(d, fo, rng=rng, trial=trial):
            if trial % 3 == 0:
                a = int(rng.integers(0, max(1, d.size - 16)))
                d[a:a + 16] = rng.integers(0, 256, d[a:a + 16].size, dtype=np.uint8)
            else:
                for _ in range(int(rng.integers(1, 12))):
                    d[int(rng.integers(0, d.size))] ^= 1 << int(rng.integers(0, 8))
            return d
        s = _compare(p, mutate=mangle)
        for k in tally:
            tally[k] += s[k]
    assert tally["ok"] > 20 and tally["err"] > 5, tally


def test_truncated_and_odd_length_data():
    p = default_params("A", BASE_SEED + 1, n_frames=2, width=64, height=48)
    data, fo = generate_clip(p)
    for cut in (1, 2, 3, 7, int(fo[1]) // 2, int(fo[1]) - 1):
        a, o = InterpDecoder(64, 48, 1), OracleDecoder(64, 48, 1)
        a.Data = o.Data = data[:cut]
        a.Offset = o.Offset = 0
        a.DecodeFrame(), o.DecodeFrame()
        if a.last_error in (-5, -6):
            continue
        assert a.last_error == o.last_error and a.Offset == o.Offset, (cut, a.last_error, o.last_error)


def test_p_frame_before_any_i_frame_is_null_reference():
    p = default_params("B", BASE_SEED, n_frames=2, width=64, height=48)
    data, fo = generate_clip(p)
    a, o = InterpDecoder(64, 48, 2), OracleDecoder(64, 48, 2)
    a.Data = o.Data = data[fo[1]:fo[2]]
    a.Offset = o.Offset = 0
    assert a.DecodeFrame() is None and o.DecodeFrame() is None
    assert a.last_error == o.last_error == -2
    assert a.Quantizer == o.Quantizer == 12  # Moflex: Quantizer==0 -> Setup(0) clamps to 12 (MD.cs:123-126, 3886-3890)


def test_clamp_fault_fixture_is_a_clamp_only_fault():
    """tests/golden/clamp_fault_mods_64x48.bin (made by tests/golden/make_clamp_fault.py): the oracle throws its index fault, the
    product's parser accepts the syntax and the kernels' arithmetic (here on the CPU) reports the clamp-table domain."""
    import os
    import numpy as np
    from mobiclipdecoder_amd import MobiclipVersion
    from tests.oracle_binding import OracleDecoder
    from tests.interp_binding import InterpDecoder
    data = np.fromfile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clamp_fault_mods_64x48.bin"), dtype=np.uint8)
    o = OracleDecoder(64, 48, MobiclipVersion.ModsDS)
    o.Data, o.Offset = data, 0
    assert o.DecodeFrame() is None and o.last_error == -1
    i = InterpDecoder(64, 48, MobiclipVersion.ModsDS)
    i.Data, i.Offset = data, 0
    assert i.DecodeFrame() is None and i.last_error == -5
