"""YUV -> ARGB (the Bitmap DecodeFrame() returns, MD.cs:260-323): the oracle's C restatement against an independent
numpy float32 restatement (CPU), and the HIP kernel against the oracle, bit-exact (GPU)."""
import numpy as np
import pytest

from mobiclipdecoder_amd import default_params, generate_clip
from mobiclipdecoder_amd.streamgen import BASE_SEED
from tests.oracle_binding import OracleDecoder


def _numpy_argb(Y, UV, W, H, version):
    """Same statement order as the reference, every operation in np.float32 (one rounding per operator)."""
    f = np.float32
    S = Y.shape[1]
    Yf = Y[:H, :W].astype(f)
    ys, xs = np.mgrid[0:H, 0:W]
    cy, cx = ys // 2, xs // 2
    UVp = np.pad(UV, ((0, 1), (0, 1)))  # neighbours are only used where the reference reads them; padding keeps indices legal

    def plane(off):
        s = lambda dy, dx: UVp[cy + dy, cx + dx + off].astype(f) - f(128)
        base = s(0, 0)
        interp = (xs != W - 1) & (ys != H - 1)
        case = (xs & 1) | ((ys & 1) << 1)
        c1 = (base + s(0, 1)) / f(2)
        c2 = (base + s(1, 0)) / f(2)
        c3 = (((base + s(0, 1)) + s(1, 0)) + s(1, 1)) / f(4)
        out = base.copy()
        for k, c in ((1, c1), (2, c2), (3, c3)):
            m = interp & (case == k)
            out[m] = c[m]
        return out

    U, V = plane(0), plane(S // 2)
    if version == 2:
        R = Yf + f(1.420) * V
        G = (Yf - f(0.344) * U) - f(0.714) * V
        B = Yf + f(1.772) * U
        R, G, B = [((c - f(16)) * f(255)) / f(239) for c in (R, G, B)]
    else:
        yi, ui, vi = Yf.astype(np.int32), np.trunc(U).astype(np.int32), np.trunc(V).astype(np.int32)
        R, G, B = (yi + ui - vi).astype(f), (yi + vi).astype(f), (yi - ui - vi).astype(f)
    R, G, B = [np.clip(c, f(0), f(255)).astype(np.int32).astype(np.uint32) for c in (R, G, B)]
    return np.uint32(0xFF000000) | (R << 16) | (G << 8) | B


@pytest.mark.parametrize("cfg", ["A", "B"])
def test_oracle_argb_matches_float32_restatement(cfg):
    p = default_params(cfg, BASE_SEED + 41, n_frames=3, width=64 if cfg == "A" else 96, height=48)
    data, fo = generate_clip(p)
    o = OracleDecoder(p.width, p.height, p.version)
    assert o.argb() is None  # no frame yet: Y[0] is null in the reference
    for f in range(p.n_frames):
        o.Data, o.Offset = data[fo[f]:fo[f + 1]], 0
        assert o.DecodeFrame() is not None
        got = o.argb()
        want = _numpy_argb(o.y(0), o.uv(0), p.width, p.height, p.version)
        assert np.array_equal(got, want), np.argwhere(got != want)[:5]
    assert (got >> 24 == 0xFF).all()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["A", "B", "C"])
def test_hip_bitmap_is_bit_exact(cfg):
    from mobiclipdecoder_amd import MobiclipDecoder
    p = default_params(cfg, BASE_SEED + 42, n_frames=4)
    data, fo = generate_clip(p)
    g = MobiclipDecoder(p.width, p.height, p.version)
    o = OracleDecoder(p.width, p.height, p.version)
    assert g.Bitmap() is None
    for f in range(p.n_frames):
        g.Data = o.Data = data[fo[f]:fo[f + 1]]
        g.Offset = o.Offset = 0
        assert g.DecodeFrame() is not None and o.DecodeFrame() is not None
        a, b = g.Bitmap(), o.argb()
        assert np.array_equal(a, b), (f, np.argwhere(a != b)[:5].tolist())
    g.close()


@pytest.mark.gpu
def test_hip_bitmap_batch_paths_agree():
    from mobiclipdecoder_amd import MobiclipBatch
    n = 3
    ps = [default_params("A", BASE_SEED + 50 + i, n_frames=3) for i in range(n)]
    clips = [generate_clip(p) for p in ps]
    b = MobiclipBatch(n, ps[0].width, ps[0].height, ps[0].version)
    oras = [OracleDecoder(p.width, p.height, p.version) for p in ps]
    for f in range(3):
        b.decode([c[0] for c in clips], [int(c[1][f]) for c in clips])
        for i in range(n):
            oras[i].Data, oras[i].Offset = clips[i][0], int(clips[i][1][f])
            oras[i].DecodeFrame()
        one = [b.bitmap(i) for i in range(n)]       # converts clip by clip
        b.convert_argb()                            # converts the whole batch on the device
        for i in range(n):
            assert np.array_equal(one[i], oras[i].argb()) and np.array_equal(b.bitmap(i), oras[i].argb())
    b.close()


@pytest.mark.gpu
def test_division_by_239_is_correctly_rounded_for_every_float():
    """The Bitmap kernel divides by 239 with q0 = x*r, q = fma(fma(-239, q0, x), r, q0).  That is not correctly
    rounded for arbitrary divisors, so it is checked for this one over all 2^32 bit patterns on the device."""
    from mobiclipdecoder_amd import decoder
    assert decoder.load_library().mobi_selftest_div239(decoder.default_device()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,version", [(16, 16, 2), (48, 32, 1), (848, 48, 2), (1024, 32, 2), (256, 64, 1), (80, 16, 2)])
def test_hip_bitmap_on_random_planes(w, h, version, profiling_library):
    """The Bitmap kernel on planes of random bytes (every Y / U / V value beside every other, the extremes included -- decoded pictures
    are smooth): one wave takes two macroblocks side by side, so widths with an odd number of macroblocks (16, 48, 848, 80), a single
    macroblock, Width == Stride (1024) and both colour paths are here; the last row and the last column take no chroma mean (MD.cs:269).
    Checked against the oracle's ARGB (itself pinned to the float32 restatement above)."""
    import ctypes as C
    from mobiclipdecoder_amd import MobiclipDecoder
    from tests.test_unit_vectors import _inject
    p = default_params("A", BASE_SEED + 43, n_frames=1, width=w, height=h, version=version)
    data, fo = generate_clip(p)
    g = MobiclipDecoder(w, h, version)
    o = OracleDecoder(w, h, version)
    g.Data = o.Data = data[fo[0]:fo[1]]
    g.Offset = o.Offset = 0
    assert g.DecodeFrame() is not None and o.DecodeFrame() is not None
    rng = np.random.default_rng(w * 131 + h)
    for trial in range(4):
        y = np.zeros((h, o.Stride), np.uint8)
        uv = np.zeros((h // 2, o.Stride), np.uint8)
        if trial == 3:  # extremes only
            y[:, :w] = rng.choice(np.array([0, 1, 15, 16, 17, 254, 255], np.uint8), (h, w))
            uv[:, : w // 2] = rng.choice(np.array([0, 1, 127, 128, 129, 255], np.uint8), (h // 2, w // 2))
            uv[:, o.Stride // 2: o.Stride // 2 + w // 2] = rng.choice(np.array([0, 1, 127, 128, 129, 255], np.uint8), (h // 2, w // 2))
        else:
            y[:, :w] = rng.integers(0, 256, (h, w), dtype=np.uint8)
            uv[:, : w // 2] = rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8)
            uv[:, o.Stride // 2: o.Stride // 2 + w // 2] = rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8)
        _inject(profiling_library, g, o, y, uv)
        a, b = g.Bitmap(), o.argb()
        assert np.array_equal(a, b), (trial, np.argwhere(a != b)[:8].tolist())
        assert np.array_equal(b, _numpy_argb(y, uv, w, h, version))
    g.close()
    o.close()
