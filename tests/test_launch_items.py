"""The launch items of intra macroblocks (ParsedFrame::intra_items, mobi_parse.cpp finish_levels) against the descriptors they summarise.
The step's launch list (LevelPlan, mobi_abi.cpp) is a concatenation of these, so what a wave of mobi_recon_intra is told -- where the
records are, whether to poll tags, whether to publish its own -- is decided here, in the parser."""
import numpy as np
import pytest

from mobiclipdecoder_amd import default_params, generate_clip
from mobiclipdecoder_amd.streamgen import BASE_SEED
from tests.interp_binding import InterpDecoder

DEP_NONE, DEP_INTER = 0xFFFF, 0x8000


@pytest.mark.parametrize("cfg,kw", [("A", dict(pm_intra=300)), ("A", dict(width=64, height=48, pm_intra=500, plane_prob=500)),
                                    ("B", dict(pm_intra=120, n_frames=4)), ("A", dict(width=1024, height=32, version=2, pm_intra=400))])
def test_items_say_what_the_descriptors_say(cfg, kw):
    p = default_params(cfg, BASE_SEED + 6100, **{"n_frames": 5, **kw})
    data, fo = generate_clip(p)
    d = InterpDecoder(p.width, p.height, p.version)
    mbw = p.width // 16
    seen_deps = seen_dependents = seen_edge = 0
    seen_classes = set()
    for f in range(p.n_frames):
        d.Data, d.Offset = data[: fo[f + 1]], int(fo[f])
        assert d.DecodeFrame() is not None
        desc, mbs, ls, items = d.command_list()
        intra = np.nonzero((desc[:, 1] & 1) == 1)[0]  # MOBI_MB_INTRA
        assert sorted(mbs.tolist()) == intra.tolist() and len(items) == len(mbs)
        assert ls[0] == 0 and ls[1] == 0 and ls[-1] == len(mbs) and np.all(np.diff(ls.astype(np.int64)) >= 0)
        named = set()  # intra macroblocks some other intra macroblock's halo reads
        for mb in mbs:
            w = desc[mb, 4:8]
            for dep in np.concatenate([w & 0xFFFF, w >> 16]):
                if dep != DEP_NONE and not dep & DEP_INTER:
                    assert (desc[dep & 0x1FFF, 1] & 1) == 1 and (dep & 0x1FFF) < mb
                    named.add(int(dep & 0x1FFF))
        # launch order inside a level: away from the picture's edges first, those nobody depends on first, fewest split areas first (a wave of four runs the longest one's steps)
        pay = d.payload()
        def klass(mb):
            splits = sum((int(pay[int(desc[mb, 0]) + 4 * a]) >> 5) & 1 for a in range(6))
            mbx = int(mb) % mbw
            return (0 if (mbx >= 1 and mbx + 1 < mbw and mb >= mbw) else 8) + (4 if int(mb) in named else 0) + min(splits, 3)
        for L in range(1, len(ls) - 1):
            keys = [(klass(mb), int(mb)) for mb in mbs[ls[L]:ls[L + 1]]]
            assert keys == sorted(keys), (f, L)
            seen_classes.update(k for k, _ in keys)
        level = {}
        for L in range(1, len(ls) - 1):
            for i in range(ls[L], ls[L + 1]):
                level[int(mbs[i])] = L
        for i, mb in enumerate(mbs):
            it = items[i]
            assert it[0] == mb and it[1] == desc[mb, 1] and it[2] == desc[mb, 0]
            w = desc[mb, 4:8]
            deps = [int(x) for x in np.concatenate([w & 0xFFFF, w >> 16]) if x != DEP_NONE]
            intra_deps = [x & 0x1FFF for x in deps if not x & DEP_INTER]
            flags = int(it[3])
            assert (flags & 0xFFFF0001) == (int(desc[mb, 3]) & 0xFFFF0001)
            assert bool(flags & 2) == bool(intra_deps) and bool(flags & 4) == (int(mb) in named)
            mbx = int(mb) % mbw
            assert bool(flags & 8) == (not (mbx >= 1 and mbx + 1 < mbw and mb >= mbw))
            assert (flags >> 5) & 0x3FF == int(desc[mb, 2]) & 0x3FF and not flags & 0x8010
            # a macroblock's level is one more than the highest level among the intra macroblocks it waits for
            assert level[int(mb)] == 1 + max([level[x] for x in intra_deps], default=0)
            seen_deps += bool(flags & 2); seen_dependents += bool(flags & 4); seen_edge += bool(flags & 8)
    assert seen_deps and seen_dependents and seen_edge and len(seen_classes) >= 3


def _intra_deps_by_frame(p):
    data, fo = generate_clip(p)
    d = InterpDecoder(p.width, p.height, p.version)
    for f in range(p.n_frames):
        d.Data, d.Offset = data[: fo[f + 1]], int(fo[f])
        assert d.DecodeFrame() is not None
        desc, mbs, _, _ = d.command_list()
        for mb in mbs:
            w = desc[mb, 4:8]
            for dep in np.concatenate([w & 0xFFFF, w >> 16]):
                if dep != DEP_NONE:
                    yield int(mb), int(dep & 0x1FFF)


@pytest.mark.parametrize("w,h,ver", [(96, 64, 1), (176, 144, 2), (336, 48, 1), (640, 480, 2), (848, 480, 2)])
def test_every_halo_lies_on_an_earlier_wavefront_when_the_picture_is_narrower_than_its_stride(w, h, ver):
    """what mobi_launch_gop_sort (mobi_gop.hip) relies on: ordered by mbx + 2 * mby, every macroblock a halo reads -- inter or intra -- comes
    before the one that reads it (the lock-step parser's descriptors are these, tests/test_lsparse.py)"""
    p = default_params("A", BASE_SEED + 6300, n_frames=3 if w < 600 else 2, width=w, height=h, version=ver, pm_intra=800, intra_sub_prob=600, iframe_interval=2)
    mbw, n = w // 16, 0
    for mb, dep in _intra_deps_by_frame(p):
        assert dep % mbw + 2 * (dep // mbw) < mb % mbw + 2 * (mb // mbw), (mb, dep)
        n += 1
    assert n > 100


def test_a_picture_as_wide_as_its_stride_wraps_to_a_later_wavefront():
    """... and why 256-, 512-, 1024-wide pictures keep the raster-order launch: the first macroblock of a row reads the LAST one of the row
    above (the reference's linear offsets with Stride == Width, MD.cs:212-217)"""
    p = default_params("A", BASE_SEED + 6301, n_frames=2, width=256, height=64, pm_intra=900, iframe_interval=1)
    mbw = p.width // 16
    later = [(mb, dep) for mb, dep in _intra_deps_by_frame(p) if dep % mbw + 2 * (dep // mbw) >= mb % mbw + 2 * (mb // mbw)]
    assert later and all(mb % mbw == 0 and dep == mb - 1 for mb, dep in later)
