"""Encoder-side analysis: Analyzer.InterPredict2x2 over every 2x2 luma block (Analyzer.cs:608-693).

CPU: the oracle's restatement against an independent numpy brute-force of the same three-step search, and on inputs whose
answer is known.  GPU (-m gpu): mobi_batch_motion_search against the oracle, bit-exact packed words."""
import numpy as np
import pytest

from mobiclipdecoder_amd import default_params, generate_clip, unpack_motion_search
from mobiclipdecoder_amd.streamgen import BASE_SEED
from tests.oracle_binding import OracleDecoder


def _decoded_oracle(cfg="A", n_frames=4, seed=900, **kw):
    p = default_params(cfg, BASE_SEED + seed, n_frames=n_frames, **kw)
    data, fo = generate_clip(p)
    ora = OracleDecoder(p.width, p.height, p.version)
    for f in range(n_frames):
        ora.Data, ora.Offset = data[fo[f]:fo[f + 1]], 0
        assert ora.DecodeFrame() is not None
    return p, data, fo, ora


def _numpy_search(past, pic):
    """Independent statement of Analyzer.cs:608-681 for one picture: plain loops over blocks, numpy only for the planes."""
    H, W = pic.shape
    out = np.zeros((H // 16, W // 16, 8, 8), np.uint32)
    for by in range(0, H, 2):
        for bx in range(0, W, 2):
            cmp = pic[by:by + 2, bx:bx + 2].astype(np.int32)
            res = (0, 0, 0, None)
            for i, ref in enumerate(past):
                if ref is None:
                    break
                cx = cy = 0
                cscore = 0
                for st in (6, 3, 1):
                    best = (None, 0, 0)
                    for y in (-st, 0, st):
                        if by + y + cy < 0 or by + 2 + y + cy > H:
                            continue
                        for x in (-st, 0, st):
                            if bx + x + cx < 0 or bx + 2 + x + cx > W:
                                continue
                            blk = ref[by + y + cy:by + y + cy + 2, bx + x + cx:bx + x + cx + 2].astype(np.int32)
                            s = int(np.abs(cmp - blk).sum())
                            nx, ny = x + cx, y + cy
                            if best[0] is None or s < best[0] or (s == best[0] and abs(nx) + abs(ny) < abs(best[1]) + abs(best[2])):
                                best = (s, nx, ny)
                    cscore, cx, cy = best
                if res[3] is None or cscore < res[3] or (cscore == res[3] and abs(2 * cx) + abs(2 * cy) < abs(res[0]) + abs(res[1])):
                    res = (2 * cx, 2 * cy, i, cscore)
            sc = 0xFFF if res[3] is None else res[3]
            out[by // 16, bx // 16, (by % 16) // 2, (bx % 16) // 2] = (res[0] & 0xFF) | ((res[1] & 0xFF) << 8) | (res[2] << 16) | (sc << 20)
    return out


def test_oracle_search_matches_an_independent_numpy_statement():
    p, _, _, ora = _decoded_oracle("A", n_frames=3)
    rng = np.random.default_rng(5)
    S = ora.Stride
    past = [ora.y(i) for i in range(5)]
    past = [None if a is None else a[:, :p.width] for a in past]
    # a picture that resembles the last frame (shifted, with noise) so that scores and ties are interesting
    pic = np.roll(past[0], (3, -6), axis=(0, 1)).astype(np.int32) + rng.integers(-3, 4, past[0].shape)
    pic = np.clip(pic, 0, 255).astype(np.uint8)
    pic[:64, :64] = past[1][:64, :64]  # one corner is an exact copy of the frame before
    got = ora.motion_search(pic)
    want = _numpy_search(past, pic)
    assert np.array_equal(got, want), np.argwhere(got != want)[:5].tolist()
    assert S >= p.width
    ora.close()


def test_known_answers():
    p, _, _, ora = _decoded_oracle("A", n_frames=2)
    y0 = ora.y(0)[:, :p.width]
    # the frame itself: zero vector, frame 0, score 0 everywhere (ties go to the shorter vector and the earlier frame)
    r = unpack_motion_search(ora.motion_search(y0))
    assert not r["dx"].any() and not r["dy"].any() and not r["frame"].any() and not r["score"].any()
    # no past frame at all: the reference's loop body never runs
    fresh = OracleDecoder(p.width, p.height, p.version)
    r = unpack_motion_search(fresh.motion_search(y0))
    assert (r["score"] == 0xFFF).all() and not r["dx"].any() and not r["frame"].any()
    fresh.close()
    # vectors are even (full pels stored as half pels) and within the reach of 6 + 3 + 1 pels
    pic = np.roll(y0, (-9, 7), axis=(0, 1))
    r = unpack_motion_search(ora.motion_search(pic))
    assert (r["dx"] % 2 == 0).all() and (np.abs(r["dx"]) <= 20).all() and (np.abs(r["dy"]) <= 20).all()
    ora.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["A", "B"])
def test_gpu_search_equals_oracle(cfg):
    from mobiclipdecoder_amd import MobiclipBatch
    n, nfr = 3, 7
    ps = [default_params(cfg, BASE_SEED + 910 + i, n_frames=nfr, pm_intra=80) for i in range(n)]
    clips = [generate_clip(p) for p in ps]
    W, H = ps[0].width, ps[0].height
    b = MobiclipBatch(n, W, H, ps[0].version)
    oras = [OracleDecoder(W, H, ps[0].version) for _ in range(n)]
    rng = np.random.default_rng(77)
    for f in range(nfr):
        if f in (0, 2, 6):  # empty ring, partly filled ring (two frames), full ring of five and more
            pics = []
            for i in range(n):
                prev = oras[i].y(0)
                base = rng.integers(0, 256, (H, W), dtype=np.uint8) if prev is None else np.roll(prev[:, :W], (int(rng.integers(-8, 9)), int(rng.integers(-8, 9))), axis=(0, 1))
                pic = np.clip(base.astype(np.int32) + rng.integers(-2, 3, (H, W)), 0, 255).astype(np.uint8)
                if i == 1 and f == 6:
                    pic[: H // 2] = oras[i].y(3)[: H // 2, :W]  # half of the picture comes straight from an older frame
                pics.append(pic)
            got = b.motion_search(pics)["packed"]
            for i in range(n):
                want = oras[i].motion_search(pics[i])
                assert np.array_equal(got[i], want), (f, i, np.argwhere(got[i] != want)[:5].tolist())
            if f == 6:
                assert len(np.unique(unpack_motion_search(got[1])["frame"])) > 1
        rcs, _ = b.decode([c[0][c[1][f]:c[1][f + 1]] for c in clips], [0] * n)
        assert rcs == [0] * n
        for i in range(n):
            oras[i].Data, oras[i].Offset = clips[i][0][clips[i][1][f]:clips[i][1][f + 1]], 0
            assert oras[i].DecodeFrame() is not None
    b.close()
