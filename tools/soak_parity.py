"""Soak: N clips (private copies of 16 generated streams) replayed for a whole 33-frame clip, twice, on a loaded chip; EVERY clip's
planes are compared with the oracle's at the I-frame, in the middle and at the end.  Looks for rare hand-off races (completion tags,
write-through stores) that a handful of clips would never show.  python tools/soak_parity.py [clips] [config] [dparse | lockstep | gop]
"dparse": the same through mobi_batch_decode with the parse on the GPU (mobi_recon_intra_cl: items per clip in raster order), 12 frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd import sharding
from tests.oracle_binding import OracleDecoder

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
cfg = sys.argv[2] if len(sys.argv) > 2 else "B"
dparse = len(sys.argv) > 3 and sys.argv[3] in ("dparse", "lockstep")
lockstep = dparse and sys.argv[3] == "lockstep"  # the lock-step parser in front (32 clips per wave)
distinct, nfr = 16, (12 if dparse else 33)
ps = [m.default_params(cfg, sharding.stream_seed(cfg, 0, i), n_frames=nfr, pm_intra=120 if i & 1 else 50, iframe_interval=11 if i % 5 == 0 else 0) for i in range(distinct)]
clips = [m.generate_clip(p) for p in ps]
W, H, ver = ps[0].width, ps[0].height, ps[0].version
b = m.MobiclipBatch(n, W, H, ver, device_parse=("lockstep" if lockstep else True) if dparse else None)
if not dparse:
    for i in range(distinct):
        assert all(r == 0 for r in b.preload(i, clips[i][0], clips[i][1]))
    for c in range(distinct, n):
        b.preload_clone(c, c % distinct)
    b.commit()
oras = [OracleDecoder(W, H, ver) for _ in range(distinct)]
bad = 0
t0 = time.time()
for f in range(nfr):
    if dparse:
        rcs, _ = b.decode([clips[c % distinct][0][clips[c % distinct][1][f]:clips[c % distinct][1][f + 1]] for c in range(n)], [0] * n)
        assert not any(rcs)
    else:
        b.replay(f)
    for i in range(distinct):
        oras[i].Data, oras[i].Offset = clips[i][0], int(clips[i][1][f])
        assert oras[i].DecodeFrame() is not None
    if f in (0, 11, 16, 22, nfr - 1) or (dparse and f % 3 == 0):
        assert dparse or b.sync() == 0
        for c in range(n):
            y, uv = b.planes(c)
            o = oras[(c + (1 if os.environ.get('SOAK_SELFTEST') and c == 7 else 0)) % distinct]  # SOAK_SELFTEST=1: clip 7 against the wrong stream must be reported
            if not (np.array_equal(y, o.y(0)) and np.array_equal(uv, o.uv(0))):
                bad += 1
                if bad < 5:
                    print("MISMATCH frame", f, "clip", c, "stream", c % distinct, np.argwhere(y != o.y(0))[:3].tolist())
        print("frame %2d: %d clips checked, %d mismatches so far, %.0f s" % (f, n, bad, time.time() - t0), flush=True)
b.close()
print("soak:", "OK" if bad == 0 else "FAILED", n, "clips of", cfg)
if len(sys.argv) > 3 and sys.argv[3] == "gop":
    # r06: the same streams in frame-parallel groups (mobi_batch_gop_begin / gop_finish, two groups begun), 31 frames: the I-frame on its own, then
    # five groups of six; EVERY clip's newest frame against the oracle after every group (the first eight clips: all six frames of the group)
    nfr, K = 31, 6
    ps = [m.default_params(cfg, sharding.stream_seed(cfg, 0, i), n_frames=nfr, pm_intra=120 if i & 1 else 50, iframe_interval=11 if i % 5 == 0 else 0) for i in range(distinct)]
    clips = [m.generate_clip(p) for p in ps]
    b = m.MobiclipBatch(n, W, H, ver, device_parse=None)
    oras = [OracleDecoder(W, H, ver) for _ in range(distinct)]
    frame = lambda f: [clips[c % distinct][0][clips[c % distinct][1][f]:clips[c % distinct][1][f + 1]] for c in range(n)]

    def oracle_step(f):
        for i in range(distinct):
            oras[i].Data, oras[i].Offset = clips[i][0], int(clips[i][1][f])
            assert oras[i].DecodeFrame() is not None

    rcs, _ = b.decode_gop([frame(0)])
    assert not any(rcs[0])
    oracle_step(0)
    bad, t0 = 0, time.time()
    groups = [list(range(1 + K * g, 1 + K * (g + 1))) for g in range((nfr - 1) // K)]
    b.gop_begin([frame(f) for f in groups[0]])
    for g in range(len(groups)):
        if g + 1 < len(groups):
            b.gop_begin([frame(f) for f in groups[g + 1]])
        rcs, _ = b.gop_finish()
        assert not any(r for row in rcs for r in row)
        for f in groups[g]:
            oracle_step(f)
        for c in range(n):
            for k in (range(K) if c < 8 else [K - 1]):
                y, uv = b.planes(c, K - 1 - k)
                o = oras[c % distinct]
                if not (np.array_equal(y, o.y(K - 1 - k)) and np.array_equal(uv, o.uv(K - 1 - k))):
                    bad += 1
                    if bad < 5:
                        print("MISMATCH group", g, "frame", groups[g][k], "clip", c)
        print("group %d (frames %d..%d): %d clips checked, %d mismatches so far, host clips %d, %.0f s" % (g, groups[g][0], groups[g][-1], n, bad, b.host_clips(), time.time() - t0), flush=True)
    b.close()
    print("soak (frame-parallel groups):", "OK" if bad == 0 else "FAILED", n, "clips of", cfg)
sys.exit(0 if bad == 0 else 1)
