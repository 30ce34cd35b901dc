"""The two situations in which a device-parsing batch leans on the host parser with EVERY clip:

  1. a ModsDS batch below quantiser 12 -- legitimate content (MD.cs:3907-3911: the dequant words then leak into the zigzag byte, every
     residual block walks through Internal[], MD.cs:3424-3429): no frame is the device parsers' to finish, the whole batch lives with the host
     parser (a wasted device parse every few hundred frames); what is measured is the steady state's rate, against the same batch at
     quantiser 12 on the device parsers;
  2. every clip of a batch glitches in the same frame, under asynchronous steps: mobi_batch_wait repairs them all (r06: one batch operation).

  python tools/exp_allhost.py [clips]        (64x48 ModsDS; default 16384)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import mobiclipdecoder_amd as m
from mobiclipdecoder_amd.streamgen import BASE_SEED
from tests.test_internal_walk import _set_quantizer

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
W, H, NFR, DISTINCT = 64, 48, 14, 8  # (at real picture sizes the reference throws on practically every P-frame below quantiser 12 -- some walk leaves
# Internal[] or wrecks a table the frame still needs; 64x48 at quantiser 10 decodes: profiles/r06_experiments.txt)


def streams(q):
    """DISTINCT streams that the reference decodes at quantiser q (checked with the oracle, the checker: below 12 a walk through Internal[]
    can also run into something the reference throws on -- such a stream is not content, it is skipped)"""
    from tests.oracle_binding import OracleDecoder
    out, seed = [], 0
    while len(out) < DISTINCT and seed < 400:
        p = m.default_params("A", BASE_SEED + 600 + seed, n_frames=NFR, width=W, height=H, quantizer=12, pm_intra=100, cbp_prob=400)
        seed += 1
        data, fo = m.generate_clip(p)
        data = data.copy()
        if q != 12:
            _set_quantizer(data[fo[0]:fo[1]], q)  # the I-frame names the quantiser; the P-frames keep it (delta 0)
        o = OracleDecoder(W, H, 1)
        ok = True
        for f in range(NFR):
            o.Data, o.Offset = data[fo[f]:fo[f + 1]], 0
            ok = ok and o.DecodeFrame() is not None
        o.close()
        if ok:
            out.append((data, fo))
    assert len(out) == DISTINCT, f"only {len(out)} streams decode at quantiser {q}"
    return out


def run(q, label):
    src = streams(q)
    b = m.MobiclipBatch(clips, W, H, 1, device_parse=True)
    ms = []
    for f in range(NFR):
        datas = [src[c % DISTINCT][0][src[c % DISTINCT][1][f]:src[c % DISTINCT][1][f + 1]] for c in range(clips)]
        rcs, _ = b.decode(datas, [0] * clips)
        assert not any(rcs), (q, f, [r for r in rcs if r][:4])
        ms.append(b.last_decode_ms())
    t = float(np.median(ms[3:]))
    print(f"{label}: {clips} clips {W}x{H} ModsDS, quantiser {q}: {t:.2f} ms per step = {clips * W * H / t / 1e6:.1f} Gpixels/s; clips with the host parser at the end: {b.host_clips()}")
    b.close()
    return t


run(12, "device parsers (quantiser 12)")
run(10, "all-host steady state (quantiser 10: every block walks through Internal[])")

# every clip glitches in frame 2 (its I-frame header re-written to quantiser 5), two asynchronous steps in flight
src = []
for i in range(3):
    p = m.default_params("A", BASE_SEED + 3005 + 100 * i, n_frames=6, width=64, height=48, quantizer=12, pm_intra=150, cbp_prob=500, iframe_interval=2)
    data, fo = m.generate_clip(p)
    data = data.copy()
    _set_quantizer(data[fo[2]:fo[3]], 5)
    src.append((data, fo))
for n in (2048, 8192):
    b = m.MobiclipBatch(n, 64, 48, 1, device_parse=True)
    frames = [[src[i % 3][0][src[i % 3][1][f]:src[i % 3][1][f + 1]] for i in range(n)] for f in range(6)]
    waits = []
    b.submit(frames[0], [0] * n)
    for f in range(1, 6):
        b.submit(frames[f], [0] * n)
        t0 = time.perf_counter()
        rcs, _ = b.wait()
        waits.append((time.perf_counter() - t0) * 1e3)
        assert not any(rcs)
    rcs, _ = b.wait()
    assert not any(rcs) and b.host_clips() == n
    print(f"every clip of {n} (64x48 ModsDS) handed over in the same frame, two asynchronous steps in flight: the mobi_batch_wait that repairs two frames of each: "
          f"{waits[2]:.1f} ms (the other waits: {[round(w, 1) for w in waits[:2] + waits[3:]]} ms)")
    b.close()
