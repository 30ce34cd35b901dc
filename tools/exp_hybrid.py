"""Where should the parse run at a mid-size batch?  mobi_batch_decode and submit / wait, 640x480 P-frames, bitstreams in host memory -> planes in HBM:
parse mode 1 (GPU, one wave per clip) against mode 2 (hybrid) with several host shares.   python tools/exp_hybrid.py [clips]"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd.streamgen import BASE_SEED
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nfr, distinct = 14, 16
gen = [m.generate_clip(m.default_params("B", BASE_SEED + i, n_frames=nfr)) for i in range(distinct)]
frames = [[gen[c % distinct][0][gen[c % distinct][1][f]:gen[c % distinct][1][f + 1]] for c in range(clips)] for f in range(nfr)]
def run(mode, share):
    if share is not None: os.environ["MOBI_HYBRID_HOST_CLIPS"] = str(share)
    else: os.environ.pop("MOBI_HYBRID_HOST_CLIPS", None)
    b = m.MobiclipBatch(clips, 640, 480, 2, device_parse=mode)
    ms = []
    for f in range(nfr):
        rcs, _ = b.decode(frames[f], [0] * clips)
        assert not any(rcs)
        if f >= 3: ms.append(b.last_decode_ms())
    hc = b.host_clips(); b.close()
    b = m.MobiclipBatch(clips, 640, 480, 2, device_parse=mode)
    for f in range(3):
        b.submit(frames[f], [0] * clips); b.wait()
    t0 = time.perf_counter()
    b.submit(frames[3], [0] * clips)
    for f in range(4, nfr):
        b.submit(frames[f], [0] * clips); rcs, _ = b.wait(); assert not any(rcs)
    b.wait()
    ta = (time.perf_counter() - t0) * 1e3 / (nfr - 3)
    b.close()
    t = float(np.median(ms))
    print(f"{clips} clips, parse mode {mode}, host share {hc}: decode {t:.2f} ms = {clips * 307200 / t / 1e6:.1f} Gpixels/s; submit/wait {ta:.2f} ms = {clips * 307200 / ta / 1e6:.1f} Gpixels/s (Python packing included)", flush=True)
run(1, None)
for share in ([int(x) for x in sys.argv[2:]] or [clips // 8, clips // 6, clips * 2 // 9, clips // 4, clips // 3]):
    run(2, share)
