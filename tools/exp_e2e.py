"""End-to-end DecodeFrame() throughput: host VLC parse (thread pool) + upload + kernels, through mobi_batch_decode."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd.streamgen import BASE_SEED
def run(clips, threads, nfr=17):
    os.environ["MOBI_PARSE_THREADS"] = str(threads)
    distinct = min(clips, 16)
    gen = [m.generate_clip(m.default_params("B", BASE_SEED + i, n_frames=nfr)) for i in range(distinct)]
    b = m.MobiclipBatch(clips, 640, 480, 2)
    datas = [gen[c % distinct][0] for c in range(clips)]
    b.decode(datas, [int(gen[c % distinct][1][0]) for c in range(clips)])  # I-frame, warm-up
    t0 = time.perf_counter()
    for f in range(1, nfr):
        rcs, _ = b.decode(datas, [int(gen[c % distinct][1][f]) for c in range(clips)])
        assert all(r == 0 for r in rcs)
    dt = time.perf_counter() - t0
    px = clips * (nfr - 1) * 640 * 480
    print(f'{clips} clips, {threads} parse thread(s): {dt / (nfr - 1) * 1e3:.1f} ms per frame step -> {px / dt / 1e9:.2f} Gpix/s end to end ({clips * (nfr - 1) / dt:.0f} frames/s)', flush=True)
    b.close()
for t in (1, 8, 32): run(256, t)
