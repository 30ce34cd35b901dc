#!/usr/bin/env python3
"""End-to-end DecodeFrame throughput (bitstream in, planes in HBM): host parse vs device parse at several batch sizes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd import sharding

def run(n_clips, device_parse, n_frames=9, distinct=16):
    streams = []
    for i in range(distinct):
        p = m.default_params("B", sharding.stream_seed("B", 0, i), n_frames=n_frames)
        streams.append(m.generate_clip(p))
    b = m.MobiclipBatch(n_clips, 640, 480, 2, device_parse=device_parse)
    t_frames = []
    for f in range(n_frames):
        datas = [streams[c % distinct][0][streams[c % distinct][1][f]:streams[c % distinct][1][f + 1]] for c in range(n_clips)]
        offs = [0] * n_clips
        t0 = time.perf_counter()
        rcs, _ = b.decode(datas, offs)
        t_frames.append(time.perf_counter() - t0)
        assert all(r == 0 for r in rcs), rcs[:8]
    b.close()
    p_ms = np.array(t_frames[2:]) * 1e3   # skip the I-frame and the first P-frame (allocations)
    px = n_clips * 640 * 480
    print(f"clips={n_clips:5d} device_parse={int(device_parse)}  I-frame {t_frames[0]*1e3:8.2f} ms   P-frame median {np.median(p_ms):8.2f} ms  "
          f"min {p_ms.min():8.2f} ms  -> {px / np.median(p_ms) / 1e3:9.1f} Mpix/s end to end", flush=True)

if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [512, 2048]
    for n in sizes:
        for dp in (False, True):
            run(n, dp)
