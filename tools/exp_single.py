#!/usr/bin/env python3
"""One clip through the single-stream API (mobi_create / mobi_decode): time per frame inside the C call, by configuration."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd import sharding
for cfg, w, h, ver in (("A", 256, 192, 1), ("B", 640, 480, 2)):
    data, fo = m.generate_clip(m.default_params(cfg, sharding.stream_seed(cfg, 0, 0), n_frames=33))
    b = m.MobiclipBatch(1, w, h, ver)
    ms, wall = [], []
    for f in range(33):
        t0 = time.perf_counter()
        rcs, _ = b.decode([data[fo[f]:fo[f + 1]]], [0])
        wall.append(time.perf_counter() - t0)
        assert rcs == [0]
        ms.append(b.last_decode_ms())
    print(f"{w}x{h}: P-frame median {np.median(ms[1:]):.3f} ms inside the C call ({1e3 / np.median(ms[1:]):.0f} frames/s, {w * h / np.median(ms[1:]) / 1e3:.1f} Mpix/s); "
          f"I-frame {ms[0]:.3f} ms; python round trip {np.median(wall[1:]) * 1e3:.3f} ms")
    b.close()
