"""Intra macroblocks per clip and P-frame in the streams the group experiments use (tools/exp_gop.py: 64 distinct, their own seeds) and in
the bench's replay (16 distinct): the intra launch's time per step follows it.   python tools/exp_intra_density.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import mobiclipdecoder_amd as m
from mobiclipdecoder_amd import sharding
from mobiclipdecoder_amd.streamgen import BASE_SEED

for label, seeds, nfr in (("bench replay (16 streams)", [sharding.stream_seed("B", 0, i) for i in range(16)], 33),
                          ("tools/exp_gop.py (64 streams)", [BASE_SEED + 100 + i for i in range(64)], 46)):
    streams = []
    for s in seeds:
        p = m.default_params("B", s, n_frames=nfr)
        streams.append((p,) + m.generate_clip(p))
    p0 = streams[0][0]
    b = m.MobiclipBatch(len(streams), p0.width, p0.height, p0.version)
    for i, (p, data, fo) in enumerate(streams):
        assert all(r == 0 for r in b.preload(i, data, fo))
    b.commit()
    per = [b.intra_stats(f)[0] / len(streams) for f in range(1, nfr)]
    print(f"{label}: intra macroblocks per clip and P-frame: mean {np.mean(per):.1f}, min {min(per):.1f}, max {max(per):.1f} (of {p0.width // 16 * (p0.height // 16)})")
    b.close()
