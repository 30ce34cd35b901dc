#!/usr/bin/env python3
"""Throughput of mobi_batch_motion_search (Analyzer.InterPredict2x2 over all 2x2 blocks) vs the oracle on one host thread."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd import sharding
from tests.oracle_binding import OracleDecoder

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
p = m.default_params("B", sharding.stream_seed("B", 0, 0), n_frames=7)
data, fo = m.generate_clip(p)
b = m.MobiclipBatch(n, 640, 480, 2)
ora = OracleDecoder(640, 480, 2)
for f in range(7):
    pkt = data[fo[f]:fo[f + 1]]
    rcs, _ = b.decode([pkt] * n, [0] * n)
    assert all(r == 0 for r in rcs)
    ora.Data, ora.Offset = pkt, 0
    ora.DecodeFrame()
rng = np.random.default_rng(1)
pic = np.clip(np.roll(ora.y(0)[:, :640], (4, -5), axis=(0, 1)).astype(np.int32) + rng.integers(-2, 3, (480, 640)), 0, 255).astype(np.uint8)
pics = [pic] * n
ts = []
for _ in range(4):
    t0 = time.perf_counter(); r = b.motion_search(pics); ts.append(time.perf_counter() - t0)
t0 = time.perf_counter(); want = ora.motion_search(pic); t_cpu = time.perf_counter() - t0
assert np.array_equal(r["packed"][n - 1], want)
px = n * 640 * 480
print(f"clips={n}: call (H2D pictures + kernel + D2H results) {min(ts) * 1e3:.2f} ms = {px / min(ts) / 1e6:.0f} Mpix/s analysed; "
      f"oracle on one host thread {t_cpu * 1e3:.1f} ms per picture = {640 * 480 / t_cpu / 1e6:.1f} Mpix/s")
