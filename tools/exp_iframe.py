"""I-frame (every macroblock intra: ~100 dependency levels per clip) of N clips 640x480 through the replay path: launch time of mobi_recon_intra."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd.streamgen import BASE_SEED
clips, distinct = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 8
b = m.MobiclipBatch(clips, 640, 480, 2)
for i in range(distinct):
    p = m.default_params("B", BASE_SEED + i, n_frames=3); data, fo = m.generate_clip(p)
    assert all(r == 0 for r in b.preload(i, data, fo))
for c in range(distinct, clips): b.preload_clone(c, c % distinct)
b.commit(); b.replay(0); b.replay(1); b.sync()
b.set_kernel_timing(2); b.time_begin()
n = 4
for i in range(n): b.replay(0)
ms = b.time_end(); km = b.kernel_ms()
print(f"I-frame, {clips} clips: {ms / n:.3f} ms per step, intra launch {km['intra_ms'] / max(1, km['intra_launches']):.3f} ms ({clips * 1200 / (km['intra_ms'] / max(1, km['intra_launches'])) / 1e3:.0f} macroblocks per us)")
b.close()
