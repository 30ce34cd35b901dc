"""Does a launch care what the chip did just before?  The headline's P-frame steps (command lists resident in HBM) back to back, then with the host asleep
between them: mobi_recon_inter8 / mobi_recon_intra per launch (HIP events).  python tools/exp_gap.py [clips]
r05: asked because mobi_recon_inter8 takes 7.5 ms in device-parsed steps and 6.4 in the headline's loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd import sharding

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
streams = []
for i in range(16):
    p = m.default_params("B", sharding.stream_seed("B", 0, i), n_frames=33)
    streams.append((p,) + m.generate_clip(p))
b = m.MobiclipBatch(n, 640, 480, streams[0][0].version)
for i, (p, data, fo) in enumerate(streams):
    assert all(r == 0 for r in b.preload(i, data, fo))
for c in range(16, n):
    b.preload_clone(c, c % 16)
b.commit()
b.replay(0)
for f in range(1, 9):
    b.replay(f)
assert b.sync() == 0
b.set_kernel_timing(2)
for gap_ms in (0, 1, 5, 30, 100, 0):
    out = []
    for f in range(9, 25):
        b.time_begin()
        b.replay(f)
        b.time_end()
        km = b.kernel_ms()
        out.append(km["inter_ms"])
        if gap_ms:
            time.sleep(gap_ms / 1e3)
    b.replay(0)
    for f in range(1, 9):
        b.replay(f)
    b.sync()
    print(f"host asleep {gap_ms:3d} ms between steps: mobi_recon_inter8 per launch min {min(out):.3f} median {sorted(out)[len(out) // 2]:.3f} max {max(out):.3f} ms", flush=True)
b.close()
