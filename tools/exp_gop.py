"""Frame-parallel groups end to end (bitstreams in host memory -> planes in HBM): ms per frame step of mobi_batch_decode_gop and of the
pipelined mobi_batch_gop_begin / mobi_batch_gop_finish, against the step-by-step calls.

  python tools/exp_gop.py [clips] [K] [groups] [distinct] [config]        (GOP_KP, GOP_GEN, GOP_MODE, GOP_STEPWISE: below)
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import mobiclipdecoder_amd as m
from mobiclipdecoder_amd.streamgen import BASE_SEED

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
groups = int(sys.argv[3]) if len(sys.argv) > 3 else 4
distinct = int(sys.argv[4]) if len(sys.argv) > 4 else 64
config = sys.argv[5] if len(sys.argv) > 5 else "B"
mode = os.environ.get("GOP_MODE", "lockstep")
mode = int(mode) if mode.isdigit() else mode

n_frames = 1 + K * (groups + 3)
# GOP_GEN="iframe_interval=30,pm_intra=100": generator overrides (an I-frame every 30 frames -- the same frames in every clip -- instead of P-frames only)
gen_over = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("GOP_GEN", "").split(",") if kv)}
streams = []
for i in range(distinct):
    p = m.default_params(config, BASE_SEED + 100 + i, n_frames=n_frames, **gen_over)
    streams.append((p,) + m.generate_clip(p))
p0 = streams[0][0]
W, H = p0.width, p0.height


def packed_group(f0, k):
    bufs = [streams[c % distinct][1][streams[c % distinct][2][f0 + j]:streams[c % distinct][2][f0 + j + 1]] for j in range(k) for c in range(clips)]
    return bufs, (C.c_void_p * len(bufs))(*[x.ctypes.data for x in bufs]), (C.c_size_t * len(bufs))(*[x.size for x in bufs])


b = m.MobiclipBatch(clips, W, H, p0.version, device_parse=mode)
lib, h = b._lib, b._h
if hasattr(lib, "mobi_debug_phase_ms"):
    b.set_kernel_timing(1)
nv = clips * K
offs, outo, rcs = (C.c_int32 * nv)(), (C.c_int32 * nv)(), (C.c_int * nv)()
# the I-frame on its own, then groups
g = packed_group(0, 1)
assert lib.mobi_batch_decode_gop(h, 1, g[1], g[2], offs, rcs) == 0 and not any(rcs[:clips])
packs = [packed_group(1 + K * i, K) for i in range(groups + 3)]
ms = []
for i in range(groups + 1):
    for j in range(nv):
        offs[j] = 0
    t0 = time.perf_counter()
    assert lib.mobi_batch_decode_gop(h, K, packs[i][1], packs[i][2], offs, rcs) == 0
    ms.append((time.perf_counter() - t0) * 1e3)
    assert not any(rcs), "stream error"
if hasattr(lib, "mobi_debug_phase_ms"):  # MOBI_LIB=mobiclipdecoder_amd/libmobiclip_hip_prof.so: where the last group's time went
    for fn in ("mobi_debug_phase_ms", "mobi_debug_stage_ms", "mobi_debug_parse_ms"):
        getattr(lib, fn).restype = C.c_float
    lib.mobi_debug_phase_ms.argtypes = [C.c_void_p, C.c_int]
    lib.mobi_debug_stage_ms.argtypes = [C.c_void_p]
    lib.mobi_debug_parse_ms.argtypes = [C.c_void_p]
    ph = [lib.mobi_debug_phase_ms(h, k) for k in range(4)]
    print(f"  last group: gather + upload enqueue {lib.mobi_debug_stage_ms(h):.1f} ms; in finish: parse waited for at {ph[0]:.1f}, hand-overs done {ph[1]:.1f}, "
          f"reconstruction enqueued {ph[2]:.1f}, all done {ph[3]:.1f} ms; parse kernels (events) {lib.mobi_debug_parse_ms(h):.1f} ms")
sync_ms = float(np.median(ms[1:])) / K
print(f"{clips} clips x K={K} ({config}, {distinct} distinct, mode {mode}): mobi_batch_decode_gop {sync_ms:.3f} ms per frame step = {clips * W * H / sync_ms / 1e6:.1f} Gpixels/s "
      f"(groups: {[round(x, 1) for x in ms]} ms); host clips {b.host_clips()}, lock-step finished {b.lockstep_finished()} of {nv}")
b.close()

# pipelined: GOP_KP frames per group (default K; up to 128: what is parsed side by side is not bound by the ring, finish hands out six at a time)
KP = int(os.environ.get("GOP_KP", K))
GP = (groups + 3) * K // KP
ppacks = packs if KP == K else [packed_group(1 + KP * i, KP) for i in range(GP)]
nvp = clips * KP
offs, outo, rcs = (C.c_int32 * nvp)(), (C.c_int32 * nvp)(), (C.c_int * nvp)()
b = m.MobiclipBatch(clips, W, H, p0.version, device_parse=mode)
lib, h = b._lib, b._h
assert lib.mobi_batch_decode_gop(h, 1, g[1], g[2], offs, rcs) == 0
C.memset(offs, 0, C.sizeof(offs))


def finish():
    pending = lib.mobi_batch_gop_frames_pending(h)
    while pending > 0:
        part = min(6, pending)
        e = lib.mobi_batch_gop_finish(h, outo, rcs)
        bad = np.flatnonzero(np.frombuffer(rcs, dtype=np.int32)[:part * clips])
        assert e == 0 and not bad.size, (e, b._lib.mobi_error_string(e), bad.size, [(int(j), rcs[j]) for j in bad[:8]])
        pending -= part


# both slots' buffers exist before the clock starts: groups 0 and 1 are begun and group 0 finished untimed
assert lib.mobi_batch_gop_begin(h, KP, ppacks[0][1], ppacks[0][2], offs) == 0
assert lib.mobi_batch_gop_begin(h, KP, ppacks[1][1], ppacks[1][2], offs) == 0
finish()
t0 = time.perf_counter()
for i in range(2, GP):
    assert lib.mobi_batch_gop_begin(h, KP, ppacks[i][1], ppacks[i][2], offs) == 0
    finish()
pipe_ms = (time.perf_counter() - t0) * 1e3 / ((GP - 2) * KP)
finish()
print(f"  pipelined, {KP} frames per group (gop_begin of group g + 1 before the gop_finish calls of group g): {pipe_ms:.3f} ms per frame step = {clips * W * H / pipe_ms / 1e6:.1f} Gpixels/s")
# the newest frame against a step-by-step batch's (a few clips)
y = [b.planes(c, 0) for c in range(min(clips, 4))]
b.close()
if os.environ.get("GOP_STEPWISE", "1") != "0":
    b = m.MobiclipBatch(clips, W, H, p0.version, device_parse=mode)
    ms = []
    last = 1 + KP * GP
    for f in range(last):
        datas = [streams[c % distinct][1][streams[c % distinct][2][f]:streams[c % distinct][2][f + 1]] for c in range(clips)]
        r, _ = b.decode(datas, [0] * clips)
        assert not any(r)
        ms.append(b.last_decode_ms())
    print(f"  step by step (mobi_batch_decode): {float(np.median(ms[2:])):.3f} ms per frame step")
    for c in range(min(clips, 4)):
        yy = b.planes(c, 0)
        assert np.array_equal(y[c][0], yy[0]) and np.array_equal(y[c][1], yy[1]), "the pipelined groups' newest frame differs from the step-by-step batch's"
    print("  newest frame identical to the step-by-step batch's (first clips)")
    b.close()
