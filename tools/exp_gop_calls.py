"""Wall time of every call of a pipelined run of frame-parallel groups (gop_begin, and each gop_finish part), to see where a group's time goes
on the host's clock.   python tools/exp_gop_calls.py [clips] [frames per group] [groups]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd.streamgen import BASE_SEED

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
KP = int(sys.argv[2]) if len(sys.argv) > 2 else 12
GP = int(sys.argv[3]) if len(sys.argv) > 3 else 8
distinct = 64
streams = []
for i in range(distinct):
    p = m.default_params("B", BASE_SEED + 100 + i, n_frames=1 + KP * GP)
    streams.append((p,) + m.generate_clip(p))
p0 = streams[0][0]


def pack(f0, k):
    bufs = [streams[c % distinct][1][streams[c % distinct][2][f0 + j]:streams[c % distinct][2][f0 + j + 1]] for j in range(k) for c in range(clips)]
    return bufs, (C.c_void_p * len(bufs))(*[x.ctypes.data for x in bufs]), (C.c_size_t * len(bufs))(*[x.size for x in bufs])


b = m.MobiclipBatch(clips, p0.width, p0.height, p0.version, device_parse="lockstep")
lib, h = b._lib, b._h
nv = clips * KP
offs, outo, rcs = (C.c_int32 * nv)(), (C.c_int32 * nv)(), (C.c_int * nv)()
g = pack(0, 1)
assert lib.mobi_batch_decode_gop(h, 1, g[1], g[2], offs, rcs) == 0
C.memset(offs, 0, C.sizeof(offs))
packs = [pack(1 + KP * i, KP) for i in range(GP)]
log = []
T0 = time.perf_counter()


def call(name, f):
    t0 = time.perf_counter()
    r = f()
    log.append((name, (t0 - T0) * 1e3, (time.perf_counter() - t0) * 1e3))
    return r


def finish(gi):
    pending = lib.mobi_batch_gop_frames_pending(h)
    part = 0
    while pending > 0:
        assert call(f"finish {gi}.{part}", lambda: lib.mobi_batch_gop_finish(h, outo, rcs)) == 0
        pending -= min(6, pending)
        part += 1


call("begin 0", lambda: lib.mobi_batch_gop_begin(h, KP, packs[0][1], packs[0][2], offs))
for i in range(1, GP):
    call(f"begin {i}", lambda: lib.mobi_batch_gop_begin(h, KP, packs[i][1], packs[i][2], offs))
    finish(i - 1)
finish(GP - 1)
for name, at, dur in log:
    print(f"{at:9.1f} ms  {name:12s} {dur:7.1f} ms")
per = [log[k + 1][1] - log[k][1] for k in range(len(log) - 1)]
starts = [at for name, at, _ in log if name.startswith("begin")]
print("group periods (begin to begin):", [round(starts[k + 1] - starts[k], 1) for k in range(len(starts) - 1)], "ms; per frame step in the steady state:",
      round((starts[-1] - starts[2]) / ((len(starts) - 3) * KP), 3) if len(starts) > 3 else None)
b.close()
