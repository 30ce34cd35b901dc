"""End to end, asynchronous: where does the wall time of a step go?  python tools/exp_async.py [clips] [steps]
Prints per step: Python-side preparation, mobi_batch_submit, mobi_batch_wait (ms)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd import sharding

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
streams = []
ND = int(os.environ.get("DISTINCT", 64))  # a lock-step wave must not hold copies of one stream
for i in range(ND):
    p = m.default_params("B", sharding.stream_seed("B", 0, i), n_frames=3 + steps, **({"iframe_interval": 1} if os.environ.get("IFRAMES") else {}))  # IFRAMES=1: every frame an I-frame
    d, fo = m.generate_clip(p)
    streams.append((d, fo))
b = m.MobiclipBatch(n, 640, 480, p.version, device_parse="lockstep" if os.environ.get("LOCKSTEP") else True)  # LOCKSTEP=1: the lock-step parser in front
lib, h = b._lib, b._h
# ctypes arrays built once per frame, outside the timed calls: what a C caller would hand over
def pack(f):
    bufs = [streams[c % ND][0][streams[c % ND][1][f]:streams[c % ND][1][f + 1]] for c in range(n)]
    ptrs = (C.c_void_p * n)(*[x.ctypes.data for x in bufs])
    lens = (C.c_size_t * n)(*[x.size for x in bufs])
    return bufs, ptrs, lens
offs = (C.c_int32 * n)()
rcs = (C.c_int * n)()
out = (C.c_int32 * n)()
packed = [pack(f) for f in range(3 + steps)]
for f in range(2):
    assert lib.mobi_batch_submit(h, packed[f][1], packed[f][2], offs) == 0
    assert lib.mobi_batch_wait(h, out, rcs) == 0
t0 = time.perf_counter()
ts = time.perf_counter(); assert lib.mobi_batch_submit(h, packed[2][1], packed[2][2], offs) == 0; print("first submit %.2f ms" % ((time.perf_counter() - ts) * 1e3))
for f in range(3, 2 + steps):
    ts = time.perf_counter()
    assert lib.mobi_batch_submit(h, packed[f][1], packed[f][2], offs) == 0
    tw = time.perf_counter()
    assert lib.mobi_batch_wait(h, out, rcs) == 0
    te = time.perf_counter()
    assert all(r == 0 for r in rcs)
    print("step %2d: submit %.2f ms  wait %.2f ms" % (f, (tw - ts) * 1e3, (te - tw) * 1e3))
assert lib.mobi_batch_wait(h, out, rcs) == 0
t = (time.perf_counter() - t0) * 1e3 / steps
print("asynchronous: %.2f ms per step of %d clips = %.1f Gpixels/s" % (t, n, n * 640 * 480 / t / 1e6))
# synchronous, same frames again is not possible (decoder state moved on): a fresh batch
b.close()
b = m.MobiclipBatch(n, 640, 480, p.version, device_parse="lockstep" if os.environ.get("LOCKSTEP") else True)  # LOCKSTEP=1: the lock-step parser in front
lib, h = b._lib, b._h
for f in range(2):
    assert lib.mobi_batch_decode(h, packed[f][1], packed[f][2], offs, rcs) == 0
    for i in range(n): offs[i] = 0
t0 = time.perf_counter()
for f in range(2, 2 + steps):
    assert lib.mobi_batch_decode(h, packed[f][1], packed[f][2], offs, rcs) == 0
    for i in range(n): offs[i] = 0
t = (time.perf_counter() - t0) * 1e3 / steps
print("synchronous (the Python loop that resets Offset included): %.2f ms per step = %.1f Gpixels/s; inside the last call %.2f ms" % (t, n * 640 * 480 / t / 1e6, b.last_decode_ms()))
b.close()
