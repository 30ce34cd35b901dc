"""Why does config C (848x480) not fit at 12288 clips after a failed attempt at 24576 in the same process?  Free HBM (hipMemGetInfo) at every step."""
import os, sys, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd.streamgen import BASE_SEED
hip = C.CDLL("libamdhip64.so")
def free_gb(tag):
    f, t = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    print(f"{tag}: free {f.value / 2**30:.1f} GiB of {t.value / 2**30:.1f}", flush=True)
streams = []
for i in range(4):
    p = m.default_params("C", BASE_SEED + i, n_frames=33); streams.append((p,) + m.generate_clip(p))
free_gb("start")
for clips in [int(a) for a in sys.argv[1:]] or [24576, 12288]:
    b = m.MobiclipBatch(clips, 848, 480, 2)
    free_gb(f"{clips}: batch created (rings)")
    try:
        for i, (p, data, fo) in enumerate(streams):
            b.preload(i, data, fo)
        for c in range(4, clips): b.preload_clone(c, c % 4)
        b.commit()
        free_gb(f"{clips}: committed")
        b.replay(0); print("sync", b.sync())
    except m.MobiclipError as e:
        print(f"{clips}: {e}")
        free_gb(f"{clips}: after the failure")
    b.close()
    free_gb(f"{clips}: closed")
