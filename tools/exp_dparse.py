#!/usr/bin/env python3
"""End-to-end DecodeFrame throughput (bitstream in, planes in HBM): host parse vs device parse at several batch sizes."""
import _prof  # noqa: F401  (the profiling twin of the library)
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd import sharding

def run(n_clips, device_parse, n_frames=9, distinct=16):
    streams = []
    for i in range(distinct):
        over = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("BENCH_GEN", "").split(",") if kv)}  # experiment hook, as in bench.py
        p = m.default_params("B", sharding.stream_seed("B", 0, i), n_frames=n_frames, **over)
        streams.append(m.generate_clip(p))
    b = m.MobiclipBatch(n_clips, 640, 480, 2, device_parse=device_parse)
    b.set_kernel_timing(1)
    import ctypes as C
    from mobiclipdecoder_amd.decoder import load_library
    lib = load_library(); lib.mobi_debug_parse_ms.restype = C.c_float; lib.mobi_debug_parse_ms.argtypes = [C.c_void_p]
    lib.mobi_debug_stage_ms.restype = C.c_float; lib.mobi_debug_stage_ms.argtypes = [C.c_void_p]
    c_ms, s_ms, ph = [], [], []
    lib.mobi_debug_phase_ms.restype = C.c_float; lib.mobi_debug_phase_ms.argtypes = [C.c_void_p, C.c_int]
    t_frames, k_ms = [], []
    for f in range(n_frames):
        datas = [streams[c % distinct][0][streams[c % distinct][1][f]:streams[c % distinct][1][f + 1]] for c in range(n_clips)]
        offs = [0] * n_clips
        t0 = time.perf_counter()
        rcs, _ = b.decode(datas, offs)
        t_frames.append(time.perf_counter() - t0)
        k_ms.append(lib.mobi_debug_parse_ms(b._h)); c_ms.append(b.last_decode_ms()); s_ms.append(lib.mobi_debug_stage_ms(b._h)); ph.append([lib.mobi_debug_phase_ms(b._h, k) for k in range(5)])
        assert all(r == 0 for r in rcs), rcs[:8]
    b.close()
    p_ms = np.array(t_frames[2:]) * 1e3   # skip the I-frame and the first P-frame (allocations)
    px = n_clips * 640 * 480
    print(f"clips={n_clips:5d} device_parse={device_parse if isinstance(device_parse, str) else int(device_parse)}  I-frame {t_frames[0]*1e3:8.2f} ms   P-frame median {np.median(p_ms):8.2f} ms  "
          f"min {p_ms.min():8.2f} ms  -> {px / np.median(p_ms) / 1e3:9.1f} Mpix/s end to end"
          + f" | inside the C call: P median {np.median(c_ms[2:]):.2f} ms = {px / np.median(c_ms[2:]) / 1e3:.0f} Mpix/s"
          + (f" | parse kernel: I {k_ms[0]:.2f} ms, P median {np.median(k_ms[2:]):.2f} ms, staging {np.median(s_ms[2:]):.2f} ms | cumulative ms after gather + upload / parse enqueued / parse done / launches enqueued / done: " + " / ".join(f"{v:.2f}" for v in np.median(np.array(ph[2:]), axis=0)) if device_parse else ""), flush=True)

if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    modes = (False,) if "--host-only" in sys.argv else (True,) if "--device-only" in sys.argv else ("hybrid",) if "--hybrid" in sys.argv else ("lockstep",) if "--lockstep" in sys.argv else \
            (True, "lockstep") if "--device-both" in sys.argv else (False, True)
    sizes = [int(a) for a in args] or [512, 2048]
    distinct = int(os.environ.get("DISTINCT", "64"))  # streams the clips are copies of (a lock-step wave of 64 lanes holds min(64, DISTINCT) different ones)
    for n in sizes:
        for dp in modes:
            run(n, dp, distinct=distinct)
