#!/usr/bin/env python3
"""mobi_batch_decode with the parse on host threads: time inside the C call per P-frame step (MOBI_PARSE_THREADS sets the pool)."""
import _prof  # noqa: F401  (the profiling twin of the library)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd import sharding
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
streams = [m.generate_clip(m.default_params("B", sharding.stream_seed("B", 0, i), n_frames=7)) for i in range(16)]
b = m.MobiclipBatch(n, 640, 480, 2, device_parse=False)
ms, pm, sm, ph = [], [], [], []
import ctypes as C
lib = m.load_library()
for fn in (lib.mobi_debug_hostparse_ms, lib.mobi_debug_stage_ms):
    fn.restype = C.c_float; fn.argtypes = [C.c_void_p]
lib.mobi_debug_phase_ms.restype = C.c_float; lib.mobi_debug_phase_ms.argtypes = [C.c_void_p, C.c_int]
for f in range(7):
    datas = [streams[c % 16][0][streams[c % 16][1][f]:streams[c % 16][1][f + 1]] for c in range(n)]
    rcs, _ = b.decode(datas, [0] * n)
    assert all(r == 0 for r in rcs)
    ms.append(b.last_decode_ms()); pm.append(lib.mobi_debug_hostparse_ms(b._h)); sm.append(lib.mobi_debug_stage_ms(b._h)); ph.append([lib.mobi_debug_phase_ms(b._h, k) for k in range(5)])
t = float(np.median(ms[2:]))
print(f"host parse, {n} clips, MOBI_PARSE_THREADS={os.environ.get('MOBI_PARSE_THREADS', 'default')}: {t:.2f} ms per step inside the C call = {n * 640 * 480 / t / 1e3:.0f} Mpix/s (parse {np.median(pm[2:]):.2f} ms, plan + staging {np.median(sm[2:]):.2f} ms; phases parse loop / plan / stage + upload calls / launch calls / sync: " + " / ".join(f"{v:.2f}" for v in np.median(np.array(ph[2:]), axis=0)) + ")")
b.close()
