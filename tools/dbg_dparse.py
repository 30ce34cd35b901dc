import _prof  # noqa: F401  (the profiling twin of the library)
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd.decoder import load_library
lib = load_library()
lib.mobi_debug_read_parse.restype = C.c_longlong
lib.mobi_debug_read_parse.argtypes = [C.c_void_p] * 5 + [C.c_size_t]
n = 8
p = m.default_params("A", 12345, n_frames=3)
data, fo = m.generate_clip(p)
b = m.MobiclipBatch(n, 256, 192, 1, device_parse=True)
n_mbs = 16 * 12
for f in range(3):
    rcs, offs = b.decode([data[fo[f]:fo[f+1]]] * n, [0] * n)
    desc = np.zeros((n, n_mbs, 8), np.uint32); items = np.zeros((n, n_mbs), np.uint32); res = np.zeros((n, 8), np.uint32)
    cap = lib.mobi_debug_read_parse(b._h, desc.ctypes.data, items.ctypes.data, res.ctypes.data, None, 0)
    print("frame", f, "rcs", rcs, "offs", offs, "cap", cap)
    print(" res", res[:5].astype(np.int32).tolist())
    d = desc.copy()
    for c in range(n):
        d[c, :, 0] -= np.uint32(c * cap)
    for c in range(1, n):
        bad = np.argwhere((d[c] != d[1]).any(axis=1))
        if bad.size: print("  clip", c, "differs from clip 1 at MBs", bad[:6].ravel().tolist())
    bad = np.argwhere((d[0] != d[1]).any(axis=1)).ravel()
    print("  clip0 vs clip1 differing MBs:", bad[:10].tolist())
    for mb in bad[:3]:
        print("   mb", mb, "c0", [hex(x) for x in d[0, mb]], "c1", [hex(x) for x in d[1, mb]])
    ni = res[:, 2]
    print("  items c0", items[0, :8].tolist(), "c1", (items[1, :8] & 0x1FFF).tolist(), "n_intra", ni.tolist())
