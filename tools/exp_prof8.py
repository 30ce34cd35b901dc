"""MOBI_DEBUG=9: where an octet wave's life goes in mobi_recon_inter8 (shader-clock stamps accumulated by the _prof twin of the kernel)."""
import _prof  # noqa: F401  (the profiling twin of the library)
import os, sys, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MOBI_DEBUG"] = "9"
import numpy as np
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd.streamgen import BASE_SEED
lib = m.load_library()
lib.mobi_debug_read_prof.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
def run(tag, clips=int(os.environ.get("CLIPS", "2048")), **kw):
    distinct = 8
    b = m.MobiclipBatch(clips, 640, 480, 2)
    for i in range(distinct):
        p = m.default_params("B", BASE_SEED + i, n_frames=33, **kw); data, fo = m.generate_clip(p)
        assert all(r == 0 for r in b.preload(i, data, fo))
    for c in range(distinct, clips): b.preload_clone(c, c % distinct)
    b.commit(); b.replay(0)
    for f in range(1, 6): b.replay(f)
    b.sync()
    nrec = clips * 150
    junk = np.zeros((nrec, 8), np.uint64)
    lib.mobi_debug_read_prof(b._h, junk.ctypes.data, junk.size * 2)  # read and clear
    b.replay(6)
    b.sync()
    recs = np.zeros((nrec, 8), np.uint64)
    lib.mobi_debug_read_prof(b._h, recs.ctypes.data, recs.size * 2)
    live = (recs[:, 6] & 0xFF) == 1
    extra = recs[live, 6]
    recs = recs[live]
    rec = recs.sum(0)
    w = float(live.sum())
    print(f"    inside A: kernel arguments {float(((extra >> 8) & 0xFFFFFF).sum()) / w:.0f}  descriptor {float((extra >> 32).sum()) / w:.0f}")
    names = ["A issue", "fetch wait", "MC", "deep", "residual", "store issue"]
    tot = sum(float(rec[k]) for k in range(6))
    print(f"{tag}: {int(w)} waves, {tot / w:.0f} cycles per wave; " + "  ".join(f"{names[k]} {float(rec[k]) / w:.0f}" for k in range(6)) + f"; coded areas per octet {float(rec[7]) / w:.1f}", flush=True)
    b.close()
run("default")
run("no-deep", pm_deep=0)
run("single-leaf", pm_split1=0, pm_deep=0)
run("pure copy", pm_split1=0, pm_deep=0, cbp_prob=0, pm_intra=0)
