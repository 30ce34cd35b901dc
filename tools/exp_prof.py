import os, sys, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MOBI_DEBUG"] = "9"
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd.streamgen import BASE_SEED
lib = m.load_library()
lib.mobi_debug_read_prof.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
import numpy as np
def run(tag, **kw):
    clips, distinct = 512, 8
    b = m.MobiclipBatch(clips, 640, 480, 2)
    for i in range(distinct):
        p = m.default_params("B", BASE_SEED + i, n_frames=33, **kw); data, fo = m.generate_clip(p)
        assert all(r == 0 for r in b.preload(i, data, fo))
    for c in range(distinct, clips): b.preload_clone(c, c % distinct)
    b.commit(); b.replay(0)
    for f in range(1, 6): b.replay(f)
    b.sync()
    rec = np.zeros((clips * 1200, 4), np.uint32)
    lib.mobi_debug_read_prof(b._h, rec.ctypes.data, rec.size)
    multi = (rec[:, 0] >> 31).astype(bool); coded = (rec[:, 2] >> 31).astype(bool)
    r = rec & 0x7FFFFFFF
    live = r.sum(1) > 0
    print(f"{tag}: inter MBs {live.sum()}; mean cycles: desc {r[live,0].mean():.0f}  pixels+MC {r[live,1].mean():.0f}  residual {r[live,2].mean():.0f}  store-drain {r[live,3].mean():.0f}  total {r[live].sum(1).mean():.0f}")
    for name, msk in (("single-leaf", live & ~multi), ("multi-leaf", live & multi)):
        if msk.any(): print(f"    {name:12s} n={msk.sum():7d} pixels+MC mean {r[msk,1].mean():.0f} p50 {np.median(r[msk,1]):.0f} p90 {np.percentile(r[msk,1],90):.0f}")
    for name, msk in (("uncoded", live & ~coded), ("coded", live & coded)):
        if msk.any(): print(f"    {name:12s} n={msk.sum():7d} residual mean {r[msk,2].mean():.0f} p50 {np.median(r[msk,2]):.0f} p90 {np.percentile(r[msk,2],90):.0f}")
    print(f"    desc p50 {np.median(r[live,0]):.0f} p90 {np.percentile(r[live,0],90):.0f}; store-drain p50 {np.median(r[live,3]):.0f} p90 {np.percentile(r[live,3],90):.0f}", flush=True)
    b.close()
run('default')
run('pure copy', pm_split1=0, pm_deep=0, cbp_prob=0, pm_intra=0)
