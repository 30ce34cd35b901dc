import _prof  # noqa: F401  (the profiling twin of the library)
import os, sys, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MOBI_DEBUG"] = "9"
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd.streamgen import BASE_SEED
lib = m.load_library()
lib.mobi_debug_read_prof.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
import numpy as np
def run(tag, **kw):
    clips, distinct = int(os.environ.get('CLIPS', '512')), 8
    b = m.MobiclipBatch(clips, 640, 480, 2)
    for i in range(distinct):
        p = m.default_params("B", BASE_SEED + i, n_frames=33, **kw); data, fo = m.generate_clip(p)
        assert all(r == 0 for r in b.preload(i, data, fo))
    for c in range(distinct, clips): b.preload_clone(c, c % distinct)
    b.commit(); b.replay(0)
    for f in range(1, 6): b.replay(f)
    b.sync()
    junk = np.zeros((clips * 1200, 4), np.uint32)
    lib.mobi_debug_read_prof(b._h, junk.ctypes.data, junk.size)  # read-and-clear: drop the I-frame's records
    b.replay(6); b.sync()
    rec = np.zeros((clips * 300, 4), np.uint32)
    lib.mobi_debug_read_prof(b._h, rec.ctypes.data, rec.size)
    nent = rec[:, 2] >> 24
    r = rec.copy(); r[:, 2] &= 0xFFFFFF
    live = r.sum(1) > 0
    print(f"{tag}: quads {live.sum()}; mean cycles: issue {r[live,0].mean():.0f}  dma-wait {r[live,1].mean():.0f}  MC(4 MBs) {r[live,2].mean():.0f}  IDCT {r[live,3].mean():.0f}  total {r[live].sum(1).mean():.0f}; coded areas/quad {nent[live].mean():.1f} (>8: {(nent[live]>8).mean()*100:.0f}%)")
    print(f"    p50/p90: issue {np.median(r[live,0]):.0f}/{np.percentile(r[live,0],90):.0f} dma {np.median(r[live,1]):.0f}/{np.percentile(r[live,1],90):.0f} MC {np.median(r[live,2]):.0f}/{np.percentile(r[live,2],90):.0f} IDCT {np.median(r[live,3]):.0f}/{np.percentile(r[live,3],90):.0f}", flush=True)
    ir = np.zeros((clips * 1200 // 4, 4), np.uint32)
    full = np.zeros((clips * 1200, 4), np.uint32)
    lib.mobi_debug_read_prof(b._h, full.ctypes.data, full.size)
    ir = full[clips * 600:clips * 600 + clips * 150]
    li = ir.sum(1) > 0
    if li.any():
        print(f"    intra items {li.sum()}: mean cycles: dep-wait {ir[li,0].mean():.0f}  loads {ir[li,1].mean():.0f}  blocks {ir[li,2].mean():.0f}  store+publish {ir[li,3].mean():.0f}; p90 {np.percentile(ir[li,0],90):.0f}/{np.percentile(ir[li,1],90):.0f}/{np.percentile(ir[li,2],90):.0f}/{np.percentile(ir[li,3],90):.0f}", flush=True)
    b.close()
run('default')
run('pure copy', pm_split1=0, pm_deep=0, cbp_prob=0, pm_intra=0)
