import _prof  # noqa: F401  (the profiling twin of the library)
import os, sys, ctypes as C; sys.path.insert(0, '/root/repo')
os.environ["MOBI_DEBUG"] = "9"
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd.streamgen import BASE_SEED
import numpy as np
lib = m.load_library()
lib.mobi_debug_read_prof.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
clips, distinct = 512, 8
b = m.MobiclipBatch(clips, 640, 480, 2)
for i in range(distinct):
    p = m.default_params("B", BASE_SEED + i, n_frames=8); data, fo = m.generate_clip(p)
    assert all(r == 0 for r in b.preload(i, data, fo))
for c in range(distinct, clips): b.preload_clone(c, c % distinct)
b.commit(); b.replay(0)
for f in range(1, 6): b.replay(f)
b.sync()
full = np.zeros((clips * 1200, 4), np.uint32)
lib.mobi_debug_read_prof(b._h, full.ctypes.data, full.size)
b.replay(6); b.sync()
lib.mobi_debug_read_prof(b._h, full.ctypes.data, full.size)
ir = full[clips * 300:]
live = np.nonzero(ir.sum(1))[0]
K = None
# infer K: items index = clip*K+slot ; live indices cluster
d = np.diff(live); 
print('n live', len(live), 'max idx', live.max())
for K in range(40, 120):
    if (live // K).max() == clips - 1 and (live % K).max() < K: pass
cl = None
# brute: find K s.t. clip counts uniform
best = None
for K in range(30, 200):
    c = live // K
    if c.max() != clips - 1: continue
    cnt = np.bincount(c, minlength=clips)
    if best is None or cnt.std() < best[1]: best = (K, cnt.std())
K = best[0]; print('K', K)
clip = live // K; slot = live % K
w = ir[live, 0].astype(np.float64)
print('dep-wait by slot decile:', [int(w[(slot >= q * K // 10) & (slot < (q + 1) * K // 10)].mean()) for q in range(10)])
lc = clip % 64
print('dep-wait by clip-in-xcd (0..63) octiles:', [int(w[(lc >= 8 * q) & (lc < 8 * q + 8)].mean()) for q in range(8)])
print('by xcd:', [int(w[clip // 64 == x].mean()) for x in range(8)])
print('overall mean', int(w.mean()), 'median', int(np.median(w)), 'p10', int(np.percentile(w, 10)))
b.close()
