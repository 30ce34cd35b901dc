import sys, time; sys.path.insert(0,'/root/repo')
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd.streamgen import BASE_SEED
import torch
def mk(clips, distinct=8, seed0=0):
    b=m.MobiclipBatch(clips,640,480,2)
    for i in range(distinct):
        p=m.default_params("B", BASE_SEED+seed0+i, n_frames=33); data,fo=m.generate_clip(p)
        assert all(r==0 for r in b.preload(i,data,fo))
    for c in range(distinct,clips): b.preload_clone(c,c%distinct)
    b.commit(); b.replay(0)
    for f in range(1,9): b.replay(f)
    b.sync(); return b
def run(nb, clips_each, steps=64):
    bs=[mk(clips_each, seed0=16*i) for i in range(nb)]
    torch.cuda.synchronize(); t0=time.perf_counter()
    for i in range(steps):
        for b in bs: b.replay(1+(i%32))
    for b in bs: b.sync()
    torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print(f'{nb} batch(es) x {clips_each} clips: {dt*1e3/steps:.3f} ms per step of {nb*clips_each} clips -> {nb*clips_each*steps*640*480/dt/1e9:.1f} Gpix/s', flush=True)
    for b in bs: b.close()
sizes = [int(a) for a in sys.argv[1:]] or [512]
for n in sizes:
    run(1, n); run(2, n // 2); run(4, n // 4)
