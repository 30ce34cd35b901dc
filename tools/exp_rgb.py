import sys, time; sys.path.insert(0,'/root/repo')
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd.streamgen import BASE_SEED
import torch
clips=512; distinct=8
b=m.MobiclipBatch(clips,640,480,2)
for i in range(distinct):
    p=m.default_params("B", BASE_SEED+i, n_frames=5); data,fo=m.generate_clip(p)
    assert all(r==0 for r in b.preload(i,data,fo))
for c in range(distinct,clips): b.preload_clone(c,c%distinct)
b.commit()
for f in range(5): b.replay(f)
b.sync()
for _ in range(3): b.convert_argb()
b.sync(); torch.cuda.synchronize()
t0=time.perf_counter(); n=20
for _ in range(n): b.convert_argb()
b.sync(); dt=(time.perf_counter()-t0)/n
px=clips*640*480
print(f'mobi_yuv_to_argb: {dt*1e3:.3f} ms per {clips} clips 640x480 -> {px/dt/1e9:.1f} Gpix/s, {px*5.5/dt/1e12:.2f} TB/s of algorithmic bytes (1.5 B read + 4 B written per pixel)')
b.close()
