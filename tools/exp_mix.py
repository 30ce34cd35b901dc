import sys, time; sys.path.insert(0,'/root/repo')
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd.streamgen import BASE_SEED
def run(tag, **kw):
    clips=512; distinct=8
    b=m.MobiclipBatch(clips,640,480,2)
    for i in range(distinct):
        p=m.default_params("B", BASE_SEED+i, n_frames=33, **kw); data,fo=m.generate_clip(p)
        assert all(r==0 for r in b.preload(i,data,fo))
    for c in range(distinct,clips): b.preload_clone(c,c%distinct)
    b.commit(); b.replay(0)
    for f in range(1,9): b.replay(f)
    b.sync(); b.set_kernel_timing(2); b.time_begin()
    for i in range(32): b.replay(1+(i%32))
    ms=b.time_end(); km=b.kernel_ms()
    print(tag, 'step %.3f ms  inter %.3f ms  intra %.3f ms/step (%d launches)'%(ms/32, km['inter_ms']/max(1,km['inter_launches']), km['intra_ms']/32, km['intra_launches']/32), flush=True)
    b.close()
run('default')
run('single-leaf', pm_split1=0, pm_deep=0)
run('no-deep', pm_deep=0)
run('single-leaf,no-intra', pm_split1=0, pm_deep=0, pm_intra=0)
run('single-leaf,no-resid', pm_split1=0, pm_deep=0, cbp_prob=0)
run('single-leaf,no-resid,no-intra,int-mv', pm_split1=0, pm_deep=0, cbp_prob=0, pm_intra=0)
