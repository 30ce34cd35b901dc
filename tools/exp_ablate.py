import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd.streamgen import BASE_SEED
def run(tag, debug, **kw):
    os.environ["MOBI_DEBUG"] = str(debug)
    clips, distinct = 512, 8
    b = m.MobiclipBatch(clips, 640, 480, 2)
    for i in range(distinct):
        p = m.default_params("B", BASE_SEED + i, n_frames=33, **kw); data, fo = m.generate_clip(p)
        assert all(r == 0 for r in b.preload(i, data, fo))
    for c in range(distinct, clips): b.preload_clone(c, c % distinct)
    b.commit(); b.replay(0)
    for f in range(1, 9): b.replay(f)
    b.sync(); b.set_kernel_timing(True); b.time_begin()
    for i in range(32): b.replay(1 + (i % 32))
    ms = b.time_end(); km = b.kernel_ms()
    print('%-40s debug=%d inter %.3f ms' % (tag, debug, km['inter_ms'] / max(1, km['inter_launches'])), flush=True)
    b.close()
copy = dict(pm_split1=0, pm_deep=0, cbp_prob=0, pm_intra=0)
for dbg in [int(x) for x in os.environ.get("DBGS", "0,1,2").split(",")]:
    run('pure copy', dbg, **copy)
if os.environ.get("DEFAULT_TOO"):
    for dbg in (0, 2):
        run('default', dbg)
