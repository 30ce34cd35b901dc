import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd.streamgen import BASE_SEED
from tests.oracle_binding import OracleDecoder
def run(tag, cfg, **kw):
    p = m.default_params(cfg, BASE_SEED + 7, **kw); data, fo = m.generate_clip(p)
    g = m.MobiclipDecoder(p.width, p.height, p.version); o = OracleDecoder(p.width, p.height, p.version)
    for f in range(p.n_frames):
        g.Data = o.Data = data[:fo[f+1]]; g.Offset = o.Offset = int(fo[f])
        a, b = g.DecodeFrame(), o.DecodeFrame()
        if a is None or b is None: print(tag, f, 'err', g.last_error, o.last_error); break
        dy = np.argwhere(a[0] != b[0]); du = np.argwhere(a[1] != b[1])
        if len(dy) or len(du):
            print(tag, 'frame', f, 'Y mismatches', len(dy), dy[:6].tolist(), 'UV', len(du), du[:6].tolist())
            if len(dy): y,x = dy[0]; print('   gpu', a[0][y, x:x+8].tolist(), 'ora', b[0][y, x:x+8].tolist())
            if len(du): y,x = du[0]; print('   gpu uv', a[1][y, x:x+8].tolist(), 'ora', b[1][y, x:x+8].tolist())
            break
    else: print(tag, 'OK')
    g.close()
base = dict(n_frames=4, width=64, height=48)
run('copy int-mv', 'A', pm_split1=0, pm_deep=0, cbp_prob=0, pm_intra=0, mv_range=0, **base)
run('copy half-pel', 'A', pm_split1=0, pm_deep=0, cbp_prob=0, pm_intra=0, **base)
run('single+resid8', 'A', pm_split1=0, pm_deep=0, pm_intra=0, t8_prob=1000, **base)
run('single+resid4', 'A', pm_split1=0, pm_deep=0, pm_intra=0, t8_prob=0, **base)
run('split1 noresid', 'A', pm_split1=1000, pm_deep=0, cbp_prob=0, pm_intra=0, **base)
run('deep noresid', 'A', pm_split1=0, pm_deep=1000, cbp_prob=0, pm_intra=0, **base)
run('default 64x48', 'A', **base)
run('default A', 'A', n_frames=4)
run('default B', 'B', n_frames=4)
