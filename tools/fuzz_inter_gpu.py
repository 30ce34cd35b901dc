"""Randomised GPU-vs-oracle sweep aimed at the inter path: many small streams (every partition shape, deep trees, older references,
all half-pel phases, vectors that leave the picture and read its zero padding, widths that leave the last octet partly empty -- 80,
144, 272, 528, 848 -- so that the zeros the octet kernel stores there are read back by the next frames, width == stride), a few intra
macroblocks in between, batches of 1..7 clips, host-parsed and device-parsed alternately.  Stops at the first difference.
python tools/fuzz_inter_gpu.py [rounds] [seed0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd.streamgen import BASE_SEED
from tests.oracle_binding import OracleDecoder

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
geoms = [(16, 16, 2), (80, 48, 1), (144, 64, 2), (256, 32, 1), (512, 32, 2), (160, 112, 2), (272, 48, 1), (128, 128, 2), (848, 32, 2), (528, 48, 2), (1024, 32, 2), (640, 48, 2)]
rng = np.random.default_rng(1234 + seed0)
t0, frames, mbs = time.time(), 0, 0
for it in range(rounds):
    w, h, ver = geoms[int(rng.integers(len(geoms)))]
    nclips = int(rng.integers(1, 8))
    nfr = int(rng.integers(4, 9))
    kw = dict(width=w, height=h, version=ver, n_frames=nfr, pm_intra=int(rng.choice([0, 50, 200])), pm_deep=int(rng.choice([0, 100, 400])), pm_multiref=int(rng.choice([0, 200, 600])), pm_skip=int(rng.choice([0, 150, 500])), pm_split1=int(rng.choice([100, 400])), intra_sub_prob=int(rng.choice([100, 500, 900])),
              plane_prob=int(rng.choice([0, 300, 700])), iframe_interval=int(rng.choice([0, 0, 4])), mv_range=int(rng.choice([2, 12, 40, 90])),
              cbp_prob=int(rng.choice([100, 300, 700])), dense_prob=int(rng.choice([0, 200])), qdelta_prob=int(rng.choice([0, 300])),
              table1_prob=int(rng.choice([0, 500])), escape_prob=int(rng.choice([0, 80])), quantizer=int(rng.choice([12, 18, 25, 29, 33, 36, 40, 46, 52])),
              intra_dc_only=int(rng.choice([0, 0, 1])), edge_mode=int(rng.choice([0, 1])))
    ps = [m.default_params("A", BASE_SEED + 100000 + 1000 * (seed0 + it) + i, **kw) for i in range(nclips)]
    clips = [m.generate_clip(p) for p in ps]
    dev = bool(it & 1)
    if it & 2:  # host-parsed rounds: a small step is ONE launch (mobi_recon_step) unless the limit is 0; both kinds in turn
        os.environ["MOBI_FUSED_STEP_MBS"] = "0"
    else:
        os.environ.pop("MOBI_FUSED_STEP_MBS", None)
    b = m.MobiclipBatch(nclips, w, h, ver, device_parse=dev)
    oras = [OracleDecoder(w, h, ver) for _ in ps]
    for f in range(nfr):
        rcs, offs = b.decode([c[0][c[1][f]:c[1][f + 1]] for c in clips], [0] * nclips)
        for i in range(nclips):
            oras[i].Data, oras[i].Offset = clips[i][0][clips[i][1][f]:clips[i][1][f + 1]], 0
            o = oras[i].DecodeFrame()
            ok = rcs[i] == 0 and o is not None and offs[i] == oras[i].Offset
            if ok:
                y, uv = b.planes(i)
                ok = np.array_equal(y, o[0]) and np.array_equal(uv, o[1])
            if not ok:
                print("DIFFERENCE round", it, "frame", f, "clip", i, "device_parse", dev, "two_launches", bool(it & 2), "rc", rcs[i], oras[i].last_error, kw, "seed", ps[i].seed)
                sys.exit(1)
        frames += nclips
        mbs += nclips * (w // 16) * (h // 16)
    b.close()
    for o in oras:
        o.close()
print("fuzz_inter_gpu: %d rounds, %d clip-frames, %d macroblocks, no difference, %.0f s" % (rounds, frames, mbs, time.time() - t0))
