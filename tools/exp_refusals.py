"""How often does the product refuse (MOBI_E_UNSUPPORTED) a frame the reference decodes?  CPU only: product parser + kernel
arithmetic (tests/tools/mobi_cmd_interp.cpp) against the oracle on corrupted streams.  A refusal is counted as a result difference
only when the oracle decoded the same frame without throwing."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mobiclipdecoder_amd import default_params, generate_clip
from mobiclipdecoder_amd.streamgen import BASE_SEED
from tests.interp_binding import InterpDecoder
from tests.oracle_binding import OracleDecoder
rng = np.random.default_rng(2026)
frames = refused = refused_ref_ok = both_ok = both_err = 0
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1500):
    ver = 1 + trial % 2
    p = default_params("AB"[trial % 2], BASE_SEED + 9000 + trial, n_frames=4, width=96, height=64, version=ver, pm_intra=120, pm_deep=150,
                       pm_multiref=250, qdelta_prob=250, escape_prob=80, table1_prob=400)
    data, fo = generate_clip(p)
    d = data.copy()
    for _ in range(int(rng.integers(1, 8))):
        d[int(rng.integers(0, d.size))] ^= 1 << int(rng.integers(0, 8))
    a, o = InterpDecoder(p.width, p.height, p.version), OracleDecoder(p.width, p.height, p.version)
    for f in range(p.n_frames):
        a.Data = o.Data = d[: fo[f + 1]]
        a.Offset = o.Offset = int(fo[f])
        a.DecodeFrame(); o.DecodeFrame()
        frames += 1
        if a.last_error == -6:
            refused += 1
            refused_ref_ok += o.last_error == 0
            break
        if a.last_error != 0 or o.last_error != 0:
            both_err += 1
            break
        both_ok += 1
import ctypes as C
from tests.interp_binding import lib as interp_lib
cnt = (C.c_ulong * 4)()
interp_lib().mobi_cmdinterp_refusals(cnt)
print("refusals by cause (all refusing frames, whether or not the reference decodes them): |MV| > 8191: %d, ModsDS quantiser < 12: %d, "
      "run past the block: %d, plane parameter outside int16: %d" % tuple(cnt))
print(f"{frames} corrupted frames: {both_ok} decoded by both, {both_err} rejected by both, {refused} refused by the product only "
      f"({refused_ref_ok} of them decoded by the reference's restatement = result differences: {100.0 * refused_ref_ok / frames:.2f} % of frames)")
