#!/usr/bin/env python3
"""Writes tests/golden/clamp_fault_mods_64x48.bin: a stream whose ONLY fault is the clamp-table domain (MobiConst.cs:587).
Search (CPU only): flip three bits of a generated 64x48 ModsDS I-frame (quantizer 52, one coefficient per block) until the
oracle throws its index fault while the command-list interpreter (the product's parser + the kernels' arithmetic on the CPU,
tests/tools/mobi_cmd_interp.cpp) parses cleanly and reports MOBI_E_CLAMP.  Deterministic: seeded generator, seeded search."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mobiclipdecoder_amd import default_params, generate_clip, MobiclipVersion  # noqa: E402
from mobiclipdecoder_amd.streamgen import BASE_SEED  # noqa: E402
from tests.oracle_binding import OracleDecoder  # noqa: E402
from tests.interp_binding import InterpDecoder  # noqa: E402

p = default_params("A", BASE_SEED + 77, n_frames=2, width=64, height=48, quantizer=52, cbp_prob=1000, max_coefs=1, scan_span=1)
data, fo = generate_clip(p)
rng = np.random.default_rng(5)
for trial in range(400):
    d2 = data.copy()
    for _ in range(3):
        d2[int(rng.integers(4, fo[1]))] ^= 1 << int(rng.integers(0, 8))
    ora = OracleDecoder(64, 48, MobiclipVersion.ModsDS)
    ora.Data, ora.Offset = d2[: fo[1]], 0
    ora.DecodeFrame()
    if ora.last_error != -1:
        continue
    it = InterpDecoder(64, 48, MobiclipVersion.ModsDS)
    it.Data, it.Offset = d2[: fo[1]], 0
    it.DecodeFrame()
    if it.last_error == -5:
        d2[: fo[1]].tofile(os.path.join(ROOT, "tests", "golden", "clamp_fault_mods_64x48.bin"))
        print("trial", trial, "->", fo[1], "bytes")
        break
else:
    raise SystemExit("no clamp-only fault found")
