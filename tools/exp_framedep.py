"""What does the PARSE of a frame need from the frame before it?  (measured before frame-parallel lanes were built.)

  python tools/exp_framedep.py [damaged trials]        CPU only (tests/tools/mobi_lsparse_host.cpp: mobi_framedep_measure)

Every frame is parsed twice: in stream order, and from the state before it with everything the frame HEADERS do not determine poisoned: the
16 interior bytes of the intra-mode cache (MD.cs:1840-1859, 2785-2843) and the MV predictor Internal[219], [220] (MD.cs:207-208).  The header
fields (Quantizer and with it the dequant tables and the cache's border bytes, MD.cs:113-143, 224-236, 3884-3925; YuvFormat; the number of
frames in the ring) are a chain over the first bytes of each frame and are kept.  Counted: frames whose command list, rc and Offset are the
same, frames whose state afterwards is the same once the bytes the poisoned parse never wrote are taken from the state before -- for the
host parser and for the lock-step parser's lane functions."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mobiclipdecoder_amd import build, default_params, generate_clip
from mobiclipdecoder_amd.streamgen import BASE_SEED
from tests.gpu_streams import COVERAGE_SUITE

L = C.CDLL(build.build_lshost())
L.mobi_framedep_measure.argtypes = [C.c_uint, C.c_uint, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_long * 6)]


def measure(p, data, fo):
    fo = np.ascontiguousarray(fo, dtype=np.uint32)
    data = np.ascontiguousarray(data, dtype=np.uint8)
    st = (C.c_long * 6)()
    assert L.mobi_framedep_measure(p.width, p.height, p.version, data.ctypes.data, fo.ctypes.data, len(fo) - 1, C.byref(st)) == 0
    return np.array(list(st), dtype=np.int64)


def report(label, t):
    print(f"{label}: {t[0]} frames compared ({t[3]} more started from a host-only state): command list / rc / Offset identical from the poisoned "
          f"state: {t[1]} ({100.0 * t[1] / max(1, t[0]):.2f} %), state afterwards identical after the merge: {t[2]}; lock-step lane functions: "
          f"{t[4]} identical, {t[5]} bail-outs (not compared), {t[0] - t[4] - t[5]} different")
    return int(t[0] - t[1]) + int(t[0] - t[2]) + int(t[0] - t[4] - t[5])


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    bad = 0
    tot = np.zeros(6, np.int64)
    for cfg, seed, kw in COVERAGE_SUITE:
        p = default_params(cfg, BASE_SEED + seed, **kw)
        tot += measure(p, *generate_clip(p))
    bad += report("coverage suite (tests/gpu_streams.py, 12 streams)", tot)
    tot = np.zeros(6, np.int64)
    for i in range(16):  # the bench's streams: SURVEY 8(d) mix
        for cfg in "ABC":
            p = default_params(cfg, BASE_SEED + i, n_frames=12)
            tot += measure(p, *generate_clip(p))
    bad += report("SURVEY 8(d) mix, configs A / B / C, 16 seeds x 12 frames", tot)
    tot = np.zeros(6, np.int64)
    rng = np.random.default_rng(0x4652)
    for trial in range(64):  # random points of the generator's parameter space, I-frames in between, quantiser deltas
        w, h = [(64, 48), (128, 96), (256, 192), (320, 240), (512, 32), (16, 144)][trial % 6]
        kw = dict(width=w, height=h, version=1 + trial % 2, n_frames=int(rng.integers(4, 12)), quantizer=int(rng.integers(12, 53)),
                  pm_skip=int(rng.integers(0, 300)), pm_split1=int(rng.integers(0, 300)), pm_deep=int(rng.integers(0, 200)), pm_intra=int(rng.integers(0, 400)),
                  pm_multiref=int(rng.integers(0, 500)), mv_range=int(rng.integers(0, 40)), cbp_prob=int(rng.integers(0, 1000)), t8_prob=int(rng.integers(0, 1000)),
                  intra_sub_prob=int(rng.integers(0, 1000)), plane_prob=int(rng.integers(0, 700)), escape_prob=int(rng.integers(0, 400)),
                  qdelta_prob=int(rng.integers(0, 600)), table1_prob=int(rng.integers(0, 1000)), iframe_interval=int(rng.integers(0, 5)))
        p = default_params("A", BASE_SEED + 8000 + trial, **kw)
        tot += measure(p, *generate_clip(p))
    bad += report("64 random generator mixes", tot)
    tot = np.zeros(6, np.int64)
    for trial in range(trials):  # the fuzz corpus of tools/exp_refusals.py: 1..7 bit flips in a rich stream
        p = default_params("AB"[trial % 2], BASE_SEED + 9000 + trial, n_frames=4, width=96, height=64, version=1 + trial % 2, pm_intra=120, pm_deep=150,
                           pm_multiref=250, qdelta_prob=250, escape_prob=80, table1_prob=400)
        data, fo = generate_clip(p)
        d = data.copy()
        for _ in range(int(rng.integers(1, 8))):
            d[int(rng.integers(0, d.size))] ^= 1 << int(rng.integers(0, 8))
        tot += measure(p, d, fo)
    bad += report(f"{trials} damaged streams (1..7 bit flips)", tot)
    print("frames whose parse depends on more than the header chain:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
