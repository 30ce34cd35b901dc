#!/usr/bin/env python3
"""Generate the committed golden fixtures: small seeded bitstreams + SHA-256 of every output plane.

    python tests/golden/make_golden.py

Streams come from the seeded generator (mobiclipdecoder_amd/csrc/mobi_streamgen.cpp); expected planes
from the CPU oracle (oracle/mobi_oracle.c).  When the mechanical transliteration of the reference is
available (oracle/_ref/libmobi_csref.so, build container only) every frame is ALSO decoded with it and
must agree before the fixture is written; the manifest records that ("csref_checked").
The reference repository itself contains no vectors, fixtures or media."""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from mobiclipdecoder_amd import default_params, generate_clip  # noqa: E402
from mobiclipdecoder_amd.streamgen import BASE_SEED  # noqa: E402
from tests.oracle_binding import OracleDecoder  # noqa: E402

CASES = [
    # name, config letter, overrides
    ("mods_64x48_rich", "A", dict(width=64, height=48, n_frames=10, pm_intra=150, pm_deep=200, pm_multiref=300, qdelta_prob=300, table1_prob=500, escape_prob=100, mv_range=12)),
    ("moflex_64x48_rich_iint", "B", dict(width=64, height=48, n_frames=10, pm_intra=150, pm_deep=200, pm_multiref=300, qdelta_prob=300, table1_prob=500, escape_prob=100, iframe_interval=4, mv_range=12)),
    ("mods_256x192_A", "A", dict(n_frames=8)),                       # config A, all five reference slots in use
    ("mods_256x192_edge_wrap", "A", dict(n_frames=5, edge_mode=1, mv_range=40, seed_add=1)),   # Stride == Width: row wrap
    ("moflex_528x48_edge_pad", "B", dict(width=528, height=48, n_frames=6, edge_mode=1, mv_range=40, pm_intra=100)),  # reads of stride padding
    ("moflex_640x480_B", "B", dict(n_frames=3)),                     # config B
    ("moflex_848x480_C", "C", dict(n_frames=2)),                     # config C (stand-in for 854x480)
    ("mods_q12", "A", dict(width=64, height=48, n_frames=4, quantizer=12)),
    ("moflex_q52", "B", dict(width=64, height=48, n_frames=4, quantizer=52)),
]


def csref():
    so = os.path.join(ROOT, "oracle", "_ref", "libmobi_csref.so")
    if not os.path.exists(so):
        return None
    L = C.CDLL(so)
    L.csref_create.restype = C.c_void_p
    L.csref_create.argtypes = [C.c_uint, C.c_uint, C.c_int]
    L.csref_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.POINTER(C.c_int)]
    for n in ("csref_y", "csref_uv"):
        getattr(L, n).restype = C.POINTER(C.c_uint8)
        getattr(L, n).argtypes = [C.c_void_p, C.c_int]
    return L


def main():
    L = csref()
    manifest = {"generator": "tests/golden/make_golden.py", "csref_checked": L is not None, "cases": []}
    for i, (name, cfg, kw) in enumerate(CASES):
        kw = dict(kw)
        seed = BASE_SEED + 7000 + i + kw.pop("seed_add", 0)
        p = default_params(cfg, seed, **kw)
        data, fo = generate_clip(p)
        open(os.path.join(HERE, name + ".bin"), "wb").write(data.tobytes())
        o = OracleDecoder(p.width, p.height, p.version)
        h = L.csref_create(p.width, p.height, p.version) if L else None
        frames = []
        for f in range(p.n_frames):
            o.Data, o.Offset = data[: fo[f + 1]], int(fo[f])
            r = o.DecodeFrame()
            assert r is not None, (name, f, o.last_error)
            if L:
                buf = np.ascontiguousarray(data[: fo[f + 1]])
                off = C.c_int(int(fo[f]))
                assert L.csref_decode(h, buf.ctypes.data, buf.size, C.byref(off)) == 0 and off.value == o.Offset
                S = o.Stride
                assert np.array_equal(np.ctypeslib.as_array(L.csref_y(h, 0), (p.height, S)), r[0])
                assert np.array_equal(np.ctypeslib.as_array(L.csref_uv(h, 0), (p.height // 2, S)), r[1])
            frames.append({"y_sha256": hashlib.sha256(r[0].tobytes()).hexdigest(), "uv_sha256": hashlib.sha256(r[1].tobytes()).hexdigest(),
                           "offset_after": o.Offset, "quantizer": o.Quantizer})
        manifest["cases"].append({"name": name, "width": p.width, "height": p.height, "version": p.version, "stride": o.Stride,
                                  "frame_off": [int(x) for x in fo], "bytes": int(data.size), "frames": frames})
        print(name, data.size, "bytes", p.n_frames, "frames")
    json.dump(manifest, open(os.path.join(HERE, "golden.json"), "w"), indent=1)
    write_text_manifest(manifest)


def write_text_manifest(manifest):
    """the same manifest as plain lines, for tests/golden/verify/VerifyGolden.cs (a C# program against the unmodified reference: no JSON
    parser needed under mono)"""
    with open(os.path.join(HERE, "golden_manifest.txt"), "w") as f:
        f.write("# written by tests/golden/make_golden.py from golden.json; read by tests/golden/verify/VerifyGolden.cs\n")
        for c in manifest["cases"]:
            f.write(f"case {c['name']} {c['width']} {c['height']} {c['version']} {len(c['frames'])}\n")
            for i, fr in enumerate(c["frames"]):
                f.write(f"frame {c['frame_off'][i]} {c['frame_off'][i + 1]} {fr['y_sha256']} {fr['uv_sha256']} {fr['offset_after']} {fr['quantizer']}\n")


if __name__ == "__main__":
    if "--text-only" in sys.argv:  # golden_manifest.txt from the committed golden.json, nothing regenerated
        write_text_manifest(json.load(open(os.path.join(HERE, "golden.json"))))
    else:
        main()
