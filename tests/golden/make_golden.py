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

# r05: one stream per class of input that r01-r04 refused (MOBI_E_UNSUPPORTED) and that is decoded now -- streams the reference decodes
# only through what ReadDCTMatrix's stores do to `Internal` (MD.cs:3424-3429), or with values beyond the command list's old fields.
# Found by tools-style searches over bit-flipped generated streams (every frame decodes in the oracle) or written by hand.
_FUZZ = dict(n_frames=4, width=64, height=48, pm_intra=120, pm_deep=150, pm_multiref=250, qdelta_prob=250, escape_prob=80, table1_prob=400)


def _flipped(cfg, version, trial, flips):
    p = default_params(cfg, BASE_SEED + 20000 + trial, version=version, **_FUZZ)
    data, fo = generate_clip(p)
    data = data.copy()
    for pos, bit in flips:
        data[pos] ^= 1 << bit
    return p.width, p.height, p.version, data, [int(x) for x in fo]


def _low_quantiser():
    p = default_params("A", BASE_SEED + 3005, n_frames=4, width=64, height=48, quantizer=12, pm_intra=150, cbp_prob=500)
    data, fo = generate_clip(p)
    data = data.copy()
    w = int(data[0]) | (int(data[1]) << 8)  # the I-frame header's 6-bit quantiser, bits 12..7 of the first word (MD.cs:224-236)
    w = (w & ~(0x3F << 7)) | (5 << 7)
    data[0], data[1] = w & 0xFF, w >> 8
    return p.width, p.height, p.version, data, [int(x) for x in fo]


def _bits_to_words(bits):
    bits += "0" * ((-len(bits)) % 16) + "0" * 64
    out = bytearray()
    for i in range(0, len(bits), 16):
        w = int(bits[i:i + 16], 2)
        out += bytes((w & 0xFF, w >> 8))
    return np.frombuffer(bytes(out), dtype=np.uint8)


def _hand_made(kind, version):
    """32x32 pictures written bit by bit (tests/test_parse_fallback.py has the same frames): a 16x16 plane and two chroma planes with
    parameters beyond int16; motion vectors of +8192 half-pels ("sixteen rows down" in the reference's linear addressing), once as the single
    leaf of a macroblock and once inside a three-leaf tree"""
    from tests.test_parse_fallback import _DC_MB, _part_code, _se_code, _ue_code
    hdr = "1" + "0" + "0" + format(20, "06b")
    frames = []
    if kind == "wide_plane":
        frames.append(hdr + _DC_MB * 3 + "0" + _ue_code(0) + "010" + _se_code(-70000) + "010" + _se_code(65536) + _se_code(-65537))
    else:
        skip = _part_code(version, 0, 0) + _ue_code(0)
        far = _part_code(version, 0, 1) + _se_code(8192) + _se_code(0) + _ue_code(0)
        deep = (_part_code(version, 0, 8) + _part_code(version, 1, 9) + _part_code(version, 5, 1) + _se_code(8192) + _se_code(0) +
                _part_code(version, 5, 1) + _se_code(8190) + _se_code(1) + _part_code(version, 1, 1) + _se_code(8196) + _se_code(-3) + _ue_code(0))
        frames += [hdr + _DC_MB * 4, "0" + _se_code(0) + far + skip * 3, "0" + _se_code(0) + deep + skip * 3]
    chunks = [_bits_to_words(b) for b in frames]
    fo = [0]
    for c in chunks:
        fo.append(fo[-1] + int(c.size))
    return 32, 32, version, np.concatenate(chunks), fo


def _with_rejected_frames(cfg, version):
    """r06: frames the reference REJECTS (DecodeFrame() swallows the exception and returns null, MD.cs:325-328) are part of what a green run
    pins: a P-frame into an empty ring (its first copied leaf dereferences a null Y[ref], MD.cs:413), then the stream's own I / P / P, then a
    P-frame whose bytes behind the first word are noise, then I / P / P again.  References reach one frame back only (pm_multiref = 0) and an
    I-frame follows every rejected frame, so no accepted frame predicts from a picture the reference left half-written."""
    p = default_params(cfg, BASE_SEED + 26000 + version, version=version, n_frames=7, width=64, height=48, pm_intra=120, pm_deep=120, pm_multiref=0, iframe_interval=3,
                       qdelta_prob=300, mv_range=10)
    data, fo = generate_clip(p)
    pk = [np.array(data[fo[f]:fo[f + 1]], copy=True) for f in range(7)]
    assert (pk[0][1] & 0x80) and (pk[3][1] & 0x80) and not (pk[1][1] & 0x80)
    from tests.oracle_binding import OracleDecoder as _O
    for fill in (0xA5, 0xFF, 0x5A, 0x3C, 0xC3, 0x00):  # noise the reference throws on (checked with the oracle, which restates its exceptions)
        noise = np.array(pk[2], copy=True)
        noise[2:] = fill
        o = _O(p.width, p.height, p.version)
        for q in (pk[0], pk[1]):
            o.Data, o.Offset = q, 0
            assert o.DecodeFrame() is not None
        o.Data, o.Offset = noise, 0
        rejected = o.DecodeFrame() is None
        o.close()
        if rejected:
            break
    else:
        raise AssertionError("no noise pattern is rejected")
    seq = [pk[1], pk[0], pk[1], noise, pk[3], pk[4], pk[5]]
    off = [0]
    for q in seq:
        off.append(off[-1] + int(q.size))
    return p.width, p.height, p.version, np.concatenate(seq), off


RAW_CASES = [
    ("r05_walk_mods_64x48", lambda: _flipped("A", 1, 64, [(74, 4), (93, 7), (403, 0)])),              # a run past its block: literal frame
    ("r05_walk_moflex_64x48", lambda: _flipped("B", 2, 3, [(714, 0), (574, 4), (408, 6)])),
    ("r05_scratch_mods_64x48", lambda: _flipped("A", 1, 514, [(228, 4), (241, 3), (245, 5), (90, 5)])),  # ... that reads Internal[154..217]
    ("r05_scratch_moflex_64x48", lambda: _flipped("B", 2, 5049, [(149, 3), (349, 2), (132, 2)])),
    ("r05_lowq_mods_64x48", _low_quantiser),                                                          # ModsDS quantiser 5
    ("r05_wide_plane_mods_32x32", lambda: _hand_made("wide_plane", 1)),
    ("r05_far_mv_mods_32x32", lambda: _hand_made("far_mv", 1)),
    ("r05_far_mv_moflex_32x32", lambda: _hand_made("far_mv", 2)),
    ("r06_rejected_mods_64x48", lambda: _with_rejected_frames("A", 1)),                               # frames the reference returns null for
    ("r06_rejected_moflex_64x48", lambda: _with_rejected_frames("B", 2)),
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
    from tests import oracle_binding
    OL = oracle_binding.lib()
    OL.mobi_oracle_coverage.argtypes = [C.c_void_p, C.c_int]
    OL.mobi_oracle_coverage.restype = None
    todo = []
    for i, (name, cfg, kw) in enumerate(CASES):
        kw = dict(kw)
        seed = BASE_SEED + 7000 + i + kw.pop("seed_add", 0)
        p = default_params(cfg, seed, **kw)
        data, fo = generate_clip(p)
        todo.append((name, p.width, p.height, p.version, data, [int(x) for x in fo]))
    todo += [(name,) + tuple(make()) for name, make in RAW_CASES]
    for name, width, height, version, data, fo in todo:
        class P:  # (what the loop below needs of the generator's parameter block)
            pass
        p = P()
        p.width, p.height, p.version, p.n_frames = width, height, version, len(fo) - 1
        data = np.ascontiguousarray(data, dtype=np.uint8)
        open(os.path.join(HERE, name + ".bin"), "wb").write(data.tobytes())
        o = OracleDecoder(p.width, p.height, p.version)
        OL.mobi_oracle_coverage(None, 1)
        h = L.csref_create(p.width, p.height, p.version) if L else None
        frames = []
        for f in range(p.n_frames):
            o.Data, o.Offset = data[: fo[f + 1]], int(fo[f])
            r = o.DecodeFrame()
            rejected = r is None
            assert not rejected or name.startswith("r06_rejected"), (name, f, o.last_error)
            if rejected:  # what the reference leaves behind a swallowed exception: the PARTIAL picture (it pins where the throw happened), Offset, Quantizer
                r = (o.y(0), o.uv(0))
            if L:
                buf = np.ascontiguousarray(data[: fo[f + 1]])
                off = C.c_int(int(fo[f]))
                assert (L.csref_decode(h, buf.ctypes.data, buf.size, C.byref(off)) != 0) == rejected and off.value == o.Offset
                S = o.Stride
                assert np.array_equal(np.ctypeslib.as_array(L.csref_y(h, 0), (p.height, S)), r[0])
                assert np.array_equal(np.ctypeslib.as_array(L.csref_uv(h, 0), (p.height // 2, S)), r[1])
            frames.append({"y_sha256": hashlib.sha256(np.ascontiguousarray(r[0]).tobytes()).hexdigest(), "uv_sha256": hashlib.sha256(np.ascontiguousarray(r[1]).tobytes()).hexdigest(),
                           "offset_after": o.Offset, "quantizer": o.Quantizer})
            if rejected:
                frames[-1]["rejected"] = True
        cov = np.zeros(363, np.uint64)  # MOBI_COV_WORDS (oracle/mobi_oracle.h): what this fixture exercises = what a green VerifyGolden run pins
        OL.mobi_oracle_coverage(cov.ctypes.data, 1)
        manifest["cases"].append({"name": name, "width": p.width, "height": p.height, "version": p.version, "stride": o.Stride,
                                  "frame_off": [int(x) for x in fo], "bytes": int(data.size), "frames": frames, "covers": covers(cov)})
        print(name, data.size, "bytes", p.n_frames, "frames")
    json.dump(manifest, open(os.path.join(HERE, "golden.json"), "w"), indent=1)
    write_text_manifest(manifest)


def covers(cov):
    """the oracle's coverage counters of one fixture (oracle/mobi_oracle.h, MOBI_COV_*) as a short description"""
    part = sum(1 for k in range(320) if cov[k])
    lst = lambda base, n: ",".join(str(k) for k in range(n) if cov[base + k]) or "-"
    return (f"partition-codes={part} intra-modes={lst(320, 20)} planes(16/8/4)={lst(340, 3)} escapes(level/run/raw)={lst(343, 3)} vlc-tables={lst(346, 2)} "
            f"refs={lst(348, 5)} phases={lst(353, 4)} idct(1p8/3p8/16p8/64p8/1p4/16p4)={lst(357, 6)}")


def write_text_manifest(manifest):
    """the same manifest as plain lines, for tests/golden/verify/VerifyGolden.cs (a C# program against the unmodified reference: no JSON
    parser needed under mono)"""
    with open(os.path.join(HERE, "golden_manifest.txt"), "w") as f:
        f.write("# written by tests/golden/make_golden.py from golden.json; read by tests/golden/verify/VerifyGolden.cs\n")
        for c in manifest["cases"]:
            f.write(f"case {c['name']} {c['width']} {c['height']} {c['version']} {len(c['frames'])}\n")
            if c.get("covers"):
                f.write(f"covers {c['covers']}\n")
            for i, fr in enumerate(c["frames"]):
                # "reject": the reference returns null for this frame (MD.cs:325-328); the hashes are those of the partial picture it keeps
                f.write(f"{'reject' if fr.get('rejected') else 'frame'} {c['frame_off'][i]} {c['frame_off'][i + 1]} {fr['y_sha256']} {fr['uv_sha256']} {fr['offset_after']} {fr['quantizer']}\n")


def selftest():
    """Nothing regenerated: (1) golden_manifest.txt is golden.json line for line; (2) every fixture decodes in the oracle to the recorded
    hashes / Offset / Quantizer / rejections; (3) every `covers` line -- what a green run of tests/golden/verify pins -- is what the oracle's
    coverage counters (the ones tests/test_coverage.py reads) say the fixture exercises TODAY, so "pins X" is never stale.  Exit code 0 / 1."""
    import io
    from tests import oracle_binding
    g = json.load(open(os.path.join(HERE, "golden.json")))
    bad = 0
    want = io.StringIO()
    real_open = open

    class _Capture:  # write_text_manifest into memory
        def __enter__(self): return want
        def __exit__(self, *a): return False
    import builtins
    builtins_open = builtins.open
    try:
        builtins.open = lambda path, mode="r", *a, **k: _Capture() if str(path).endswith("golden_manifest.txt") and "w" in mode else builtins_open(path, mode, *a, **k)
        write_text_manifest(g)
    finally:
        builtins.open = builtins_open
    if want.getvalue() != real_open(os.path.join(HERE, "golden_manifest.txt")).read():
        print("golden_manifest.txt is not what golden.json says (python tests/golden/make_golden.py --text-only)")
        bad += 1
    OL = oracle_binding.lib()
    OL.mobi_oracle_coverage.argtypes = [C.c_void_p, C.c_int]
    OL.mobi_oracle_coverage.restype = None
    n_frames = n_rej = 0
    for c in g["cases"]:
        data = np.fromfile(os.path.join(HERE, c["name"] + ".bin"), dtype=np.uint8)
        o = OracleDecoder(c["width"], c["height"], c["version"])
        OL.mobi_oracle_coverage(None, 1)
        for f, exp in enumerate(c["frames"]):
            o.Data, o.Offset = data[: c["frame_off"][f + 1]], c["frame_off"][f]
            r = o.DecodeFrame()
            y, uv = (o.y(0), o.uv(0)) if r is None else r
            ok = ((r is None) == bool(exp.get("rejected")) and o.Offset == exp["offset_after"] and o.Quantizer == exp["quantizer"] and
                  hashlib.sha256(np.ascontiguousarray(y).tobytes()).hexdigest() == exp["y_sha256"] and hashlib.sha256(np.ascontiguousarray(uv).tobytes()).hexdigest() == exp["uv_sha256"])
            n_frames += 1
            n_rej += r is None
            if not ok:
                print(f"{c['name']} frame {f}: the oracle no longer gives what the fixture records")
                bad += 1
        cov = np.zeros(363, np.uint64)
        OL.mobi_oracle_coverage(cov.ctypes.data, 1)
        if covers(cov) != c.get("covers"):
            print(f"{c['name']}: 'covers' is stale:\n  recorded {c.get('covers')}\n  today    {covers(cov)}")
            bad += 1
        o.close()
    print(f"selftest: {len(g['cases'])} fixtures, {n_frames} frames ({n_rej} of them rejected by the reference's restatement), {bad} problems")
    return 1 if bad else 0


if __name__ == "__main__":
    if "--selftest" in sys.argv:
        sys.exit(selftest())
    if "--text-only" in sys.argv:  # golden_manifest.txt from the committed golden.json, nothing regenerated
        write_text_manifest(json.load(open(os.path.join(HERE, "golden.json"))))
    else:
        main()
