"""r05: the answer must not depend on which side parses.  The device parsers finish frames that decode without incident; every other frame
is parsed again by the host parser from the decoder state the clip had when the frame started (mobi_state.h, mobi_abi.cpp).  What that rests
on, checked here without a GPU:

  * the host parser's picture of Internal[] -- dequant words, the coefficient block, the transforms' scratch, the MV predictor and row
    cache -- equals the ORACLE's Internal[] (which follows the reference statement by statement) after every frame both decode;
  * what mobi_parse_tail rebuilds from a clean frame's command list (mobi_tail_scan_mb / mobi_tail_finish, the same functions on the CPU)
    equals the parser's own bookkeeping, frame after frame;
  * a parser seeded with that state takes a clip over in the middle and produces the same rc, Offset, command list and pictures as a parser
    that saw the clip from its first frame.
GPU: the same streams through the C ABI in all four parse modes."""
import ctypes as C

import numpy as np
import pytest

from mobiclipdecoder_amd import default_params, generate_clip
from mobiclipdecoder_amd.streamgen import BASE_SEED
from tests.interp_binding import InterpDecoder, lib as interp_lib
from tests.oracle_binding import OracleDecoder, lib as oracle_lib


class DevState(C.Structure):
    _fields_ = [("quant", C.c_uint32), ("yuvfmt", C.c_uint32), ("frames_started", C.c_int32), ("tables_set", C.c_uint32),
                ("mcache", C.c_uint8 * 40), ("predx", C.c_int32), ("predy", C.c_int32)]


class DevTail(C.Structure):
    _fields_ = [("ib", C.c_uint32 * 64), ("scratch", C.c_uint32 * 64), ("mvc", C.c_int32 * 132), ("pad", C.c_uint32 * 4)]


def _bind():
    L = interp_lib()
    L.mobi_cmdinterp_internal.restype = C.c_uint32
    L.mobi_cmdinterp_internal.argtypes = [C.c_void_p, C.c_uint32]
    L.mobi_cmdinterp_export_state.argtypes = [C.c_void_p, C.POINTER(DevState), C.POINTER(DevTail)]
    L.mobi_cmdinterp_import_state.argtypes = [C.c_void_p, C.POINTER(DevState), C.POINTER(DevTail), C.c_int]
    L.mobi_cmdinterp_copy_ring.argtypes = [C.c_void_p, C.c_void_p]
    L.mobi_cmdinterp_tail.argtypes = [C.c_void_p, C.POINTER(DevTail), C.POINTER(DevTail)]
    return L


def _rich(trial, version=None, **kw):
    ver = version if version is not None else 1 + trial % 2
    args = dict(n_frames=5, width=96, height=64, version=ver, pm_intra=120, pm_deep=150, pm_multiref=250, qdelta_prob=250, escape_prob=80,
                table1_prob=400)
    args.update(kw)
    return default_params("AB"[trial % 2], BASE_SEED + 9000 + trial, **args)


def _flipped(trial, rng, n_flips=None, **kw):
    p = _rich(trial, **kw)
    data, fo = generate_clip(p)
    data = data.copy()
    for _ in range(int(rng.integers(1, 8)) if n_flips is None else n_flips):
        data[int(rng.integers(0, data.size))] ^= 1 << int(rng.integers(0, 8))
    return p, data, fo


def test_internal_words_match_the_oracle():
    """Internal[10..] of the host parser against the oracle's after every frame both decode: clean streams and bit-flipped ones (walks
    through Internal[], transforms of every variant, I-frames behind P-frames)"""
    L = _bind()
    OL = oracle_lib()
    rng = np.random.default_rng(55)
    frames = walked = 0
    for trial in range(120):
        p, data, fo = _flipped(trial, rng, n_flips=0 if trial % 3 == 0 else None)
        a, o = InterpDecoder(p.width, p.height, p.version), OracleDecoder(p.width, p.height, p.version)
        n_words = 221 + 2 * (p.width // 16 + 2)
        for f in range(p.n_frames):
            a.Data = o.Data = data[: fo[f + 1]]
            a.Offset = o.Offset = int(fo[f])
            a.DecodeFrame(); o.DecodeFrame()
            if a.last_error != 0 or o.last_error != 0:
                break
            I = np.ctypeslib.as_array(OL.mobi_oracle_internal(o.h), (392,))
            mine = np.array([L.mobi_cmdinterp_internal(a.h, i) for i in range(10, n_words)], dtype=np.uint32)
            assert np.array_equal(mine, I[10:n_words]), (p.seed, f, np.nonzero(mine != I[10:n_words])[0][:8] + 10)
            frames += 1
        a.close(); o.close()
    assert frames > 250, frames


def test_tail_from_the_command_list_equals_the_parsers_bookkeeping():
    """mobi_parse_tail's arithmetic on the CPU: ib / scratch rebuilt from the command list of every clean frame, chained from frame to frame,
    against the parser's own Internal[90..217]"""
    L = _bind()
    frames = 0
    for trial in range(60):
        for content in ({}, dict(cbp_prob=150), dict(cbp_prob=700, pm_intra=300)):
            p = _rich(trial, **content)
            data, fo = generate_clip(p)
            a = InterpDecoder(p.width, p.height, p.version)
            tail = DevTail()
            for f in range(p.n_frames):
                a.Data = data[: fo[f + 1]]
                a.Offset = int(fo[f])
                a.DecodeFrame()
                assert a.last_error == 0
                out = DevTail()
                L.mobi_cmdinterp_tail(a.h, C.byref(tail), C.byref(out))
                st, mine = DevState(), DevTail()
                L.mobi_cmdinterp_export_state(a.h, C.byref(st), C.byref(mine))
                assert list(out.ib) == list(mine.ib), (p.seed, f, "ib")
                assert list(out.scratch) == list(mine.scratch), (p.seed, f, "scratch")
                tail = out
                frames += 1
            a.close()
    assert frames >= 800


def _takeover(L, p, data, fo, at):
    """frames [0, at) by decoder A ("the device": its state leaves as MobiDevState + a tail chained through mobi_cmdinterp_tail), frame `at`
    and the rest by a NEW decoder seeded with that state; beside them decoder R that sees everything.  Returns the per-frame results of both."""
    A, R = InterpDecoder(p.width, p.height, p.version), InterpDecoder(p.width, p.height, p.version)
    tail = DevTail()
    ra, rr = [], []
    for f in range(p.n_frames):
        R.Data = data[: fo[f + 1]]; R.Offset = int(fo[f])
        pr = R.DecodeFrame()
        rr.append((R.last_error, R.Offset, R.Quantizer, None if pr is None else (pr[0].copy(), pr[1].copy())))
    B = None
    for f in range(p.n_frames):
        if f == at:
            st, own = DevState(), DevTail()
            L.mobi_cmdinterp_export_state(A.h, C.byref(st), C.byref(own))
            own.ib[:] = tail.ib[:]        # what the device would hold: rebuilt from the command lists, not the parser's own words
            own.scratch[:] = tail.scratch[:]
            B = InterpDecoder(p.width, p.height, p.version)
            L.mobi_cmdinterp_import_state(B.h, C.byref(st), C.byref(own), f)
            L.mobi_cmdinterp_copy_ring(B.h, A.h)
        d = B if B is not None else A
        d.Data = data[: fo[f + 1]]; d.Offset = int(fo[f])
        pd = d.DecodeFrame()
        ra.append((d.last_error, d.Offset, d.Quantizer, None if pd is None else (pd[0].copy(), pd[1].copy())))
        if B is None:
            if d.last_error != 0:
                return None  # (frames before the take-over must be clean: the device would not have kept them)
            out = DevTail()
            L.mobi_cmdinterp_tail(A.h, C.byref(tail), C.byref(out))
            tail = out
    return ra, rr


def test_a_seeded_parser_takes_a_clip_over_in_the_middle():
    L = _bind()
    rng = np.random.default_rng(77)
    compared = errors = 0
    for trial in range(150):
        p = _rich(trial, n_frames=6)
        data, fo = generate_clip(p)
        data = data.copy()
        at = 1 + trial % 4
        # corrupt frames from `at` on only: the frames before it are the device's
        lo, hi = int(fo[at]), int(fo[p.n_frames])
        for _ in range(int(rng.integers(1, 8))):
            data[int(rng.integers(lo, hi))] ^= 1 << int(rng.integers(0, 8))
        r = _takeover(L, p, data, fo, at)
        assert r is not None
        ra, rr = r
        for f in range(p.n_frames):
            assert ra[f][:3] == rr[f][:3], (p.seed, at, f, ra[f][:3], rr[f][:3])
            if ra[f][0] != 0:
                errors += 1
                break
            assert np.array_equal(ra[f][3][0], rr[f][3][0]) and np.array_equal(ra[f][3][1], rr[f][3][1]), (p.seed, at, f)
            compared += 1
    assert compared > 500 and errors > 20, (compared, errors)


def test_stale_quantiser_tables():
    """SetupQuantizationTables assigns Quantizer before its table index can throw (MD.cs:3886-3890).  ModsDS, I-frames with
    q = 20, then 60 (throws: Quantizer = 60, tables of 20), then 60 again (no set-up: decodes with the tables of 20)"""
    from tests.test_internal_walk import _set_quantizer
    for q_bad in (54, 60, 63):
        p = default_params("A", BASE_SEED + 4242, n_frames=1, width=64, height=48, quantizer=20, pm_intra=1000, cbp_prob=600)
        data, fo = generate_clip(p)
        good = data[: fo[1]].copy()
        bad = good.copy()
        _set_quantizer(bad, q_bad)
        a, o = InterpDecoder(p.width, p.height, p.version), OracleDecoder(p.width, p.height, p.version)
        for k, frame in enumerate((good, bad, bad)):
            a.Data = o.Data = frame
            a.Offset = o.Offset = 0
            ra, ro = a.DecodeFrame(), o.DecodeFrame()
            assert (a.last_error == 0) == (o.last_error == 0), (q_bad, k, a.last_error, o.last_error)
            if k == 1:
                assert a.last_error != 0
            else:
                assert a.last_error == 0 and a.Quantizer == o.Quantizer
                assert np.array_equal(ra[0], ro[0]) and np.array_equal(ra[1], ro[1]), (q_bad, k)
        a.close(); o.close()


def _se_code(v):
    """Elias-gamma bits of the signed value v (MD.cs:2998-3015: odd codes map to non-positive values)"""
    u = 2 * v if v > 0 else 1 - 2 * v  # se: u even -> u >> 1, u odd -> (1 - u) >> 1
    z = u.bit_length() - 1
    return "0" * z + format(u, "b")


def _ue_code(v):
    u = v + 1
    z = u.bit_length() - 1
    return "0" * z + format(u, "b")


def _part_code(ver, shape, code):
    """bits of partition code `code` for block shape index `shape` (wi * 4 + hi), from the decoder's own look-up tables (Appendix B)"""
    from tests.tables import load
    T = load()
    v = 0 if ver == 2 else 1
    nbits, width = int(T["mobi_part_bits"][v][shape][code]), 32 - int(T["mobi_part_shift"][v][shape])
    for peek in range(1 << width):
        if int(T["mobi_part_lut"][v][shape][peek]) == code and (peek & ((1 << (width - nbits)) - 1)) == 0:
            return format(peek >> (width - nbits), "0%db" % nbits)
    raise AssertionError((ver, shape, code))


def _frame(bits):
    bits += "0" * ((-len(bits)) % 16) + "0" * 64
    out = bytearray()
    for i in range(0, len(bits), 16):
        w = int(bits[i:i + 16], 2)
        out += bytes((w & 0xFF, w >> 8))
    return np.frombuffer(bytes(out), dtype=np.uint8).copy()


def _decode_both(w, h, ver, frames):
    a, o = InterpDecoder(w, h, ver), OracleDecoder(w, h, ver)
    n_ok = 0
    for data in frames:
        a.Data = o.Data = data
        a.Offset = o.Offset = 0
        ra, ro = a.DecodeFrame(), o.DecodeFrame()
        assert a.last_error == o.last_error, (a.last_error, o.last_error)
        if o.last_error != 0:
            break
        assert a.Offset == o.Offset and np.array_equal(ra[0], ro[0]) and np.array_equal(ra[1], ro[1])
        n_ok += 1
    a.close(); o.close()
    return n_ok


_DC_MB = "0" + _ue_code(0) + "011" + "011"  # full intra macroblock, nothing coded, luma mode 3 (DC), chroma mode 3


def test_wide_plane_parameters_travel():
    """r01-r04 refused a plane parameter beyond int16 (a code of 33 bits and more).  A 32x32 ModsDS I-frame whose last macroblock -- the one
    with a row above and a column to its left -- is a 16x16 plane and two chroma planes with such parameters: the oracle's picture"""
    hdr = "1" + "0" + "0" + format(20, "06b")
    for py, pu, pv in ((40000, 3, -2), (-70000, 65536, -65537), (2 ** 20 + 5, -(2 ** 22), 12345678), (5, -40000, 7)):
        bits = hdr + _DC_MB * 3 + "0" + _ue_code(0) + "010" + _se_code(py) + "010" + _se_code(pu) + _se_code(pv)
        assert _decode_both(32, 32, 1, [_frame(bits)]) == 1, (py, pu, pv)


def test_far_motion_vectors_travel():
    """r01-r04 refused |MV| > 8191 half-pels.  The reference addresses linearly (MD.cs:400-416): on a 32x32 picture (Stride 256) the vector
    (+8192, 0) is "sixteen rows down".  Once as the single leaf of a macroblock, once in a three-leaf tree (the MV cell map's 14-bit fields)"""
    for ver in (1, 2):
        hdr = "1" + "0" + "0" + format(20, "06b")
        iframe = _frame(hdr + _DC_MB * 4)
        skip = _part_code(ver, 0, 0) + _ue_code(0)  # 16x16, predicted vector, nothing coded
        far = _part_code(ver, 0, 1) + _se_code(8192) + _se_code(0) + _ue_code(0)  # 16x16, ref 1, dx = +8192 half-pels
        # code 8 at 16x16 (two 16x8), the top one split again by code 9 (two 8x8): three leaves.  Shapes: 16x8 = wi 0, hi 1 -> 1; 8x8 -> 5.
        deep = (_part_code(ver, 0, 8) + _part_code(ver, 1, 9) + _part_code(ver, 5, 1) + _se_code(8192) + _se_code(0) +
                _part_code(ver, 5, 1) + _se_code(8190) + _se_code(1) + _part_code(ver, 1, 1) + _se_code(8196) + _se_code(-3) + _ue_code(0))
        for first in (far, deep):
            pframe = _frame("0" + _se_code(0) + first + skip * 3)
            assert _decode_both(32, 32, ver, [iframe, pframe]) == 2, ver


# ---- GPU: all four parse modes give the host parser's answer -------------------------------------------------------------------------
def _fuzz_streams(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for trial in range(n):
        p, data, fo = _flipped(trial, rng, n_frames=6)
        out.append((p, data, fo))
    return out


def test_the_payload_bound_the_device_arena_is_sized_by():
    """mobi_abi.cpp dp_parse sizes a clip's part of the payload arena as 64 words per macroblock + one level word per THREE bits of the frame
    (+ slack): every command list -- the device parsers' and the host parser's, which is copied over a clip's rows when it takes the clip
    over -- has to fit, whatever the stream.  Clean and corrupted frames, both versions; and the reason it holds: no table code with a level
    is shorter than three bits."""
    blob = (C.c_uint8 * 65536)()
    L = interp_lib()
    if hasattr(L, "mobi_cmdinterp_tables"):
        for ver in (1, 2):
            n = L.mobi_cmdinterp_tables(ver, blob)
            A = np.frombuffer(bytes(blob[:16384]), np.uint16)
            short = A[(A & 15) < 3]
            assert short.size and not np.any((short >> 4) & 31), "a table code with a level and fewer than three bits"
    rng = np.random.default_rng(77)
    worst = 0.0
    for trial in range(120):
        ver = 1 + trial % 2
        p = default_params("AB"[trial % 2], BASE_SEED + 31000 + trial, n_frames=3, width=96, height=64, version=ver, pm_intra=150, pm_deep=250,
                           escape_prob=120, table1_prob=400, qdelta_prob=200)
        data, fo = generate_clip(p)
        d = data.copy()
        if trial % 3:
            for _ in range(int(rng.integers(1, 6))):
                d[int(rng.integers(0, d.size))] ^= 1 << int(rng.integers(0, 8))
        a = InterpDecoder(p.width, p.height, p.version)
        n_mbs = (p.width // 16) * (p.height // 16)
        for f in range(p.n_frames):
            a.Data = d[: fo[f + 1]]
            a.Offset = int(fo[f])
            a.DecodeFrame()
            if a.last_error != 0:
                break
            words, bits = int(a.payload().size), 8 * int(fo[f + 1] - fo[f])
            bound = n_mbs * 64 + (bits + 2) // 3
            assert words <= bound, (trial, f, words, bound)
            worst = max(worst, words / bound)
        a.close()
    assert worst > 0.05  # (the check is not vacuous: real lists come within sight of the bound)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_gpu_every_parse_mode_gives_the_same_answer(mode):
    """bit-flipped streams as a BATCH per geometry / version through mobi_batch_decode in parse mode 0..3: rc, Offset, Quantizer and the
    planes of every frame equal the oracle's (the same assertion in every mode), and the device modes did hand clips to the host parser"""
    from mobiclipdecoder_amd import MobiclipBatch
    streams = _fuzz_streams(96, 909)
    total_same = total_rej = handed = 0
    for ver in (1, 2):
        group = [s for s in streams if s[0].version == ver]
        p0 = group[0][0]
        b = MobiclipBatch(len(group), p0.width, p0.height, ver, device_parse=mode)
        oras = [OracleDecoder(p0.width, p0.height, ver) for _ in group]
        alive = [True] * len(group)
        for f in range(p0.n_frames):
            datas = [g[1][: g[2][f + 1]] for g in group]
            offs = [int(g[2][f]) for g in group]
            rcs, offs_out = b.decode(datas, offs)
            for i, (p, data, fo) in enumerate(group):
                o = oras[i]
                o.Data = datas[i]; o.Offset = offs[i]
                ro = o.DecodeFrame()
                if not alive[i]:
                    continue
                assert rcs[i] != -6, (mode, p.seed, f)
                if rcs[i] == -5:
                    assert o.last_error == -1, (mode, p.seed, f, o.last_error)
                    alive[i] = False; total_rej += 1
                    continue
                assert (rcs[i] == 0) == (o.last_error == 0), (mode, p.seed, f, rcs[i], o.last_error)
                if rcs[i] != 0:
                    alive[i] = False; total_rej += 1
                    continue
                assert offs_out[i] == o.Offset and b.quantizer(i) == o.Quantizer, (mode, p.seed, f)
                y, uv = b.planes(i)
                assert np.array_equal(y, ro[0]) and np.array_equal(uv, ro[1]), (mode, p.seed, f)
                total_same += 1
        handed += b.host_clips()
        b.close()
        for o in oras:
            o.close()
    assert total_same > 200 and total_rej > 20, (total_same, total_rej)
    if mode == 0:
        assert handed == len(streams)
    else:
        assert 10 < handed < len(streams), handed


@pytest.mark.gpu
@pytest.mark.parametrize("lockstep", [False, True])
def test_gpu_asynchronous_steps_repair_what_the_device_cannot_finish(lockstep):
    """mobi_batch_submit / mobi_batch_wait with two steps in flight over bit-flipped streams: a frame the device parser cannot finish is found
    in wait, parsed by the host parser and reconstructed on its own -- and so is the frame of the step already in flight behind it"""
    from mobiclipdecoder_amd import MobiclipBatch
    streams = [s for s in _fuzz_streams(64, 1234) if s[0].version == 2]
    p0 = streams[0][0]
    b = MobiclipBatch(len(streams), p0.width, p0.height, 2, device_parse=3 if lockstep else 1)
    oras = [OracleDecoder(p0.width, p0.height, 2) for _ in streams]
    alive = [True] * len(streams)
    same = 0

    def check(f, rcs, offs_out, ring_idx):
        nonlocal same
        for i, (p, data, fo) in enumerate(streams):
            o = oras[i]
            o.Data = data[: fo[f + 1]]; o.Offset = int(fo[f])
            ro = o.DecodeFrame()
            if not alive[i]:
                continue
            assert rcs[i] != -6
            if rcs[i] == -5:
                assert o.last_error == -1
                alive[i] = False
                continue
            assert (rcs[i] == 0) == (o.last_error == 0), (p.seed, f, rcs[i], o.last_error)
            if rcs[i] != 0:
                alive[i] = False
                continue
            assert offs_out[i] == o.Offset, (p.seed, f)
            y, uv = b.planes(i, ring_idx)
            assert np.array_equal(y, ro[0]) and np.array_equal(uv, ro[1]), (p.seed, f)
            same += 1

    nf = p0.n_frames
    b.submit([s[1][: s[2][1]] for s in streams], [int(s[2][0]) for s in streams])
    for f in range(1, nf):
        b.submit([s[1][: s[2][f + 1]] for s in streams], [int(s[2][f]) for s in streams])
        rcs, offs_out = b.wait()
        check(f - 1, rcs, offs_out, 1)
    rcs, offs_out = b.wait()
    check(nf - 1, rcs, offs_out, 0)
    assert same > 40 and 3 < b.host_clips() < len(streams), (same, b.host_clips())
    b.close()
    for o in oras:
        o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2, 3])
def test_gpu_every_clip_of_a_batch_handed_over_at_once(mode):
    """every clip's second frame is damaged beyond what the device parsers finish (a ModsDS I-frame header re-written to quantiser 5: every
    residual block then walks through Internal[]): the whole batch goes to the host parser within that call, and stays decodable"""
    from mobiclipdecoder_amd import MobiclipBatch
    from tests.test_internal_walk import _set_quantizer
    n = 12
    ps = [default_params("A", BASE_SEED + 3005 + 100 * (i % 3), n_frames=4, width=64, height=48, quantizer=12, pm_intra=150, cbp_prob=500, iframe_interval=2) for i in range(n)]
    clips = []
    for p in ps:
        data, fo = generate_clip(p)
        data = data.copy()
        frame2 = data[fo[2]:fo[3]]      # iframe_interval=2: frame 2 is an I-frame
        assert frame2[1] & 0x80
        _set_quantizer(frame2, 5)
        clips.append((data, fo))
    b = MobiclipBatch(n, 64, 48, 1, device_parse=mode)
    oras = [OracleDecoder(64, 48, 1) for _ in range(n)]
    for f in range(4):
        datas = [c[0][c[1][f]:c[1][f + 1]] for c in clips]
        rcs, offs = b.decode(datas, [0] * n)
        if f == 1:
            assert b.host_clips() < n
        if f == 2:
            assert b.host_clips() == n  # (hybrid: its share was there already; the others came with this frame)
        for i in range(n):
            oras[i].Data, oras[i].Offset = datas[i], 0
            ro = oras[i].DecodeFrame()
            assert rcs[i] == oras[i].last_error == 0 and offs[i] == oras[i].Offset and b.quantizer(i) == oras[i].Quantizer, (mode, f, i, rcs[i], oras[i].last_error)
            y, uv = b.planes(i)
            assert np.array_equal(y, ro[0]) and np.array_equal(uv, ro[1]), (mode, f, i)
    b.close()
    for o in oras:
        o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [64, 2048])
def test_gpu_every_clip_of_an_asynchronous_batch_handed_over_at_once(n):
    """the same in asynchronous steps, two in flight: mobi_batch_wait learns that NO clip's frame 2 was the device parsers' to finish, and that
    frame 3 of every clip -- already parsed from a wrong state and reconstructed -- has to be done again as well.  r06: the repair is a batch
    operation (states in one go, parses on the pool, one reconstruction of the failed clips per affected step): 2048 clips well inside a
    second, where r05's clip-by-clip repair took ~1.5 ms x 2 per clip"""
    import time
    from mobiclipdecoder_amd import MobiclipBatch
    from tests.test_internal_walk import _set_quantizer
    nfr = 6
    ps = [default_params("A", BASE_SEED + 3005 + 100 * i, n_frames=nfr, width=64, height=48, quantizer=12, pm_intra=150, cbp_prob=500, iframe_interval=2) for i in range(3)]
    src = []
    for p in ps:
        data, fo = generate_clip(p)
        data = data.copy()
        frame2 = data[fo[2]:fo[3]]
        assert frame2[1] & 0x80
        _set_quantizer(frame2, 5)
        src.append((data, fo))
    b = MobiclipBatch(n, 64, 48, 1, device_parse=True)
    oras = [OracleDecoder(64, 48, 1) for _ in range(3)]
    want = []
    for f in range(nfr):
        row = []
        for i in range(3):
            oras[i].Data, oras[i].Offset = src[i][0][src[i][1][f]:src[i][1][f + 1]], 0
            ro = oras[i].DecodeFrame()
            assert oras[i].last_error == 0
            row.append((oras[i].Offset, oras[i].Quantizer, ro[0].copy(), ro[1].copy()))
        want.append(row)
    frames = [[src[i % 3][0][src[i % 3][1][f]:src[i % 3][1][f + 1]] for i in range(n)] for f in range(nfr)]
    waits = []
    b.submit(frames[0], [0] * n)
    for f in range(1, nfr):
        b.submit(frames[f], [0] * n)
        t0 = time.perf_counter()
        rcs, offs = b.wait()  # reports frame f - 1
        waits.append((time.perf_counter() - t0) * 1e3)
        assert not any(rcs) and all(offs[i] == want[f - 1][i % 3][0] for i in range(n)), (f - 1, [r for r in rcs if r][:4])
    rcs, offs = b.wait()
    assert not any(rcs) and all(offs[i] == want[nfr - 1][i % 3][0] for i in range(n))
    assert b.host_clips() == n
    for i in list(range(6)) + [n - 1]:
        y, uv = b.planes(i)
        assert b.quantizer(i) == want[nfr - 1][i % 3][1]
        assert np.array_equal(y, want[nfr - 1][i % 3][2]) and np.array_equal(uv, want[nfr - 1][i % 3][3]), i
    assert b.compare_clips(3) == 0  # every other clip against its source clip, on the device
    print(f"asynchronous hand-over of {n} clips at once: the wait that repairs frames 2 and 3 took {waits[2]:.1f} ms (the other waits: {[round(w, 1) for w in waits[:2] + waits[3:]]})")
    assert waits[2] < 1000.0, waits
    b.close()
    for o in oras:
        o.close()


@pytest.mark.gpu
def test_gpu_hybrid_mode_in_asynchronous_steps():
    """r05: the hybrid mode's host share is parsed inside mobi_batch_submit, its command lists ride behind the parse kernels"""
    from mobiclipdecoder_amd import MobiclipBatch
    n, nfr = 10, 6
    ps = [default_params("A", BASE_SEED + 5100 + i, n_frames=nfr, pm_intra=120, iframe_interval=4) for i in range(n)]
    clips = [generate_clip(p) for p in ps]
    b = MobiclipBatch(n, 256, 192, 1, device_parse="hybrid")
    oras = [OracleDecoder(256, 192, 1) for _ in range(n)]

    def check(f, rcs, offs, ring_idx):
        for i in range(n):
            oras[i].Data, oras[i].Offset = clips[i][0][clips[i][1][f]:clips[i][1][f + 1]], 0
            ro = oras[i].DecodeFrame()
            assert rcs[i] == 0 and offs[i] == oras[i].Offset, (f, i)
            y, uv = b.planes(i, ring_idx)
            assert np.array_equal(y, ro[0]) and np.array_equal(uv, ro[1]), (f, i)

    frames = [[c[0][c[1][f]:c[1][f + 1]] for c in clips] for f in range(nfr)]
    b.submit(frames[0], [0] * n)
    for f in range(1, nfr):
        b.submit(frames[f], [0] * n)
        rcs, offs = b.wait()
        check(f - 1, rcs, offs, 1)
    rcs, offs = b.wait()
    check(nfr - 1, rcs, offs, 0)
    assert b.host_clips() == 2  # a fifth of ten
    b.close()
    for o in oras:
        o.close()


def _glitch_then_clean(seed, n_clean):
    """a clip whose first four frames are the golden walk stream (one of them a run through Internal[]: the host parser's) followed by the
    P-frames of a clean generated stream of the same geometry -- the syntax of a frame does not depend on the pictures before it"""
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    import json
    case = [c for c in json.load(open(os.path.join(here, "golden.json")))["cases"] if c["name"] == "r05_walk_moflex_64x48"][0]
    x = np.fromfile(os.path.join(here, case["name"] + ".bin"), dtype=np.uint8)
    frames = [x[case["frame_off"][f]:case["frame_off"][f + 1]] for f in range(len(case["frames"]))]
    p = default_params("B", seed, n_frames=n_clean + 1, width=64, height=48, version=2, pm_intra=80, pm_deep=100)
    y, fo = generate_clip(p)
    frames += [y[fo[f]:fo[f + 1]] for f in range(1, n_clean + 1)]
    return frames


@pytest.mark.gpu
@pytest.mark.parametrize("mode,asynchronous", [(1, False), (3, False), (1, True)])
def test_gpu_a_clip_goes_back_to_the_device_parser(mode, asynchronous):
    """a glitch, then clean frames: the host parser takes the clip over at the glitch and hands it back -- with its state -- after a run of
    frames the device parsers finish; every frame of every clip equals the oracle's on the way there and back"""
    from mobiclipdecoder_amd import MobiclipBatch
    n_clean = 22
    clips = [_glitch_then_clean(BASE_SEED + 7700 + i, n_clean) for i in range(3)]
    for i in range(2):  # and two clips that never leave the device parsers
        p = default_params("B", BASE_SEED + 7750 + i, n_frames=4 + n_clean, width=64, height=48, version=2, pm_intra=80)
        y, fo = generate_clip(p)
        clips.append([y[fo[f]:fo[f + 1]] for f in range(4 + n_clean)])
    n, nfr = len(clips), 4 + n_clean
    b = MobiclipBatch(n, 64, 48, 2, device_parse=mode)
    oras = [OracleDecoder(64, 48, 2) for _ in range(n)]
    on_host = []

    def check(f, rcs, offs, ring_idx):
        for i in range(n):
            oras[i].Data, oras[i].Offset = clips[i][f], 0
            ro = oras[i].DecodeFrame()
            assert rcs[i] == oras[i].last_error == 0 and offs[i] == oras[i].Offset, (mode, f, i, rcs[i], oras[i].last_error)
            y, uv = b.planes(i, ring_idx)
            assert np.array_equal(y, ro[0]) and np.array_equal(uv, ro[1]), (mode, f, i)

    if not asynchronous:
        for f in range(nfr):
            rcs, offs = b.decode([c[f] for c in clips], [0] * n)
            check(f, rcs, offs, 0)
            on_host.append(b.host_clips())
    else:
        b.submit([c[0] for c in clips], [0] * n)
        for f in range(1, nfr):
            b.submit([c[f] for c in clips], [0] * n)
            rcs, offs = b.wait()
            check(f - 1, rcs, offs, 1)
            on_host.append(b.host_clips())
        rcs, offs = b.wait()
        check(nfr - 1, rcs, offs, 0)
        on_host.append(b.host_clips())
    assert max(on_host) == 3 and on_host[-1] == 0, on_host  # handed over at the glitch, back after eight clean frames
    b.close()
    for o in oras:
        o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 3])
def test_gpu_long_streams_with_sparse_glitches(mode):
    """48 clips x 40 frames, one or two damaged frames per clip somewhere: clips leave for the host parser at a glitch and come back after a
    clean run, some of them twice -- and every frame's rc, Offset, Quantizer and planes equal those of a batch parsed on the host throughout"""
    from mobiclipdecoder_amd import MobiclipBatch
    n, nfr = 48, 40
    rng = np.random.default_rng(4040 + mode)
    clips = []
    for i in range(n):
        p = default_params("AB"[mode % 2], BASE_SEED + 8800 + i, n_frames=nfr, width=64, height=48, version=2, pm_intra=100, pm_deep=120, pm_multiref=200,
                           qdelta_prob=150, escape_prob=60, table1_prob=300, iframe_interval=13)
        data, fo = generate_clip(p)
        data = data.copy()
        if i % 6:  # (every sixth clip stays intact)
            for f in rng.choice(np.arange(2, nfr - 10), size=int(rng.integers(1, 3)), replace=False):
                for _ in range(int(rng.integers(1, 4))):
                    data[int(rng.integers(fo[f], fo[f + 1]))] ^= 1 << int(rng.integers(0, 8))
        clips.append((data, fo))
    db, hb = MobiclipBatch(n, 64, 48, 2, device_parse=mode), MobiclipBatch(n, 64, 48, 2, device_parse=0)
    seen_host, decoded = [], 0
    for f in range(nfr):
        datas = [c[0][c[1][f]:c[1][f + 1]] for c in clips]
        r1, o1 = db.decode(datas, [0] * n)
        r2, o2 = hb.decode(datas, [0] * n)
        assert r1 == r2 and o1 == o2, (mode, f, [(i, a, b) for i, (a, b) in enumerate(zip(r1, r2)) if a != b])
        for i in range(n):
            assert db.quantizer(i) == hb.quantizer(i), (mode, f, i)
            if r1[i] == 0:
                ya, ua = db.planes(i)
                yb, ub = hb.planes(i)
                assert np.array_equal(ya, yb) and np.array_equal(ua, ub), (mode, f, i)
                decoded += 1
        seen_host.append(db.host_clips())
    assert max(seen_host) >= 8 and seen_host[-1] < max(seen_host) and decoded > n * nfr * 0.8, (seen_host, decoded)
    db.close()
    hb.close()
