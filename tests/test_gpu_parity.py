"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit-exact Y/U/V planes,
post-call Offset and Quantizer, on seeded synthetic streams."""
import numpy as np
import pytest

from mobiclipdecoder_amd import MobiclipBatch, MobiclipDecoder, MobiclipVersion, default_params, generate_clip
from mobiclipdecoder_amd.streamgen import BASE_SEED
from tests.oracle_binding import OracleDecoder

pytestmark = pytest.mark.gpu


def _run_stream(params, whole_file=False):
    data, fo = generate_clip(params)
    gpu = MobiclipDecoder(params.width, params.height, params.version)
    ora = OracleDecoder(params.width, params.height, params.version)
    assert gpu.Stride == ora.Stride
    for f in range(params.n_frames):
        if whole_file:  # MOC5 style: Data = whole file, Offset = frame start (Form1.cs:292-302)
            gpu.Data = ora.Data = data
        else:           # Moflex/Mods style: one buffer per frame, Offset = 0 (Program.cs:69-71)
            gpu.Data = ora.Data = data[fo[f]:fo[f + 1]]
        gpu.Offset = ora.Offset = int(fo[f]) if whole_file else 0
        g = gpu.DecodeFrame()
        o = ora.DecodeFrame()
        assert gpu.last_error == ora.last_error == 0, (f, gpu.last_error, ora.last_error)
        assert gpu.Offset == ora.Offset, f
        assert gpu.Quantizer == ora.Quantizer, f
        assert np.array_equal(g[0], o[0]), f"Y mismatch frame {f}: {np.argwhere(g[0] != o[0])[:4].tolist()}"
        assert np.array_equal(g[1], o[1]), f"UV mismatch frame {f}: {np.argwhere(g[1] != o[1])[:4].tolist()}"
    # the whole ring, not just slot 0
    for r in range(6):
        gy, oy = gpu.Y[r], ora.y(r)
        assert (gy is None) == (oy is None)
        if gy is not None:
            assert np.array_equal(gy, oy) and np.array_equal(gpu.UV[r], ora.uv(r))
    gpu.close()


@pytest.mark.parametrize("cfg,seed", [("A", 0), ("A", 1), ("B", 0), ("B", 1), ("C", 0)])
def test_default_streams_bit_exact(cfg, seed):
    _run_stream(default_params(cfg, BASE_SEED + seed, n_frames=9))


def test_coverage_suite_streams():
    """Every stream of tests/gpu_streams.py (whose union tests/test_coverage.py proves to reach the whole syntax: every partition
    shape x table version x code, every intra mode in both sizes, the plane predictors, the escapes, both VLC tables, reference
    slots 1..5, the four CopyBlock phases and the six transform classes), bit-exact against the oracle."""
    from tests.gpu_streams import suite_params
    for p in suite_params():
        _run_stream(p)


@pytest.mark.parametrize("cfg", ["A", "B"])
def test_rich_streams_bit_exact(cfg):
    p = default_params(cfg, BASE_SEED + 77, n_frames=10, pm_intra=150, pm_deep=150, pm_multiref=300,
                       qdelta_prob=300, table1_prob=500, escape_prob=100, iframe_interval=6)
    _run_stream(p, whole_file=True)


def test_edge_mvs_pad_and_wrap():
    _run_stream(default_params("A", BASE_SEED + 5, n_frames=8, edge_mode=1, mv_range=40))
    _run_stream(default_params("B", BASE_SEED + 6, n_frames=8, edge_mode=1, mv_range=40))


def test_batch_and_replay_match_single():
    nclips, nfr = 5, 7
    ps = [default_params("A", BASE_SEED + 100 + i, n_frames=nfr, pm_intra=100) for i in range(nclips)]
    clips = [generate_clip(p) for p in ps]
    oras = [OracleDecoder(256, 192, MobiclipVersion.ModsDS) for _ in range(nclips)]
    b = MobiclipBatch(nclips, 256, 192, MobiclipVersion.ModsDS)
    for f in range(nfr):
        rcs, offs = b.decode([c[0] for c in clips], [int(c[1][f]) for c in clips])
        for i in range(nclips):
            oras[i].Data, oras[i].Offset = clips[i][0], int(clips[i][1][f])
            o = oras[i].DecodeFrame()
            assert rcs[i] == 0 and offs[i] == oras[i].Offset
            y, uv = b.planes(i)
            assert np.array_equal(y, o[0]) and np.array_equal(uv, o[1]), (f, i)
    b.close()
    # replay path: pre-parsed command lists resident in HBM, cloned clip included
    b = MobiclipBatch(nclips + 1, 256, 192, MobiclipVersion.ModsDS)
    for i in range(nclips):
        assert all(r == 0 for r in b.preload(i, clips[i][0], clips[i][1]))
    b.preload_clone(nclips, 2)
    b.commit()
    for f in range(nfr):
        b.replay(f)
    assert b.sync() == 0
    for i in range(nclips + 1):
        src = i if i < nclips else 2
        y, uv = b.planes(i)
        assert np.array_equal(y, oras[src].y(0)) and np.array_equal(uv, oras[src].uv(0)), i
    assert b.cmd_bytes(1) > 0
    b.close()


def test_error_codes_match_oracle_class():
    """Streams the reference would throw on: null reference slot (P-frame first), truncated data."""
    p = default_params("A", BASE_SEED + 9, n_frames=3)
    data, fo = generate_clip(p)
    gpu = MobiclipDecoder(256, 192, MobiclipVersion.ModsDS)
    ora = OracleDecoder(256, 192, MobiclipVersion.ModsDS)
    # P-frame with nothing in the ring -> NullReference in CopyBlock
    gpu.Data = ora.Data = data[fo[1]:fo[2]]
    gpu.Offset = ora.Offset = 0
    assert gpu.DecodeFrame() is None and ora.DecodeFrame() is None
    assert gpu.last_error == ora.last_error == -2
    # a good I-frame afterwards still decodes identically (ring rotated on both sides)
    gpu.Data = ora.Data = data[fo[0]:fo[1]]
    gpu.Offset = ora.Offset = 0
    g, o = gpu.DecodeFrame(), ora.DecodeFrame()
    assert np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1])
    gpu.close()


def test_full_size_batch_640x480_bit_exact_and_replay_is_deterministic():
    """BASELINE config "batch of independent 640x480 Moflex clips": every clip of a 24-clip batch, every frame,
    bit-exact against the oracle at full size; then the replay path twice -> identical planes (idempotent
    given the same ring history), and a cloned clip equals its source."""
    nclips, nfr = 24, 9
    ps = [default_params("B", BASE_SEED + 400 + i, n_frames=nfr) for i in range(nclips)]
    clips = [generate_clip(p) for p in ps]
    oras = [OracleDecoder(640, 480, MobiclipVersion.Moflex3DS) for _ in range(nclips)]
    b = MobiclipBatch(nclips, 640, 480, MobiclipVersion.Moflex3DS)
    for f in range(nfr):
        rcs, offs = b.decode([c[0][: c[1][f + 1]] for c in clips], [int(c[1][f]) for c in clips])
        assert all(r == 0 for r in rcs)
        for i in range(nclips):
            oras[i].Data, oras[i].Offset = clips[i][0][: clips[i][1][f + 1]], int(clips[i][1][f])
            o = oras[i].DecodeFrame()
            assert offs[i] == oras[i].Offset and b.quantizer(i) == oras[i].Quantizer
            if f in (0, 1, nfr - 1) or i % 5 == 0:
                y, uv = b.planes(i)
                assert np.array_equal(y, o[0]) and np.array_equal(uv, o[1]), (f, i)
    b.close()

    def replay_all():
        bb = MobiclipBatch(nclips + 1, 640, 480, MobiclipVersion.Moflex3DS)
        for i in range(nclips):
            assert all(r == 0 for r in bb.preload(i, clips[i][0], clips[i][1]))
        bb.preload_clone(nclips, 3)
        bb.commit()
        for f in range(nfr):
            bb.replay(f)
        assert bb.sync() == 0
        out = [bb.planes(i) for i in range(nclips + 1)]
        bb.close()
        return out
    a, c = replay_all(), replay_all()
    for i in range(nclips + 1):
        src = i if i < nclips else 3
        assert np.array_equal(a[i][0], c[i][0]) and np.array_equal(a[i][1], c[i][1])
        assert np.array_equal(a[i][0], oras[src].y(0)) and np.array_equal(a[i][1], oras[src].uv(0)), i


def test_848x480_wii_class_and_854_is_rejected():
    """Config "854x480 Wii MOC5": the reference cannot decode widths that are not a multiple of 16
    (MD.cs:216-217 row-pointer drift -> exception), so 848x480 stands in and 854 is refused."""
    _run_stream(default_params("C", BASE_SEED + 31, n_frames=5, pm_intra=80), whole_file=True)
    with pytest.raises(Exception):
        MobiclipDecoder(854, 480, MobiclipVersion.Moflex3DS)


def test_clamp_domain_fault_is_reported():
    """A residual that pushes pred+res outside the clamp table's domain makes the reference throw
    (MobiConst.cs:587; MD.cs:3551); the kernels flag it -> MOBI_E_CLAMP, and the oracle throws too.
    The stream is a committed fixture (tests/golden/clamp_fault_mods_64x48.bin: a 64x48 ModsDS I-frame at quantizer 52 with
    three flipped bits, found by tests/golden/make_clamp_fault.py): its syntax parses, only the clamp domain fails."""
    import os
    data = np.fromfile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clamp_fault_mods_64x48.bin"), dtype=np.uint8)
    ora = OracleDecoder(64, 48, MobiclipVersion.ModsDS)
    ora.Data, ora.Offset = data, 0
    assert ora.DecodeFrame() is None and ora.last_error == -1  # the reference's IndexOutOfRangeException
    gpu = MobiclipDecoder(64, 48, MobiclipVersion.ModsDS)
    gpu.Data, gpu.Offset = data, 0
    assert gpu.DecodeFrame() is None
    assert gpu.last_error == -5  # MOBI_E_CLAMP: found by the kernels, after a clean parse
    # Both values are fixed properties of this fixture: the reference (oracle) throws inside the macroblock whose residual leaves the
    # table, with 44 bytes read; the product's parse does not touch pixels, runs to the end of the frame (78 of the file's 122 bytes) and the
    # fault is found by the kernels afterwards ("Offset after an error"); the host-side command-list interpreter
    # (tests/test_interp_parity.py) pins the same 78 without a GPU.
    assert ora.Offset == 44
    assert gpu.Offset == 78
    gpu.close()


def test_intra_heavy_steps_are_bit_exact():
    """A frame step = the inter launch + ONE intra launch whose waves wait on per-macroblock completion tags.  It must
    reproduce the oracle, I-frames (deep dependency chains) and intra-heavy P-frames included, also in a batch."""
    mode = "1"
    _run_stream(default_params("B", BASE_SEED + 31, n_frames=8, pm_intra=300, iframe_interval=4))
    _run_stream(default_params("A", BASE_SEED + 32, n_frames=8, pm_intra=200, edge_mode=1, mv_range=40))
    nclips, nfr = 11, 6  # more clips than XCDs: the one-launch layout walks several clips per XCD
    ps = [default_params("A", BASE_SEED + 300 + i, n_frames=nfr, pm_intra=250) for i in range(nclips)]
    clips = [generate_clip(p) for p in ps]
    b = MobiclipBatch(nclips, ps[0].width, ps[0].height, ps[0].version)
    oras = [OracleDecoder(p.width, p.height, p.version) for p in ps]
    for f in range(nfr):
        rcs, offs = b.decode([c[0] for c in clips], [int(c[1][f]) for c in clips])
        for c in range(nclips):
            oras[c].Data, oras[c].Offset = clips[c][0], int(clips[c][1][f])
            o = oras[c].DecodeFrame()
            assert rcs[c] == 0 and offs[c] == oras[c].Offset
            y, uv = b.planes(c)
            assert np.array_equal(y, o[0]) and np.array_equal(uv, o[1]), (mode, f, c)
    b.close()


def test_loaded_gpu_hand_offs_are_exact():
    """The completion-tag hand-off between waves (intra macroblocks waiting on other macroblocks of the same launch)
    must hold on a LOADED chip, not just for a handful of clips: 320 clips of 640x480 (every CU busy, waves of many
    clips interleaved), intra-heavy, 7 frames incl. the I-frame's ~100-level wavefront, replayed twice.  Every clip
    is a private copy of one of 8 streams, so every copy must equal the oracle's frame, word for word, every frame."""
    mode = "1"
    nclips, distinct, nfr = 320, 8, 7
    ps = [default_params("B", BASE_SEED + 700 + i, n_frames=nfr, pm_intra=200, pm_deep=100) for i in range(distinct)]
    clips = [generate_clip(p) for p in ps]
    oras = [OracleDecoder(640, 480, MobiclipVersion.Moflex3DS) for _ in range(distinct)]
    b = MobiclipBatch(nclips, 640, 480, MobiclipVersion.Moflex3DS)
    for i in range(distinct):
        assert all(r == 0 for r in b.preload(i, clips[i][0], clips[i][1]))
    for c in range(distinct, nclips):
        b.preload_clone(c, c % distinct)
    b.commit()
    for rep in range(2):
        for f in range(nfr):
            b.replay(f)
            if rep == 0:
                for i in range(distinct):
                    oras[i].Data, oras[i].Offset = clips[i][0], int(clips[i][1][f])
                    assert oras[i].DecodeFrame() is not None
            if f in (0, 3, nfr - 1):
                assert b.sync() == 0
                want = None
                for c in range(nclips):
                    if rep == 0:
                        want = (oras[c % distinct].y(0), oras[c % distinct].uv(0))
                    y, uv = b.planes(c)
                    if rep == 0:
                        assert np.array_equal(y, want[0]) and np.array_equal(uv, want[1]), (mode, f, c)
                    elif f == nfr - 1:  # second pass: same command lists on a different ring history -> only P-chain end state is comparable
                        assert np.array_equal(y, oras[c % distinct].y(0)) and np.array_equal(uv, oras[c % distinct].uv(0)), (mode, "rep", c)
    b.close()


def test_octet_inter_kernel_on_every_kind_of_inter_macroblock():
    """mobi_recon_inter8 takes eight macroblocks per wave.  It must be bit-exact on a mix with every kind of inter
    macroblock (single, two halves top/bottom and left/right, deep trees, multi-reference, residual 8x8 and 4x4) and odd
    widths in macroblocks (848 = 53: the last octet of a row is partial)."""
    _run_stream(default_params("C", BASE_SEED + 61, n_frames=6, pm_deep=150, pm_multiref=200))
    _run_stream(default_params("A", BASE_SEED + 62, n_frames=6, pm_split1=400, t8_prob=500))
    _run_stream(default_params("B", BASE_SEED + 63, n_frames=6, pm_split1=600, pm_deep=200, pm_multiref=300, cbp_prob=600))


@pytest.mark.parametrize("w,h,ver,nclips", [(16, 16, 2, 1), (32, 16, 1, 3), (48, 32, 2, 6), (256, 16, 1, 5), (512, 32, 2, 2), (80, 48, 2, 7)])
def test_intra_rows_of_four_padding_and_picture_edges(w, h, ver, nclips):
    """mobi_recon_intra carries four macroblocks per wave and pads every dependency level to whole waves; macroblocks at the picture's
    edges take the per-sample ownership path and are sorted behind the others.  Tiny pictures (every macroblock on an edge, width ==
    stride for 256 and 512: the halo wraps into neighbouring rows), intra-heavy P-frames, clip counts that leave 1..3 rows of the last
    wave of a level empty."""
    ps = [default_params("A", BASE_SEED + 900 + 17 * i + w, n_frames=6, width=w, height=h, version=ver, pm_intra=400, mv_range=6,
                         iframe_interval=4 if i & 1 else 0) for i in range(nclips)]
    clips = [generate_clip(p) for p in ps]
    b = MobiclipBatch(nclips, w, h, ver, device_parse=False)
    oras = [OracleDecoder(w, h, ver) for _ in ps]
    for f in range(6):
        rcs, offs = b.decode([c[0] for c in clips], [int(c[1][f]) for c in clips])
        for c in range(nclips):
            oras[c].Data, oras[c].Offset = clips[c][0], int(clips[c][1][f])
            o = oras[c].DecodeFrame()
            assert rcs[c] == 0 and offs[c] == oras[c].Offset, (f, c, rcs[c])
            y, uv = b.planes(c)
            assert np.array_equal(y, o[0]) and np.array_equal(uv, o[1]), (f, c)
    b.close()
    for o in oras:
        o.close()


@pytest.mark.parametrize("q", [12, 22, 27, 31, 34, 37, 41, 46, 52])
def test_residual_rounds_16_bit_and_32_bit_by_quantizer(q):
    """The octet kernel transforms on packed int16 while every area's sum of |coefficient| stays below MOBI_PK_LIMIT and in int32
    otherwise (mobi_kernels.hip, stage C).  Coefficients grow with the quantiser (x 2 every 6 steps), so a sweep through it with many
    coded areas, dense blocks and raw-level escapes crosses that limit inside frames, inside octets and between them; both paths, and
    octets with more than sixteen coded areas (a second packed round), must give the reference's pixels."""
    for ver_cfg, w, h in (("A", 256, 48), ("B", 640, 32)):
        p = default_params(ver_cfg, BASE_SEED + 900 + q, n_frames=5, width=w, height=h, quantizer=q, cbp_prob=800, t8_prob=700, dense_prob=150,
                           escape_prob=60, max_coefs=10, scan_span=40, pm_intra=20, qdelta_prob=300)
        _run_stream(p)


@pytest.mark.parametrize("w,h,ver", [(64, 48, 1), (256, 32, 1), (512, 32, 2), (640, 48, 2), (848, 32, 2), (1024, 32, 2)])
def test_deeper_trees_cell_by_cell(w, h, ver):
    """Every macroblock a deeper partition tree (the octet kernel's cell path: one lane = one 2x2 cell, r04), vectors that leave the picture,
    read the stride padding, wrap around plane rows (Width == Stride: 256, 512, 1024) and into the other chroma plane's half; older
    reference frames; with and without residuals."""
    for cbp in (0, 400):
        p = default_params("A" if ver == 1 else "B", BASE_SEED + 950 + w + cbp, n_frames=7, width=w, height=h, pm_deep=900, pm_split1=50, pm_skip=20,
                           pm_intra=10, pm_multiref=500, mv_range=70, edge_mode=1, cbp_prob=cbp)
        _run_stream(p)


def test_host_parsed_step_chunk_by_chunk_and_its_fallback():
    """mobi_batch_decode with the parse on the host takes the clips in chunks of 128 and uploads a chunk's commands while the next is
    parsed -- when the step fits the buffers as they are; a step larger than every one before it is staged after the parse, as all were
    before.  288 clips (three chunks) of 16 streams; the second half of the batch hands over nothing at step 0 (an exception, as
    Data.Length == 0 would be) and starts its stream a step late: step 0 sizes the buffers for half a batch of I-frames, step 1 brings the
    late half's I-frames beside the first half's P-frames and outgrows them in its second chunk (the fallback, with a chunk already on its
    way), the steps after that go chunk by chunk.  Every clip against its own oracle, every frame."""
    n, distinct, nfr, W, H = 288, 16, 5, 160, 112
    ps = [default_params("A", BASE_SEED + 5200 + i, n_frames=nfr, width=W, height=H, version=2, pm_intra=80, cbp_prob=500) for i in range(distinct)]
    clips = [generate_clip(p) for p in ps]
    late = lambda c: c >= n // 2
    oras = [OracleDecoder(W, H, MobiclipVersion.Moflex3DS) for _ in range(2 * distinct)]  # (stream, on time / late)
    b = MobiclipBatch(n, W, H, MobiclipVersion.Moflex3DS, device_parse=False)
    for step in range(nfr + 1):
        def frame_of(c):
            f = step - 1 if late(c) else step
            return f if 0 <= f < nfr else None
        datas = []
        for c in range(n):
            f = frame_of(c)
            data, fo = clips[c % distinct]
            datas.append(data[fo[f]:fo[f + 1]] if f is not None else b"")
        rcs, offs = b.decode(datas, [0] * n)
        want = {}
        for k in range(2 * distinct):
            f = frame_of(k % distinct + (n // 2 if k >= distinct else 0))
            data, fo = clips[k % distinct]
            oras[k].Data, oras[k].Offset = (data[fo[f]:fo[f + 1]] if f is not None else b""), 0
            want[k] = (oras[k].DecodeFrame(), oras[k].Offset, oras[k].last_error)
        for c in range(n):
            o, off, err = want[c % distinct + (distinct if late(c) else 0)]
            if o is None:
                assert rcs[c] == err != 0, (step, c)
                continue
            assert rcs[c] == 0 and offs[c] == off, (step, c)
            if c % 7 == 0 or step == nfr:
                y, uv = b.planes(c)
                assert np.array_equal(y, o[0]) and np.array_equal(uv, o[1]), (step, c)
    b.close()
