"""Container readers (row f2 of SURVEY.md section 8): Mods and MOC5.  The reference ships neither sample files nor
writers for these containers, so the fixtures are written by tests/containers.py from the byte layout the reference's
readers expect; what is checked is that reading gives back exactly what was written, with the reference's ReadFrame /
JumpToKeyFrame / frame-loop semantics, and that the frames drive the decoder (GPU test)."""
import numpy as np
import pytest

from mobiclipdecoder_amd import default_params, generate_clip
from mobiclipdecoder_amd.demux import ModsDemuxer, MoLiveDemux, moc5_blocks, moc5_info
from mobiclipdecoder_amd.streamgen import BASE_SEED
from tests.containers import write_mods, write_moc5, write_moflex


def _frames(cfg, n, **kw):
    p = default_params(cfg, BASE_SEED + 900, n_frames=n, **kw)
    data, fo = generate_clip(p)
    return p, data, [bytes(data[fo[f]:fo[f + 1]]) for f in range(n)]


def test_mods_roundtrip_and_keyframe_semantics():
    p, _, frames = _frames("A", 9, iframe_interval=4)
    keys = [0, 4, 8]
    tail = bytes(range(37))
    blob = write_mods(frames, p.width, p.height, keys, audio_tail=tail)
    d = ModsDemuxer(blob)
    h = d.Header
    assert (h.mods_string, h.frame_count, h.width, h.height, h.keyframe_count) == (b"MODS", 9, p.width, p.height, 3)
    assert h.biggest_frame == max(len(f) for f in frames) + len(tail)
    assert [k[0] for k in d.KeyFrames] == keys
    got = []
    while True:
        r = d.ReadFrame()
        if r is None:
            break
        got.append(r)
    assert len(got) == 9 and d.ReadFrame() is None            # CurFrame >= FrameCount -> null, repeatedly
    for f, (pkt, n_audio, is_key) in enumerate(got):
        assert bytes(pkt) == frames[f] + tail and n_audio == 3
        # ModsDemuxer.cs:102-107: the flag is raised when CurFrame reaches the NEXT index entry; frame 0 itself is
        # entry 0, and JumpToKeyFrame(0) has already stepped past it
        assert is_key == (f in keys[1:])
    d.JumpToKeyFrame(1)                                       # :88-95
    pkt, _, is_key = d.ReadFrame()
    assert bytes(pkt) == frames[4] + tail and not is_key      # NextKeyFrame now points at entry 2
    d.JumpToKeyFrame(7)                                       # out of range: ignored
    assert bytes(d.ReadFrame()[0]) == frames[5] + tail
    d.close()


def test_mods_rejects_truncated_files():
    p, _, frames = _frames("A", 3)
    blob = write_mods(frames, p.width, p.height, [0])
    with pytest.raises(ValueError):
        ModsDemuxer(blob[:0x20])                              # no room for the header
    with pytest.raises(ValueError):
        ModsDemuxer(blob[:-4])                                # the key frame index is cut
    # a packet header that promises more bytes than the file holds
    cut = blob.copy()
    cut[0x30:0x34] = np.frombuffer(np.uint32((len(blob) << 14) & 0xFFFFFFFF).tobytes(), np.uint8)
    d = ModsDemuxer(cut)
    with pytest.raises(EOFError):
        d.ReadFrame()
    d.close()


def test_moc5_block_walk():
    p, _, frames = _frames("C", 5)
    blob, offs = write_moc5(frames, p.width, p.height, fps_x128=3840)
    info = moc5_info(blob)
    assert (info.width, info.height, info.fps_x128, info.first_block) == (p.width, p.height, 3840, 0xE0 + 8)
    walked = list(moc5_blocks(blob))
    assert [w[0] for w in walked] == offs
    for (dec, bs), f in zip(walked, frames):
        assert bytes(blob[dec:dec + len(f)]) == f and bs >= len(f) + 4
    with pytest.raises(ValueError):
        moc5_info(blob[:0x20])


def test_moflex_roundtrip_through_the_reference_muxers_layout():
    """tests/containers.write_moflex restates the reference's own writers (MoflexMuxer / MoflexSimpleVideoMuxer); the C++
    reader restates MoLiveDemux.  Frames of every block shape: tiny, one block, several blocks, exactly one / two full
    blocks (the EndFrame flag then sits on a full block)."""
    rng = np.random.default_rng(5)
    frames = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (1, 17, 3967, 3968, 3969, 7936, 12001, 64)]
    blob = write_moflex(frames, 640, 480, fps_rate=24000, fps_scale=1001, stream_index=0)
    d = MoLiveDemux(blob)
    codes = [d.ReadPacket() for _ in range(3)]
    assert codes == [0, 0, 0]                      # synchronise; adopt the packet size (retry); header + first data block
    first = d.pop_frame()
    assert first is not None and bytes(first[1]) == frames[0] + b"\x00\x00"   # two zero bytes appended (MoLiveDemux.cs:353)
    st = first[0]
    assert (st.chunk_id, st.stream_index, st.codec_id) == (1, 0, 0)
    assert (st.fps_rate, st.fps_scale, st.width, st.height, st.pel_ratio_rate, st.pel_ratio_scale) == (24000, 1001, 640, 480, 1, 1)
    rest = list(d.frames())
    assert [bytes(f[1][:-2]) for f in rest] == frames[1:]
    assert d.ReadPacket() == 73                    # what the reference's callers stop on (Program.cs:164-166)
    d.close()


def test_moflex_synchronisation_search_and_damage():
    frames = [bytes([7]) * 500, bytes([9]) * 5000]
    blob = write_moflex(frames, 256, 192)
    # garbage in front: the reader slides until the synchro header's check word matches (MoLiveDemux.cs:77-96)
    d = MoLiveDemux(np.concatenate([np.frombuffer(b"\x4c\x32junkjunk" + bytes(range(40)), np.uint8), blob]))
    assert [bytes(f[1][:-2]) for f in d.frames()] == frames
    d.close()
    # no synchro header at all
    d = MoLiveDemux(np.zeros(0x3000, np.uint8))
    assert d.ReadPacket() == 0x80
    d.close()
    # a file cut inside the last block: the packet-size check ends the stream (73) and the frame is simply missing
    d = MoLiveDemux(blob[:-0x1000 - 100])
    assert [bytes(f[1][:-2]) for f in d.frames()] == frames[:1]
    d.close()
    # fewer than 14 bytes: "1" (not enough data), no frames
    d = MoLiveDemux(blob[:10])
    assert d.ReadPacket() == 1 and d.next_frame() is None
    d.close()


@pytest.mark.gpu
def test_containers_drive_the_decoder():
    """A .mods file and a MOC5 file, demuxed by the C++ readers, decode to the oracle's frames: packets with audio
    behind the video bits at Offset 0 (Program.cs:241-244), MOC5 with the whole file as Data (Form1.cs:292-302)."""
    from mobiclipdecoder_amd import MobiclipDecoder
    from tests.oracle_binding import OracleDecoder
    p, data, frames = _frames("A", 6)
    d = ModsDemuxer(write_mods(frames, p.width, p.height, [0], audio_tail=b"\x5a" * 64))
    g, o = MobiclipDecoder(d.Header.width, d.Header.height, p.version), OracleDecoder(p.width, p.height, p.version)
    f = 0
    while (r := d.ReadFrame()) is not None:
        g.Data, g.Offset = r[0], 0
        o.Data, o.Offset = r[0], 0  # the same packet: the bit reader's 16-bit read-ahead depends on what follows the frame
        a, b = g.DecodeFrame(), o.DecodeFrame()
        assert a is not None and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and g.Offset == o.Offset, f
        f += 1
    assert f == 6
    g.close()
    p, data, frames = _frames("C", 4)
    blob, _ = write_moc5(frames, p.width, p.height)
    info = moc5_info(blob)
    g, o = MobiclipDecoder(info.width, info.height, p.version), OracleDecoder(p.width, p.height, p.version)
    for f, (dec, _) in enumerate(moc5_blocks(blob)):
        g.Data, g.Offset = blob, dec
        o.Data, o.Offset = blob, dec
        a, b = g.DecodeFrame(), o.DecodeFrame()
        assert a is not None and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), f
    g.close()
    # Moflex: frames come out with two zero bytes appended (what the 3DS decoder's read-ahead expects), Offset 0 (Program.cs:69-71)
    p, data, frames = _frames("B", 5)
    dm = MoLiveDemux(write_moflex(frames, p.width, p.height))
    g = o = None
    for f, (st, pkt) in enumerate(dm.frames()):
        if g is None:
            g, o = MobiclipDecoder(st.width, st.height, p.version), OracleDecoder(st.width, st.height, p.version)
        g.Data, g.Offset = pkt, 0
        o.Data, o.Offset = pkt, 0
        a, b = g.DecodeFrame(), o.DecodeFrame()
        assert a is not None and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and g.Offset == o.Offset, f
    assert f == 4
    g.close()


def _hand_made_moflex():
    """Two 64-byte Moflex packets assembled byte by byte from the container grammar as the reference's READER defines it
    (MoLiveDemux.cs:375-414 sync header and check word, :164-214 stream chunks, :216-264 block flags, :266-373 bit-packed ep
    headers) -- independent of tests/containers.py, which restates the reference's WRITER."""
    time = 0x0000000000012345
    check = (time & 0xFFFF) ^ (time >> 16 & 0xFFFF) ^ (time >> 32 & 0xFFFF) ^ (time >> 48 & 0xFFFF) ^ 0xAAAA
    sync = b"L2" + check.to_bytes(2, "big") + time.to_bytes(8, "big") + (64 - 1).to_bytes(2, "big")
    video = bytes([1, 12,            # chunk type 1 (video), 12 bytes
                   0, 0,             # stream index, codec id
                   0, 24, 0, 1,      # fps 24 / 1
                   1, 0, 0, 192,     # 256 x 192
                   1, 1])            # pel ratio 1 : 1
    end_of_chunks = bytes([0, 0])    # type 0, no body

    def ep(payload):  # stream 0 (unary length 1: "1", index "0"), end of frame "1", frame type ("1", "0"), time stamp (+, "1" = 28 bits, 0), size - 1
        bits = "1" + "0" + "1" + "1" + "0" + "0" + "1" + "0" * 28 + format(len(payload) - 1, "013b")
        assert len(bits) % 8 == 0
        return int(bits, 2).to_bytes(len(bits) // 8, "big") + payload

    p1 = sync + video + end_of_chunks + bytes([0x00]) + ep(bytes([0xAA, 0xBB, 0xCC, 0xDD, 0xEE]))  # flags 0: fixed size, no counter, sync counter 0
    p2 = bytes([0x00]) + ep(bytes([1, 2, 3]))
    return p1.ljust(64, b"\x00") + p2.ljust(64, b"\x00")


def test_moflex_hand_assembled_packets():
    d = MoLiveDemux(np.frombuffer(_hand_made_moflex(), np.uint8))
    got = list(d.frames())
    assert [bytes(f[1]) for f in got] == [bytes([0xAA, 0xBB, 0xCC, 0xDD, 0xEE, 0, 0]), bytes([1, 2, 3, 0, 0])]
    st = got[0][0]
    assert (st.chunk_id, st.stream_index, st.fps_rate, st.fps_scale, st.width, st.height) == (1, 0, 24, 1, 256, 192)
    assert d.ReadPacket() == 73
    d.close()
    # the check word decides: one flipped bit in the time stamp and no synchronisation is found
    bad = bytearray(_hand_made_moflex())
    bad[7] ^= 1
    d = MoLiveDemux(np.frombuffer(bytes(bad), np.uint8))
    assert d.ReadPacket() == 0x80
    d.close()


def test_damaged_containers_do_not_hang_the_caller():
    """A packet that ends right behind its stream table makes the reference's reader drop and regain synchronisation on the same
    bytes for ever (ReadPacket returns 0 without moving); next_frame gives up instead.  A MOC5 block size near 2^32 must not
    wrap the offset backwards."""
    time = 5
    check = (time & 0xFFFF) ^ 0xAAAA
    blob = b"L2" + check.to_bytes(2, "big") + time.to_bytes(8, "big") + (16 - 1).to_bytes(2, "big") + bytes([0, 0])
    d = MoLiveDemux(np.frombuffer(blob, np.uint8))
    with pytest.raises(ValueError):
        d.next_frame()
    d.close()
    import ctypes as C
    from mobiclipdecoder_amd.demux import _lib as demux_lib
    L = demux_lib()
    moc = np.zeros(64, np.uint8)
    moc[8:12] = [0xFE, 0xFF, 0xFF, 0xFF]  # block size 0xFFFFFFFE at offset 8
    offs = C.c_uint32(8)
    L.mobi_moc5_next_block.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
    L.mobi_moc5_next_block.restype = C.c_int
    # the oversized block is handed out once (the reference would pass it to the decoder), the offset lands on the end of the file,
    # and the walk is over: no wrap, no loop
    assert L.mobi_moc5_next_block(moc.ctypes.data, moc.size, C.byref(offs), None, None) == 1 and offs.value == 64
    assert L.mobi_moc5_next_block(moc.ctypes.data, moc.size, C.byref(offs), None, None) == 0


def test_moc5_truncated_last_block_is_still_handed_out():
    """Form1.cs:282-320 passes the whole file as Data for every block, the last one included even if the file was cut inside it."""
    p = default_params("A", BASE_SEED + 77, n_frames=3, width=64, height=48)
    data, fo = generate_clip(p)
    frames = [data[fo[f]:fo[f + 1]] for f in range(p.n_frames)]
    blob, offs = write_moc5(frames, p.width, p.height)
    cut = blob[: len(blob) - 5]  # inside the last block
    walked = list(moc5_blocks(cut))
    assert len(walked) == 3 and [w[0] for w in walked] == [w[0] for w in moc5_blocks(blob)]


def test_moflex_file_shorter_than_the_default_window():
    """The reference reads into a zero-filled byte[0x1000] before it knows the packet size (MoLiveDemux.cs:71): a file shorter than
    that is parsed against zeros behind its end, not rejected as out of range."""
    p = default_params("B", BASE_SEED + 78, n_frames=2, width=64, height=48)
    data, fo = generate_clip(p)
    frames = [data[fo[f]:fo[f + 1]] for f in range(p.n_frames)]
    blob = write_moflex(frames, p.width, p.height)[:-0x1000]  # without the 0x1000 zero bytes FinalizeMoflex appends
    assert len(blob) < 0x1000
    d = MoLiveDemux(blob)
    codes = [d.ReadPacket() for _ in range(4)]
    # synchronisation is found in the short window, and no call ends in the "managed exception" code: reads behind the file's end
    # see the zero-filled rest of the reference's packet array
    assert codes[0] == 0 and all(c not in (0xFFFFFFFF, -1) for c in codes), codes
    d.close()
