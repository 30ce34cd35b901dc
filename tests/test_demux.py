"""Container readers (row f2 of SURVEY.md section 8): Mods and MOC5.  The reference ships neither sample files nor
writers for these containers, so the fixtures are written by tests/containers.py from the byte layout the reference's
readers expect; what is checked is that reading gives back exactly what was written, with the reference's ReadFrame /
JumpToKeyFrame / frame-loop semantics, and that the frames drive the decoder (GPU test)."""
import numpy as np
import pytest

from mobiclipdecoder_amd import default_params, generate_clip
from mobiclipdecoder_amd.demux import ModsDemuxer, moc5_blocks, moc5_info
from mobiclipdecoder_amd.streamgen import BASE_SEED
from tests.containers import write_mods, write_moc5


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
