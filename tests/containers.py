"""Test fixtures: write synthetic .mods / MOC5 files around generated Mobiclip frames, byte layout as the reference's
readers expect it (ModsDemuxer.cs:44-116, Form1.cs:282-320).  The reference ships no writer for either container."""
import struct

import numpy as np


def write_mods(frames, width, height, key_frames, audio_tail=b"", fps=0x18000000):
    """frames: list of bytes (video bits of each frame); key_frames: frame numbers that get an index entry.
    Every packet = video bits + audio_tail (stand-in for the audio packets that follow the video in real files)."""
    packets, offsets = [], []
    pos = 0x30
    for f in frames:
        body = bytes(f) + audio_tail
        offsets.append(pos)
        packets.append(struct.pack("<I", (len(body) << 14) | (3 if audio_tail else 0)) + body)
        pos += 4 + len(body)
    index = b"".join(struct.pack("<II", k, offsets[k]) for k in key_frames)
    biggest = max(len(p) - 4 for p in packets)
    header = struct.pack("<4sHHIIIIHHIIIII", b"MODS", 0x0A, 0x0C, len(frames), width, height, fps,
                         0, 0, 0, biggest, 0, pos, len(key_frames))
    assert len(header) == 0x30
    return np.frombuffer(header + b"".join(packets) + index, np.uint8).copy()


def write_moc5(frames, width, height, fps_x128=30 * 128, header_extra=0xE0):
    """Header of 8 + header_extra bytes (u32 at 4 = header_extra), then blocks: u32 blocksize, 4 bytes, frame bits;
    the next block starts at offs + 4 + (blocksize & ~1), rounded up to 4 (Form1.cs:316-317)."""
    head = bytearray(8 + header_extra)
    head[0:4] = b"MOC5"
    struct.pack_into("<I", head, 4, header_extra)
    struct.pack_into("<I", head, 0xC, fps_x128)
    struct.pack_into("<I", head, 0x1C, width)
    struct.pack_into("<I", head, 0x20, height)
    out = bytearray(head)
    decode_offsets = []
    for f in frames:
        body = bytes(f)
        blocksize = 4 + len(body) + (len(body) & 1)      # covers the 4 unknown bytes + the bits, kept even
        decode_offsets.append(len(out) + 8)
        out += struct.pack("<I", blocksize) + b"\x00" * 4 + body
        target = decode_offsets[-1] - 8 + 4 + (blocksize & ~1)
        while target % 4:
            target += 1
        out += b"\x00" * (target - len(out))
    return np.frombuffer(bytes(out), np.uint8).copy(), decode_offsets


# ---- Moflex: restatement of the reference's own writers, MoflexMuxer.cs:21-95 and MoflexSimpleVideoMuxer.cs:14-66 ----
def _varbyte(v):  # MoLive.WriteVariableByte (MoLive.cs:57-88)
    assert v < (1 << 28)
    if v < 0x80:
        return bytes([v])
    if v < 0x2000:
        return bytes([(v >> 7) | 0x80, v & 0x7F])
    if v < 0x200000:
        return bytes([(v >> 14) | 0x80, ((v >> 7) & 0x7F) | 0x80, v & 0x7F])
    return bytes([(v >> 21) | 0x80, ((v >> 14) | 0x80) & 0xFF, ((v >> 7) & 0x7F) | 0x80, v & 0x7F])


def _synchro_header(ts=1, packetsize_field=0x1000):  # MoflexMuxer.WriteSynchroHeader (:21-36)
    hi = (ts >> 32) & 0xFFFFFFFF
    v19 = hi & 0x7FFFFFFF if (((hi - 1) & 0xFFFFFFFF) >= 0x80000000) else hi
    crc = ((ts >> 16) & 0xFFFF) ^ (v19 >> 16) ^ 0xAAAA ^ (v19 & 0xFFFF) ^ (ts & 0xFFFF)
    return b"\x4c\x32" + struct.pack(">H", crc & 0xFFFF) + struct.pack(">Q", ts) + struct.pack(">H", packetsize_field)


def _ep(ep, data, is_end_frame):  # MoflexMuxer.WriteEp (:55-94)
    if data is None:
        return b"\x00"
    nrbits = 1 if ep == 0 else ep.bit_length()
    val, total = 0x8000000000000000 >> (nrbits - 1), nrbits
    val |= ep << ((64 - total) - nrbits); total += nrbits
    val |= (1 if is_end_frame else 0) << ((64 - total) - 1); total += 1
    if is_end_frame:
        val |= 1 << ((64 - total) - 1); total += 1      # frame type: one bit "1", then its value bit 0
        total += 1
        total += 1                                      # sign 0
        val |= 1 << ((64 - total) - 1); total += 1      # length prefix "1": 28 bits follow
        total += 28                                     # time stamp delta 0
    val |= (len(data) - 1) << ((64 - total) - 13); total += 13
    nrbytes = (total + 4) // 8
    return (val & 0xFFFFFFFFFFFFFFFF).to_bytes(8, "big")[:nrbytes] + bytes(data)


def write_moflex(frames, width, height, fps_rate=30, fps_scale=1, stream_index=0):
    """MoflexSimpleVideoMuxer: synchro header, one MoLiveStreamVideo chunk, terminator, then per frame data blocks of at
    most 0x1000 - 0x80 payload bytes (flag byte 1 = variable packet size), the last EP of a frame flagged EndFrame;
    FinalizeMoflex appends 0x1000 zero bytes."""
    video = bytes([stream_index, 0]) + struct.pack(">HHHH", fps_rate, fps_scale, width, height) + bytes([1, 1])
    out = bytearray(_synchro_header() + _varbyte(1) + _varbyte(12) + video + _varbyte(0) + _varbyte(0))
    lim = 0x1000 - 0x80
    for f in frames:
        data = bytes(f)
        if len(data) <= lim:
            out += b"\x01" + _ep(stream_index, data, True) + _ep(0, None, False)
        else:
            pos, left = 0, len(data)
            while left >= lim:
                out += b"\x01" + _ep(stream_index, data[pos:pos + lim], left == lim) + _ep(0, None, False)
                pos += lim
                left -= lim
            if left > 0:
                out += b"\x01" + _ep(stream_index, data[pos:pos + left], True) + _ep(0, None, False)
    out += bytes(0x1000)
    return np.frombuffer(bytes(out), np.uint8).copy()
