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
