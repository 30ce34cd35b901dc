"""Container readers (include/mobiclip_demux.h): ctypes mirror of LibMobiclip.Containers.Mods.ModsDemuxer and of the
MOC5 frame loop in the reference GUI.  Host-side byte parsing only; frames come out as numpy views of the file."""
import ctypes as C

import numpy as np

from .decoder import load_library


class _ModsHeader(C.Structure):
    _fields_ = [("mods_string", C.c_char * 4), ("tag_id", C.c_uint16), ("tag_id_size_dword", C.c_uint16),
                ("frame_count", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32), ("fps", C.c_uint32),
                ("audio_codec", C.c_uint16), ("nb_channel", C.c_uint16), ("frequency", C.c_uint32),
                ("biggest_frame", C.c_uint32), ("audio_offset", C.c_uint32), ("keyframe_index_offset", C.c_uint32),
                ("keyframe_count", C.c_uint32)]


class MoflexStream(C.Structure):
    """MoLiveStream chunk of a frame: video (chunk_id 1), audio (2), video with layout (3), timeline (4)."""
    _fields_ = [("chunk_id", C.c_uint32), ("stream_index", C.c_int32), ("codec_id", C.c_uint32),
                ("fps_rate", C.c_uint32), ("fps_scale", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("pel_ratio_rate", C.c_uint32), ("pel_ratio_scale", C.c_uint32),
                ("image_layout", C.c_uint32), ("image_rotation", C.c_uint32),
                ("frequency", C.c_uint32), ("channel", C.c_uint32), ("associated_stream_index", C.c_uint32)]


class _Moc5Info(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("fps_x128", C.c_uint32), ("first_block", C.c_uint32)]


# names must match include/mobiclip_demux.h (tests/test_abi_symbols.py checks header, library and this table)
_SIGS = {
    "mobi_mods_open": (C.c_void_p, [C.c_void_p, C.c_size_t]),
    "mobi_mods_close": (None, [C.c_void_p]),
    "mobi_mods_get_header": (C.c_int, [C.c_void_p, C.POINTER(_ModsHeader)]),
    "mobi_mods_keyframe": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mobi_mods_audio_codebook": (C.c_void_p, [C.c_void_p, C.c_int]),
    "mobi_mods_jump_to_keyframe": (None, [C.c_void_p, C.c_int]),
    "mobi_mods_read_frame": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "mobi_moflex_open": (C.c_void_p, [C.c_void_p, C.c_size_t]),
    "mobi_moflex_close": (None, [C.c_void_p]),
    "mobi_moflex_read_packet": (C.c_int, [C.c_void_p]),
    "mobi_moflex_pop_frame": (C.c_int, [C.c_void_p, C.POINTER(MoflexStream), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "mobi_moflex_next_frame": (C.c_int, [C.c_void_p, C.POINTER(MoflexStream), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "mobi_moc5_open": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(_Moc5Info)]),
    "mobi_moc5_next_block": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]),
}
_BOUND = False


def _lib():
    global _BOUND
    lib = load_library()
    if not _BOUND:
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _BOUND = True
    return lib


class ModsDemuxer:
    """new ModsDemuxer(stream): Header, KeyFrames, ReadFrame(), JumpToKeyFrame() (ModsDemuxer.cs:16-117)."""

    def __init__(self, data):
        self._buf = np.ascontiguousarray(np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data)
        self._lib = _lib()
        self._h = self._lib.mobi_mods_open(self._buf.ctypes.data, self._buf.size)
        if not self._h:
            raise ValueError("not a readable .mods file (shorter than its header says)")
        self.Header = _ModsHeader()
        self._lib.mobi_mods_get_header(self._h, C.byref(self.Header))
        self.KeyFrames = []
        for k in range(self.Header.keyframe_count):
            fn, off = C.c_uint32(), C.c_uint32()
            self._lib.mobi_mods_keyframe(self._h, k, C.byref(fn), C.byref(off))
            self.KeyFrames.append((fn.value, off.value))

    def JumpToKeyFrame(self, k):
        self._lib.mobi_mods_jump_to_keyframe(self._h, int(k))

    def ReadFrame(self):
        """-> (packet bytes as a numpy view, NrAudioPackets, IsKeyFrame), or None after the last frame."""
        p, size, na, key = C.c_void_p(), C.c_uint32(), C.c_uint32(), C.c_int()
        rc = self._lib.mobi_mods_read_frame(self._h, C.byref(p), C.byref(size), C.byref(na), C.byref(key))
        if rc == 0:
            return None
        if rc < 0:
            raise EOFError("file ends inside a packet")
        start = p.value - self._buf.ctypes.data
        return self._buf[start:start + size.value], na.value, bool(key.value)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mobi_mods_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def moc5_info(data):
    buf = np.ascontiguousarray(np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data)
    info = _Moc5Info()
    if _lib().mobi_moc5_open(buf.ctypes.data, buf.size, C.byref(info)) != 0:
        raise ValueError("not a readable MOC5 file")
    return info


def moc5_blocks(data):
    """Yields (decode_offset, block_size) per frame: Data = the whole file, Offset = decode_offset (Form1.cs:293-318)."""
    buf = np.ascontiguousarray(np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data)
    offs = C.c_uint32(moc5_info(buf).first_block)
    lib = _lib()
    while True:
        dec, bs = C.c_int32(), C.c_uint32()
        rc = lib.mobi_moc5_next_block(buf.ctypes.data, buf.size, C.byref(offs), C.byref(dec), C.byref(bs))
        if rc == 0:
            return
        if rc < 0:
            raise EOFError("file ends inside a block header")
        yield dec.value, bs.value


class MoLiveDemux:
    """MoLiveDemux over a file held in memory (MoLiveDemux.cs): ReadPacket() with the reference's return codes, completed
    frames via frames() / next_frame() instead of the OnCompleteFrameReceived event."""

    def __init__(self, data):
        self._buf = np.ascontiguousarray(np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data)
        self._lib = _lib()
        self._h = self._lib.mobi_moflex_open(self._buf.ctypes.data, self._buf.size)
        if not self._h:
            raise MemoryError("mobi_moflex_open")

    def ReadPacket(self):
        return self._lib.mobi_moflex_read_packet(self._h)

    def _take(self, fn):
        st, p, n = MoflexStream(), C.c_void_p(), C.c_size_t()
        rc = fn(self._h, C.byref(st), C.byref(p), C.byref(n))
        if rc <= 0:
            return rc, None
        return rc, (st, np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n.value,)).copy())

    def pop_frame(self):
        return self._take(self._lib.mobi_moflex_pop_frame)[1]

    def next_frame(self):
        """-> (MoflexStream, frame bytes incl. the two appended zero bytes), or None at the end of the stream."""
        rc, fr = self._take(self._lib.mobi_moflex_next_frame)
        if rc < 0:
            raise ValueError(f"Moflex demux error {-rc:#x}")
        return fr

    def frames(self):
        while (fr := self.next_frame()) is not None:
            yield fr

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mobi_moflex_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
