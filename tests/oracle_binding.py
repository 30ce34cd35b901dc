"""ctypes binding of the CPU oracle (oracle/mobi_oracle.h).  TEST INFRASTRUCTURE: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product package."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "_build", "libmobi_oracle.so")
        if not os.path.exists(path):
            from mobiclipdecoder_amd import build
            build.build_oracle()
        L = C.CDLL(path)
        L.mobi_oracle_create.restype = C.c_void_p
        L.mobi_oracle_create.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
        L.mobi_oracle_destroy.argtypes = [C.c_void_p]
        L.mobi_oracle_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int32)]
        L.mobi_oracle_y.restype = C.POINTER(C.c_uint8)
        L.mobi_oracle_y.argtypes = [C.c_void_p, C.c_int]
        L.mobi_oracle_uv.restype = C.POINTER(C.c_uint8)
        L.mobi_oracle_uv.argtypes = [C.c_void_p, C.c_int]
        L.mobi_oracle_stride.argtypes = [C.c_void_p]
        L.mobi_oracle_quantizer.argtypes = [C.c_void_p]
        L.mobi_oracle_quantizer.restype = C.c_uint32
        L.mobi_oracle_yuvformat.argtypes = [C.c_void_p]
        L.mobi_oracle_yuvformat.restype = C.c_uint32
        L.mobi_oracle_argb.argtypes = [C.c_void_p, C.c_void_p]
        L.mobi_oracle_motion_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mobi_oracle_motion_search.restype = None
        L.mobi_oracle_decode_clip.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.mobi_oracle_internal.argtypes = [C.c_void_p]
        L.mobi_oracle_internal.restype = C.POINTER(C.c_uint32)
        L.mobi_oracle_idct8.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.mobi_oracle_idct4.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.mobi_oracle_copyblock.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.mobi_oracle_predict.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.mobi_oracle_plane.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        _LIB = L
    return _LIB


class OracleDecoder:
    """Same call shape as the reference class: Data / Offset / DecodeFrame() / Y[i] / UV[i]."""

    def __init__(self, width, height, version):
        self.L = lib()
        self.Width, self.Height, self.Version = width, height, int(version)
        self.h = self.L.mobi_oracle_create(width, height, int(version))
        self.Stride = self.L.mobi_oracle_stride(self.h)
        self.Data = None
        self.Offset = 0
        self.last_error = 0

    def DecodeFrame(self):
        buf = np.ascontiguousarray(np.frombuffer(self.Data, np.uint8) if not isinstance(self.Data, np.ndarray) else self.Data)
        off = C.c_int32(int(self.Offset))
        rc = self.L.mobi_oracle_decode(self.h, buf.ctypes.data, buf.size, C.byref(off))
        self.Offset = off.value
        self.last_error = rc
        return None if rc != 0 else (self.y(0), self.uv(0))

    def y(self, idx):
        p = self.L.mobi_oracle_y(self.h, idx)
        if not p:
            return None
        return np.ctypeslib.as_array(p, (self.Height, self.Stride)).copy()

    def uv(self, idx):
        p = self.L.mobi_oracle_uv(self.h, idx)
        if not p:
            return None
        return np.ctypeslib.as_array(p, (self.Height // 2, self.Stride)).copy()

    def argb(self):
        """Bitmap of ring slot 0 (MD.cs:260-323) as (Height, Width) uint32 0xAARRGGBB; None before the first frame."""
        out = np.empty((self.Height, self.Width), np.uint32)
        return out if self.L.mobi_oracle_argb(self.h, out.ctypes.data) == 0 else None

    def decode_clip(self, data, frame_off, with_bitmap=False):
        """Every frame of a clip in one C call (timing); returns the number of frames decoded or a negative error."""
        buf = np.ascontiguousarray(data, dtype=np.uint8)
        fo = np.ascontiguousarray(frame_off, dtype=np.uint32)
        argb = np.empty((self.Height, self.Width), np.uint32) if with_bitmap else None
        return self.L.mobi_oracle_decode_clip(self.h, buf.ctypes.data, fo.ctypes.data, fo.size - 1, argb.ctypes.data if with_bitmap else None)

    def motion_search(self, picture):
        """Analyzer.InterPredict2x2 over every 2x2 block (Analyzer.cs:608-693) -> packed (mbh, mbw, 8, 8) uint32."""
        pic = np.ascontiguousarray(picture, dtype=np.uint8)
        assert pic.shape == (self.Height, self.Width)
        out = np.empty((self.Height // 16, self.Width // 16, 8, 8), np.uint32)
        self.L.mobi_oracle_motion_search(self.h, pic.ctypes.data, out.ctypes.data)
        return out

    @property
    def Quantizer(self):
        return self.L.mobi_oracle_quantizer(self.h)

    @property
    def YuvFormat(self):
        return self.L.mobi_oracle_yuvformat(self.h)

    def close(self):
        if self.h:
            self.L.mobi_oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
