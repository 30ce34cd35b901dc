"""ctypes binding of tests/tools/libmobi_cmdinterp.so -- the CPU command-list interpreter (TEST TOOL)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(os.path.join(ROOT, "tests", "tools", "libmobi_cmdinterp.so"))
        L.mobi_cmdinterp_create.restype = C.c_void_p
        L.mobi_cmdinterp_create.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
        L.mobi_cmdinterp_destroy.argtypes = [C.c_void_p]
        L.mobi_cmdinterp_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int32)]
        L.mobi_cmdinterp_y.restype = C.POINTER(C.c_uint8)
        L.mobi_cmdinterp_y.argtypes = [C.c_void_p, C.c_int]
        L.mobi_cmdinterp_uv.restype = C.POINTER(C.c_uint8)
        L.mobi_cmdinterp_uv.argtypes = [C.c_void_p, C.c_int]
        for n in ("desc", "intra_mbs", "level_start", "intra_items", "payload"):
            f = getattr(L, "mobi_cmdinterp_" + n)
            f.argtypes = [C.c_void_p]
            f.restype = C.POINTER(C.c_uint32)
        for n in ("stride", "quantizer", "cmd_bytes", "levels", "n_mbs", "n_intra", "payload_words"):
            f = getattr(L, "mobi_cmdinterp_" + n)
            f.argtypes = [C.c_void_p]
            f.restype = C.c_uint32
        _LIB = L
    return _LIB


class InterpDecoder:
    def __init__(self, width, height, version):
        self.L = lib()
        self.Width, self.Height = width, height
        self.h = self.L.mobi_cmdinterp_create(width, height, int(version))
        assert self.h
        self.Stride = self.L.mobi_cmdinterp_stride(self.h)
        self.Data, self.Offset, self.last_error = None, 0, 0

    def DecodeFrame(self):
        buf = np.ascontiguousarray(self.Data)
        off = C.c_int32(int(self.Offset))
        rc = self.L.mobi_cmdinterp_decode(self.h, buf.ctypes.data, buf.size, C.byref(off))
        self.Offset, self.last_error = off.value, rc
        return None if rc != 0 else (self.y(0), self.uv(0))

    def y(self, i):
        return np.ctypeslib.as_array(self.L.mobi_cmdinterp_y(self.h, i), (self.Height, self.Stride)).copy()

    def uv(self, i):
        return np.ctypeslib.as_array(self.L.mobi_cmdinterp_uv(self.h, i), (self.Height // 2, self.Stride)).copy()

    @property
    def Quantizer(self):
        return self.L.mobi_cmdinterp_quantizer(self.h)

    @property
    def levels(self):
        return self.L.mobi_cmdinterp_levels(self.h)

    @property
    def cmd_bytes(self):
        return self.L.mobi_cmdinterp_cmd_bytes(self.h)

    def close(self):
        if self.h:
            self.L.mobi_cmdinterp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def command_list(self):
        """The last frame's command list as the host parser left it: (desc [n_mbs, 8], intra_mbs [n_intra], level_start [levels + 2],
        intra_items [n_intra, 4]) -- copies."""
        n, ni, nl = self.L.mobi_cmdinterp_n_mbs(self.h), self.L.mobi_cmdinterp_n_intra(self.h), self.levels
        arr = lambda f, shape: np.ctypeslib.as_array(f(self.h), shape).copy() if int(np.prod(shape)) else np.zeros(shape, np.uint32)
        return (arr(self.L.mobi_cmdinterp_desc, (n, 8)), arr(self.L.mobi_cmdinterp_intra_mbs, (ni,)),
                arr(self.L.mobi_cmdinterp_level_start, (nl + 2,)), arr(self.L.mobi_cmdinterp_intra_items, (ni, 4)))

    def payload(self):
        n = self.L.mobi_cmdinterp_payload_words(self.h)
        return np.ctypeslib.as_array(self.L.mobi_cmdinterp_payload(self.h), (n,)).copy() if n else np.zeros(0, np.uint32)
