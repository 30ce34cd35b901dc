"""Host-side mirror of the reference decoder class over the C ABI (include/mobiclip_hip.h).

``MobiclipDecoder`` keeps the public surface of LibMobiclip.Codec.Mobiclip.MobiclipDecoder
(MobiclipDecoder.cs:13-61): ctor ``(Width, Height, Version)``, fields ``Data`` / ``Offset`` /
``Width`` / ``Height`` / ``Stride`` / ``Quantizer`` / ``YuvFormat``, plane accessors ``Y[i]`` /
``UV[i]`` and ``DecodeFrame()``.  The callers' contract (MobiConverter/Program.cs:69-71,243-250):

    d.Data = frame; d.Offset = 0; d.DecodeFrame(); audio_start = d.Offset - 2

Differences, on purpose: ``DecodeFrame`` returns the planes (Y, UV) instead of a System.Drawing
Bitmap (the RGB conversion MD.cs:260-323 is outside the graded path), and where the reference
swallows every exception and returns null (MD.cs:325-328) this returns ``None`` and leaves the
reason in ``last_error``.  Reconstruction runs only on the GPU; there is no CPU fallback.
"""
import ctypes as C
import enum
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class MobiclipVersion(enum.IntEnum):  # MD.cs:32-37
    VxDS = 0
    ModsDS = 1
    Moflex3DS = 2


class MobiclipError(RuntimeError):
    pass


MOBI_E_ARG = -7  # include/mobiclip_hip.h


# names must match include/mobiclip_hip.h (tests/test_abi_symbols.py checks the header against the .so)
_SIGS = {
    "mobi_create": (C.c_void_p, [C.c_uint32, C.c_uint32, C.c_int, C.c_int]),
    "mobi_destroy": (None, [C.c_void_p]),
    "mobi_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int32)]),
    "mobi_get_planes": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mobi_get_argb": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mobi_selftest_div239": (C.c_longlong, [C.c_int]),
    "mobi_stride": (C.c_int, [C.c_void_p]),
    "mobi_quantizer": (C.c_uint32, [C.c_void_p]),
    "mobi_yuv_format": (C.c_uint32, [C.c_void_p]),
    "mobi_width": (C.c_uint32, [C.c_void_p]),
    "mobi_height": (C.c_uint32, [C.c_void_p]),
    "mobi_batch_create": (C.c_void_p, [C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_int]),
    "mobi_batch_destroy": (None, [C.c_void_p]),
    "mobi_batch_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mobi_batch_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mobi_batch_wait": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mobi_batch_in_flight": (C.c_int, [C.c_void_p]),
    "mobi_batch_decode_gop": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mobi_batch_gop_begin": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mobi_batch_gop_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mobi_batch_gop_in_flight": (C.c_int, [C.c_void_p]),
    "mobi_batch_gop_frames_pending": (C.c_int, [C.c_void_p]),
    "mobi_batch_host_clips": (C.c_int, [C.c_void_p]),
    "mobi_batch_compare_clips": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mobi_forward_dct": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "mobi_batch_get_planes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mobi_batch_convert_argb": (C.c_int, [C.c_void_p]),
    "mobi_batch_get_argb": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mobi_batch_get_argb_at": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "mobi_batch_quantizer": (C.c_uint32, [C.c_void_p, C.c_int]),
    "mobi_batch_yuv_format": (C.c_uint32, [C.c_void_p, C.c_int]),
    "mobi_batch_set_parse_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "mobi_batch_lockstep_finished": (C.c_int, [C.c_void_p]),
    "mobi_batch_last_decode_ms": (C.c_float, [C.c_void_p]),
    "mobi_batch_motion_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mobi_batch_stride": (C.c_int, [C.c_void_p]),
    "mobi_batch_n_clips": (C.c_int, [C.c_void_p]),
    "mobi_batch_preload": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]),
    "mobi_batch_preload_clone": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mobi_batch_commit": (C.c_int, [C.c_void_p]),
    "mobi_batch_replay": (C.c_int, [C.c_void_p, C.c_int]),
    "mobi_batch_sync": (C.c_int, [C.c_void_p]),
    "mobi_batch_cmd_bytes": (C.c_uint64, [C.c_void_p, C.c_int]),
    "mobi_batch_intra_stats": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "mobi_batch_time_begin": (C.c_int, [C.c_void_p]),
    "mobi_batch_time_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "mobi_batch_set_kernel_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "mobi_batch_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mobi_error_string": (C.c_char_p, [C.c_int]),
    "mobi_build_info": (C.c_char_p, []),
}

LIB_PATH = os.environ.get("MOBI_LIB") or os.path.join(_HERE, "libmobiclip_hip.so")  # MOBI_LIB: A/B-test another build of the library


def bind_library(path):
    """dlopen one build of the library and bind every entry point of include/mobiclip_hip.h"""
    lib = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def load_library():
    """dlopen libmobiclip_hip.so and bind every entry point.  Raises if it was not built: the HIP
    library is the only implementation, there is nothing to fall back to."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} missing: run `python -m mobiclipdecoder_amd.build` (or __graft_entry__.build())")
        _LIB = bind_library(LIB_PATH)
    return _LIB


def default_device():
    """HIP ordinal used when a caller names none: MOBI_DEVICE (default 0).  Lets a whole test run exercise another GPU of the node --
    MOBI_DEVICE=1 python -m pytest tests -m gpu -- so that no code path is only ever reached with ordinal 0 (the path shards by clip over
    GPUs: SURVEY.md 8(e))."""
    return int(os.environ.get("MOBI_DEVICE", "0"))


def error_string(rc):
    return load_library().mobi_error_string(rc).decode()


def _as_u8(data):
    a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    if a.dtype != np.uint8:
        a = a.view(np.uint8)
    return np.ascontiguousarray(a)


class _PlaneRing:
    """d.Y[i] / d.UV[i] (MD.cs:19-20): device ring slot i copied to a numpy array on access."""

    def __init__(self, dec, which):
        self._dec, self._which = dec, which

    def __getitem__(self, idx):
        return self._dec._plane(idx, self._which)

    def __len__(self):
        return 6


class MobiclipDecoder:
    def __init__(self, Width, Height, Version, device=None):
        self._lib = load_library()
        self.Width, self.Height, self.Version = int(Width), int(Height), MobiclipVersion(Version)
        device = default_device() if device is None else device
        self._h = self._lib.mobi_create(self.Width, self.Height, int(self.Version), device)
        if not self._h:
            raise MobiclipError(
                f"mobi_create({Width},{Height},{Version},dev={device}) failed: {error_string(-8)} "
                "(dimensions must be multiples of 16; a HIP device is required)")
        self.Data = None
        self.Offset = 0
        self.last_error = 0
        self.Y = _PlaneRing(self, 0)
        self.UV = _PlaneRing(self, 1)

    # -- reference fields -------------------------------------------------------------------
    @property
    def Stride(self):
        return self._lib.mobi_stride(self._h)

    @property
    def Quantizer(self):
        return self._lib.mobi_quantizer(self._h)

    @property
    def YuvFormat(self):
        return self._lib.mobi_yuv_format(self._h)

    # -- reference method -------------------------------------------------------------------
    def DecodeFrame(self):
        """MD.cs:56-61.  Returns (Y, UV) of the new frame, or None where the reference returns null."""
        if self.Data is None:
            self.last_error = -2
            return None
        buf = _as_u8(self.Data)
        off = C.c_int32(int(self.Offset))
        rc = self._lib.mobi_decode(self._h, buf.ctypes.data, buf.size, C.byref(off))
        self.Offset = off.value
        self.last_error = rc
        if rc != 0:
            return None
        return self._plane(0, 0), self._plane(0, 1)

    def _plane(self, idx, which):
        S, H = self.Stride, self.Height
        y = np.empty(S * H, np.uint8) if which == 0 else None
        uv = np.empty(S * H // 2, np.uint8) if which == 1 else None
        rc = self._lib.mobi_get_planes(self._h, idx, y.ctypes.data if y is not None else None,
                                       uv.ctypes.data if uv is not None else None)
        if rc == -2:
            return None  # null slot, like the reference's unfilled ring entries
        if rc != 0:
            raise MobiclipError(error_string(rc))
        return (y.reshape(H, S) if which == 0 else uv.reshape(H // 2, S))

    def Bitmap(self):
        """The Bitmap DecodeFrame() returns in the reference (MD.cs:260-323), for the frame just decoded:
        (Height, Width) uint32 array of 0xAARRGGBB; None before the first frame."""
        out = np.empty((self.Height, self.Width), np.uint32)
        rc = self._lib.mobi_get_argb(self._h, out.ctypes.data)
        if rc == -2:  # MOBI_E_NULLREF
            return None
        if rc != 0:
            raise MobiclipError(error_string(rc))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mobi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def unpack_motion_search(words):
    """Packed results of mobi_batch_motion_search / the oracle -> dict(dx, dy, frame, score)."""
    w = np.asarray(words, dtype=np.uint32)
    return {"dx": (w & 0xFF).astype(np.int8).astype(np.int32), "dy": ((w >> 8) & 0xFF).astype(np.int8).astype(np.int32),
            "frame": ((w >> 16) & 7).astype(np.int32), "score": (w >> 20).astype(np.int32), "packed": w}


class MobiclipBatch:
    """N independent decoder instances of equal geometry decoded in lock step on one GPU
    (decoder instances share nothing: MD.cs:15-39).  Also the pre-parsed replay path used for
    throughput measurement."""

    def __init__(self, n_clips, Width, Height, Version, device=None, device_parse=None):
        """device_parse: True = decode() parses the bitstreams on the GPU (one wavefront per clip, mobi_dparse.hip),
        "lockstep" = the same with the lock-step parser in front (clips in lock step, one per lane, mobi_lsparse.hip),
        False = on host threads, "hybrid" = most clips on the GPU and a fixed share (a fifth, at most 1024) on the host pool at the same
        time, None = library default (device parse from 20 clips per host parse thread, 640 at least; env MOBI_DEVICE_PARSE=0/1/2)."""
        self._lib = load_library()
        self.n, self.Width, self.Height, self.Version = int(n_clips), int(Width), int(Height), MobiclipVersion(Version)
        device = default_device() if device is None else device
        self._h = self._lib.mobi_batch_create(self.n, self.Width, self.Height, int(self.Version), device)
        if not self._h:
            raise MobiclipError(f"mobi_batch_create failed: {error_string(-8)}")
        self.Stride = self._lib.mobi_batch_stride(self._h)
        if device_parse is not None:
            if isinstance(device_parse, int) and not isinstance(device_parse, bool):
                mode = device_parse  # the library's own numbering: 0 host, 1 device, 2 hybrid, 3 device with the lock-step parser in front
            else:
                mode = 2 if device_parse == "hybrid" else 3 if device_parse == "lockstep" else int(bool(device_parse))
            rc = self._lib.mobi_batch_set_parse_mode(self._h, mode)
            if rc != 0:
                raise MobiclipError(error_string(rc))

    def decode(self, datas, offsets):
        """One DecodeFrame() per clip.  datas: list of byte buffers; offsets: list of ints.
        Returns (rc list, new offsets list)."""
        bufs = [_as_u8(d) for d in datas]
        ptrs = (C.c_void_p * self.n)(*[b.ctypes.data for b in bufs])
        lens = (C.c_size_t * self.n)(*[b.size for b in bufs])
        offs = (C.c_int32 * self.n)(*[int(o) for o in offsets])
        rcs = (C.c_int * self.n)()
        e = self._lib.mobi_batch_decode(self._h, ptrs, lens, offs, rcs)
        if e != 0:
            raise MobiclipError(error_string(e))
        return list(rcs), list(offs)

    def submit(self, datas, offsets):
        """Asynchronous decode(): enqueue one frame step (device parse only, at most two in flight); the buffers may be reused at
        once.  Results come from wait(), oldest step first."""
        bufs = [_as_u8(d) for d in datas]
        ptrs = (C.c_void_p * self.n)(*[b.ctypes.data for b in bufs])
        lens = (C.c_size_t * self.n)(*[b.size for b in bufs])
        offs = (C.c_int32 * self.n)(*[int(o) for o in offsets])
        e = self._lib.mobi_batch_submit(self._h, ptrs, lens, offs)
        if e != 0:
            raise MobiclipError(error_string(e))

    # -- frame-parallel groups: K consecutive frames of every clip per call (mobiclip_hip.h, mobi_batch_decode_gop) -------------
    def _gop_arrays(self, frames, offsets):
        """frames[k][c] = byte buffer of frame k of clip c; offsets[k][c] (or None: all 0) -> the C arrays, [k * n + c]"""
        K = len(frames)
        bufs = [_as_u8(d) for fr in frames for d in fr]
        assert len(bufs) == K * self.n
        ptrs = (C.c_void_p * (K * self.n))(*[b.ctypes.data for b in bufs])
        lens = (C.c_size_t * (K * self.n))(*[b.size for b in bufs])
        flat = [0] * (K * self.n) if offsets is None else [int(o) for row in offsets for o in row]
        offs = (C.c_int32 * (K * self.n))(*flat)
        return K, bufs, ptrs, lens, offs

    def decode_gop(self, frames, offsets=None):
        """K = len(frames) <= 6 DecodeFrame() calls per clip in one go, parsed side by side on the GPU.  -> (rc, offsets), each a list of
        K lists of n: exactly what K decode() calls return."""
        K, bufs, ptrs, lens, offs = self._gop_arrays(frames, offsets)
        rcs = (C.c_int * (K * self.n))()
        e = self._lib.mobi_batch_decode_gop(self._h, K, ptrs, lens, offs, rcs)
        if e != 0:
            raise MobiclipError(error_string(e))
        return [list(rcs[k * self.n:(k + 1) * self.n]) for k in range(K)], [list(offs[k * self.n:(k + 1) * self.n]) for k in range(K)]

    def gop_begin(self, frames, offsets=None):
        """first half of decode_gop(): gather, upload (and parse, when no group is in front); at most two groups begun and not finished"""
        K, bufs, ptrs, lens, offs = self._gop_arrays(frames, offsets)
        e = self._lib.mobi_batch_gop_begin(self._h, K, ptrs, lens, offs)
        if e != 0:
            raise MobiclipError(error_string(e))

    def gop_finish(self):
        """second half, for the OLDEST group begun: hand-overs, the reconstruction steps; -> (rc, offsets) as decode_gop().  A group of more
        than six frames (gop_begin takes up to 128) is handed out six at a time: call again while gop_frames_pending() > 0."""
        K = min(6, self._lib.mobi_batch_gop_frames_pending(self._h))
        offs = (C.c_int32 * (K * self.n))()
        rcs = (C.c_int * (K * self.n))()
        e = self._lib.mobi_batch_gop_finish(self._h, offs, rcs)
        if e != 0:
            raise MobiclipError(error_string(e))
        return [list(rcs[k * self.n:(k + 1) * self.n]) for k in range(K)], [list(offs[k * self.n:(k + 1) * self.n]) for k in range(K)]

    def gop_frames_pending(self):
        return self._lib.mobi_batch_gop_frames_pending(self._h)

    def compare_clips(self, modulus):
        """clips whose newest frame differs from that of clip (index mod modulus), compared on the device (batches made of copies)"""
        e = self._lib.mobi_batch_compare_clips(self._h, int(modulus), None)
        if e < 0:
            raise MobiclipError(error_string(e))
        return e

    def host_clips(self):
        """clips the host parser parses at present: all of them in host mode, else the hybrid share plus every clip that has had a frame
        the device parser could not finish (the result is the same either way: mobiclip_hip.h, mobi_batch_set_parse_mode)"""
        return self._lib.mobi_batch_host_clips(self._h)

    def lockstep_finished(self):
        """device_parse="lockstep": clips of the last finished step the lock-step parser finished itself (-1 in other modes)."""
        return self._lib.mobi_batch_lockstep_finished(self._h)

    def wait(self):
        """(rc list, new offsets list) of the oldest submitted step, when its reconstruction is done."""
        offs = (C.c_int32 * self.n)()
        rcs = (C.c_int * self.n)()
        e = self._lib.mobi_batch_wait(self._h, offs, rcs)
        if e != 0:
            raise MobiclipError(error_string(e))
        return list(rcs), list(offs)

    def planes(self, clip, idx=0):
        S, H = self.Stride, self.Height
        y = np.empty(S * H, np.uint8)
        uv = np.empty(S * H // 2, np.uint8)
        rc = self._lib.mobi_batch_get_planes(self._h, clip, idx, y.ctypes.data, uv.ctypes.data)
        if rc == -2:
            return None
        if rc != 0:
            raise MobiclipError(error_string(rc))
        return y.reshape(H, S), uv.reshape(H // 2, S)

    def convert_argb(self):
        """Bitmaps of every clip's current frame into a device-resident buffer (asynchronous)."""
        rc = self._lib.mobi_batch_convert_argb(self._h)
        if rc != 0:
            raise MobiclipError(error_string(rc))

    def bitmap(self, clip, idx=0):
        """the Bitmap of the frame at ring index idx (0 = the newest)"""
        out = np.empty((self.Height, self.Width), np.uint32)
        rc = self._lib.mobi_batch_get_argb_at(self._h, clip, idx, out.ctypes.data) if idx else self._lib.mobi_batch_get_argb(self._h, clip, out.ctypes.data)
        if rc != 0:
            raise MobiclipError(error_string(rc))
        return out

    def quantizer(self, clip):
        return self._lib.mobi_batch_quantizer(self._h, clip)

    def yuv_format(self, clip):
        return self._lib.mobi_batch_yuv_format(self._h, clip)

    def motion_search(self, pictures):
        """Analyzer.InterPredict2x2 for every 2x2 luma block (Analyzer.cs:608-693) of `pictures` (one (Height, Width) uint8 luma
        picture per clip) against this batch's ring.  -> dict of (n, mbh, mbw, 8, 8) arrays: dx, dy (half pels), frame, score."""
        pics = [np.ascontiguousarray(p, dtype=np.uint8) for p in pictures]
        assert len(pics) == self.n and all(p.shape == (self.Height, self.Width) for p in pics)
        ptrs = (C.c_void_p * self.n)(*[p.ctypes.data for p in pics])
        out = np.empty((self.n, self.Height // 16, self.Width // 16, 8, 8), np.uint32)
        rc = self._lib.mobi_batch_motion_search(self._h, ptrs, out.ctypes.data)
        if rc != 0:
            raise MobiclipError(error_string(rc))
        return unpack_motion_search(out)

    def last_decode_ms(self):
        """Wall time the last decode() spent inside the library."""
        return float(self._lib.mobi_batch_last_decode_ms(self._h))

    # -- replay ---------------------------------------------------------------------------------
    def preload(self, clip, data, frame_off):
        buf = _as_u8(data)
        fo = np.ascontiguousarray(frame_off, dtype=np.uint32)
        n_frames = fo.size - 1
        rcs = (C.c_int * n_frames)()
        rc = self._lib.mobi_batch_preload(self._h, clip, buf.ctypes.data, buf.size, fo.ctypes.data, n_frames, rcs)
        if rc == MOBI_E_ARG:  # bad clip index / frame offsets: nothing was staged (a stream error comes back per frame instead)
            raise MobiclipError(error_string(rc))
        return list(rcs)

    def preload_clone(self, clip, src):
        rc = self._lib.mobi_batch_preload_clone(self._h, clip, src)
        if rc != 0:
            raise MobiclipError(error_string(rc))

    def commit(self):
        rc = self._lib.mobi_batch_commit(self._h)
        if rc != 0:
            raise MobiclipError(error_string(rc))

    def replay(self, frame_idx):
        rc = self._lib.mobi_batch_replay(self._h, frame_idx)
        if rc != 0:
            raise MobiclipError(error_string(rc))

    def sync(self):
        return self._lib.mobi_batch_sync(self._h)

    def cmd_bytes(self, frame_idx):
        return int(self._lib.mobi_batch_cmd_bytes(self._h, frame_idx))

    def intra_stats(self, frame_idx):
        """(intra macroblocks, command-list bytes that belong to them) of one frame step, all clips."""
        n, by = C.c_uint64(), C.c_uint64()
        rc = self._lib.mobi_batch_intra_stats(self._h, frame_idx, C.byref(n), C.byref(by))
        if rc != 0:
            raise MobiclipError(error_string(rc))
        return int(n.value), int(by.value)

    def time_begin(self):
        self._lib.mobi_batch_time_begin(self._h)

    def time_end(self):
        ms = C.c_float()
        rc = self._lib.mobi_batch_time_end(self._h, C.byref(ms))
        if rc != 0:
            raise MobiclipError(error_string(rc))
        return ms.value

    def set_kernel_timing(self, level):
        """0 / False: off; 1 / True: HIP events around the inter launches; 2: around every launch."""
        self._lib.mobi_batch_set_kernel_timing(self._h, int(level))

    def kernel_ms(self):
        a, b, na, nb = C.c_float(), C.c_float(), C.c_int(), C.c_int()
        self._lib.mobi_batch_kernel_ms(self._h, C.byref(a), C.byref(b), C.byref(na), C.byref(nb))
        return {"inter_ms": a.value, "intra_ms": b.value, "inter_launches": na.value, "intra_launches": nb.value}

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mobi_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def forward_dct(blocks, device=None):
    """MobiEncoder.DCT64 / DCT16 (Encoder/MobiEncoder.cs:962, 1146) on the GPU: blocks = int32 array (n_blocks, 64) or (n_blocks, 16)
    of residuals -> coefficients of the same shape, as the reference returns them."""
    a = np.ascontiguousarray(blocks, dtype=np.int32)
    if a.ndim != 2 or a.shape[1] not in (64, 16):
        raise ValueError("blocks must be (n, 64) or (n, 16)")
    out = np.empty_like(a)
    rc = load_library().mobi_forward_dct(default_device() if device is None else device, 8 if a.shape[1] == 64 else 4, a.ctypes.data, out.ctypes.data, a.shape[0])
    if rc != 0:
        raise MobiclipError(error_string(rc))
    return out
