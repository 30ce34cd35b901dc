#!/usr/bin/env python3
"""bench.py -- decoded Mpixels/s of Mobiclip P-frame reconstruction on MI355X (BASELINE.json metric).

A "step" = one P-frame of every resident clip: the reconstruction kernels (MC + dequant/IDCT +
intra) run over pre-parsed command lists that already sit in HBM (the serial VLC
parse and PCIe cannot feed a TB/s kernel, so they are outside the timed region; the `end_to_end`
object of the output line times the whole DecodeFrame path, bitstream in).  Workload = BASELINE config
"640x480 3DS Moflex stream" at a batch sized for this part's HBM: `--clips` independent clips per GPU
(default 24576 = 183 GB of the 288 GB; weak scaling: per-GPU work fixed).  Few, long launches (9 ms) are measurably
more efficient on this part than many short ones: DESIGN.md has the same measurement from 512 to 24576 clips.

  python bench.py                      # 1 GPU, defaults finish in about a minute
  python bench.py --gpus N             # N GPUs of this node: spawns one rank per GPU itself (LOCAL_RANK = GPU index)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   # the same under an external launcher
  python bench.py --gpus 2 --dry-run   # the rank plumbing only (streams, sharding, barrier, reduction, the one line): no HIP call

Clips shard across ranks with no data-path collective (decoder instances share nothing,
MobiclipDecoder.cs:15-39); torch.distributed (a gloo group: nothing travels over RCCL) is used only for the barrier /
max-over-ranks timing.

Prints ONE JSON line (rank 0).  `roofline` = algorithmic bytes of the dominant kernel
(mobi_recon_inter8: the inter macroblocks' pixels and commands only) per launch / its average duration from HIP events
recorded on the launch stream inside the timed region; `roofline.whole_step_frac` = the same for the whole step (both
kernels, every macroblock, every command byte).  `cpu_baseline` = the CPU oracle (a C restatement of the reference decoder;
the C# original cannot run here) on this host, single thread, same stream, parse + reconstruction.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
N_PFRAMES = 32         # P-frames per generated clip (1 I + 32 P)


def cpu_baseline(params, data, fo, budget_s):
    """Oracle ("port" of MobiclipDecoder.cs; the C# original cannot run here, and a C restatement is expected to be faster than it:
    no GC, no per-row allocations) timed on this host on the same stream: one thread (the reference runs one decode thread per open
    file, Form1.cs:199) = `value`; every host cpu with one clip each; and one thread including the Bitmap conversion DecodeFrame() ends with."""
    import concurrent.futures
    from tests.oracle_binding import OracleDecoder  # checker used here ONLY as the reported CPU baseline
    px_clip = params.width * params.height * params.n_frames

    def clips_for(seconds, with_bitmap=False):
        n, t_used = 0, 0.0
        while t_used < seconds:
            d = OracleDecoder(params.width, params.height, params.version)
            t0 = time.perf_counter()
            assert d.decode_clip(data, fo, with_bitmap) == params.n_frames  # one C call per clip: no Python in the loop, the GIL is released
            t_used += time.perf_counter() - t0
            n += 1
            d.close()
        return n, t_used

    n1, t1 = clips_for(budget_s * 0.4)
    nb, tb = clips_for(budget_s * 0.2, with_bitmap=True)
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(cores) as ex:
        done = sum(n for n, _ in ex.map(lambda _: clips_for(budget_s * 0.4), range(cores)))
    t_all = time.perf_counter() - t0
    return {"value": round(n1 * px_clip / t1 / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "sample": f"{n1} x ({params.n_frames}-frame {params.width}x{params.height} clip, 1 I + {params.n_frames - 1} P), "
                      f"VLC parse + reconstruction, planes only, {t1:.1f} s on 1 thread of {cores} host cpus",
            "all_cpus": {"value": round(done * px_clip / t_all / 1e6, 1), "unit": "Mpixels/s", "cores": cores,
                         "sample": f"{done} clips, one decoder per thread, {t_all:.1f} s"},
            "with_bitmap": {"value": round(nb * px_clip / tb / 1e6, 2), "unit": "Mpixels/s", "cores": 1,
                            "sample": f"{nb} clips including the YUV->ARGB Bitmap of every frame (MD.cs:260-323), {tb:.1f} s"}}


def end_to_end(m, streams, W, H, version, device, n_clips, n_steps, device_parse=True, groups=True):
    """Bitstreams in host memory -> planes in HBM: mobi_batch_decode with the parse on the GPU (row f3).  Reported next to
    the headline value, never as it: the timed region of `value` starts with the command lists already in HBM."""
    b = m.MobiclipBatch(n_clips, W, H, version, device=device, device_parse=device_parse)
    ms = []
    for f in range(2 + n_steps):  # the I-frame, one untimed P-frame (allocations), then n_steps P-frames
        datas = [streams[c % len(streams)][1][streams[c % len(streams)][2][f]:streams[c % len(streams)][2][f + 1]] for c in range(n_clips)]
        rcs, _ = b.decode(datas, [0] * n_clips)
        assert all(r == 0 for r in rcs), "stream error in the end-to-end leg"
        if f >= 2:
            ms.append(b.last_decode_ms())
    verified = verify_clips(b, streams, len(streams), n_clips, 1 + n_steps, W, H)
    b.close()
    t = float(np.median(ms))
    out = {"value": round(n_clips * W * H / t / 1e3, 1), "unit": "Mpixels/s", "ms_per_step": round(t, 3), "clips": n_clips, "steps": n_steps,
           "parse": "device: mobi_parse_frames_ls (clips in lock step, one per lane; about two wavefronts per SIMD) in front of mobi_parse_frames" if device_parse == "lockstep"
                    else "device: mobi_parse_frames, one wavefront per clip",
           "includes": "host staging and H2D of the frame bytes, parse, reconstruction, result read-back, sync (wall time inside mobi_batch_decode)",
           "distinct_streams": len(streams), "verified": verified}
    # the same frames through mobi_batch_submit / mobi_batch_wait, two steps in flight: wall time per step over the timed P-frames.
    # The pointer arrays are packed beforehand (what a C caller hands over), as the synchronous figure is the time inside the C call.
    import ctypes as C
    import time as _t
    b = m.MobiclipBatch(n_clips, W, H, version, device=device, device_parse=device_parse)
    lib, h = b._lib, b._h
    packed = []
    for f in range(3 + n_steps):
        bufs = [streams[c % len(streams)][1][streams[c % len(streams)][2][f]:streams[c % len(streams)][2][f + 1]] for c in range(n_clips)]
        packed.append((bufs, (C.c_void_p * n_clips)(*[x.ctypes.data for x in bufs]), (C.c_size_t * n_clips)(*[x.size for x in bufs])))
    offs, outo, rcs = (C.c_int32 * n_clips)(), (C.c_int32 * n_clips)(), (C.c_int * n_clips)()
    for f in range(3):  # the I-frame and two P-frames untimed (allocations)
        assert lib.mobi_batch_submit(h, packed[f][1], packed[f][2], offs) == 0 and lib.mobi_batch_wait(h, outo, rcs) == 0
    t0 = _t.perf_counter()
    assert lib.mobi_batch_submit(h, packed[3][1], packed[3][2], offs) == 0
    for f in range(4, 3 + n_steps):
        assert lib.mobi_batch_submit(h, packed[f][1], packed[f][2], offs) == 0 and lib.mobi_batch_wait(h, outo, rcs) == 0
        assert not any(rcs), "stream error in the asynchronous end-to-end leg"
    assert lib.mobi_batch_wait(h, outo, rcs) == 0 and not any(rcs)
    ta = (_t.perf_counter() - t0) * 1e3 / n_steps
    verified_async = verify_clips(b, streams, len(streams), n_clips, 2 + n_steps, W, H)
    b.close()
    out["async"] = {"value": round(n_clips * W * H / ta / 1e3, 1), "unit": "Mpixels/s", "ms_per_step": round(ta, 3),
                    "how": "mobi_batch_submit / mobi_batch_wait, two steps in flight; wall time per step over the same P-frames",
                    "verified": verified_async}
    if not groups:
        return out
    try:
        # (device_parse=True names the one-wavefront-per-clip parser for EVERY step; a group offers n_clips * K lanes, and which parser is in front
        # of those is the library's own choice, as in any batch created without a mode: mobi_abi.cpp ls_decide)
        out["groups"] = gop_leg(m, streams, W, H, version, device, n_clips, None if device_parse is True else device_parse)
    except Exception as e:  # (e.g. the six command lists of a group do not fit beside what is resident)
        out["groups"] = {"error": f"{type(e).__name__}: {e}"}
    return out


GOP_PARSE_MAX = 32  # frames per mobi_batch_gop_begin in this bench (the library takes up to MOBI_GOP_PARSE_MAX = 128, mobi_gop.h: batches of 1024 clips fill a turn
# of the parser with that many; the two batches measured here fill one with 32 and with 5)
GOP_K = 6  # frames per group: the ring holds six pictures (MD.cs:19-20), so every frame of a group is still readable when the call returns


def gop_leg(m, streams, W, H, version, device, n_clips, device_parse, K=None):
    """r06, frame-parallel groups: the same bitstreams, K consecutive P-frames of every clip per call (mobi_batch_decode_gop: the device
    parsers run over n_clips * K frames side by side; the reconstruction stays one step per frame), and the same with group g + 1 begun
    before group g is finished (mobi_batch_gop_begin / mobi_batch_gop_finish).  ms_per_step = wall time per FRAME step."""
    import ctypes as C
    import time as _t
    # How many frames per group.  The lock-step parser works in TURNS of 2048 waves (eight per CU), a wave is the cheaper per lane the fuller it
    # is, and a workgroup's LDS holds full waves of 64 lanes since the MV row cache left it (mobi_launch_parse_ls): the frames in flight should fill whole turns --
    # n_clips x K close to a multiple of 2048 x 64 -- as far as K (6 per call, 32 per gop_begin here) and HBM allow: a group's command lists
    # are resident until it is reconstructed, worst-case payload room per frame (mobi_abi.cpp, gop_enqueue_parse).
    mbw, n_mbs = W // 16, (W // 16) * (H // 16)
    turn = 2048 * min(64, (160 * 1024 - 18400) // (8 * (96 + 128 + 40)))
    frame_len = max(int(s[2][f + 1] - s[2][f]) for s in streams for f in range(1, len(s[2]) - 1))
    per_frame = 32 * n_mbs + 4 * (64 * n_mbs + (8 * frame_len * 5 // 4 + 2) // 3 + 512) + 4 * n_mbs + 16 * n_mbs + 1200 + 2 * frame_len  # descriptors, payload part, items, sorted items, states, bits
    import torch
    room = torch.cuda.mem_get_info(device)[0] - n_clips * (W if W > 512 else 512 if W > 256 else 256) * H * 9 - (12 << 30)  # free HBM less the rings and a margin

    def pick(kmax, slots):
        fits = [k for k in range(1, kmax + 1) if slots * n_clips * k * per_frame <= room] or [1]
        fill = {k: n_clips * k / (-(-n_clips * k // turn) * turn) for k in fits}
        return max(k for k in fits if fill[k] >= 0.9 * max(fill.values()))

    if K is None:
        K = pick(GOP_K, 1)
    Kp = pick(GOP_PARSE_MAX, 2)
    # its own streams of the same seeds and mix, long enough for a warm-up group and at least four timed ones in the pipelined part (three
    # timed groups, as the 33-frame clips of the replay give, start on a GPU whose clocks have just sat through the checker's seconds)
    G = max(8, -(-(6 if Kp <= 16 else 5) * Kp // K))  # (groups of more than 16 frames: three timed ones are a third of a second)
    # ... and 64 distinct ones: the lock-step parser's lanes are consecutive clips, up to 24 per wave under a group, and a wave that holds two
    # copies of one stream diverges less than content allows (16 distinct streams: 330 instead of 240 Gpixels/s at 4096 clips x 12 -- flattery)
    longer = []
    for j in range(max(64, len(streams))):
        p = streams[j % len(streams)][0]
        q = type(p).from_buffer_copy(p)
        q.n_frames = 1 + K * G
        q.seed = p.seed + 1000003 * (j // len(streams))
        longer.append((q,) + m.generate_clip(q))
    streams = longer
    nv = n_clips * K

    def pack(f0, k):
        bufs = [streams[c % len(streams)][1][streams[c % len(streams)][2][f0 + j]:streams[c % len(streams)][2][f0 + j + 1]] for j in range(k) for c in range(n_clips)]
        return bufs, (C.c_void_p * len(bufs))(*[x.ctypes.data for x in bufs]), (C.c_size_t * len(bufs))(*[x.size for x in bufs])

    iframe = pack(0, 1)
    packs = [pack(1 + K * g, K) for g in range(G)]
    offs, outo, rcs = (C.c_int32 * nv)(), (C.c_int32 * nv)(), (C.c_int * nv)()

    def zero():
        C.memset(offs, 0, C.sizeof(offs))

    b = m.MobiclipBatch(n_clips, W, H, version, device=device, device_parse=device_parse)
    lib, h = b._lib, b._h
    zero()
    assert lib.mobi_batch_decode_gop(h, 1, iframe[1], iframe[2], offs, rcs) == 0 and not any(rcs[:n_clips])
    ms = []
    for g in range(min(G, 4)):  # the first group untimed (allocations)
        zero()
        t0 = _t.perf_counter()
        assert lib.mobi_batch_decode_gop(h, K, packs[g][1], packs[g][2], offs, rcs) == 0
        ms.append((_t.perf_counter() - t0) * 1e3 / K)
        assert not np.frombuffer(rcs, dtype=np.int32).any(), "stream error in the group leg"
    host_clips = b.host_clips()
    verified = verify_clips(b, streams, len(streams), n_clips, K * min(G, 4), W, H)
    b.close()
    t = float(np.median(ms[1:]))
    out = {"value": round(n_clips * W * H / t / 1e3, 1), "unit": "Mpixels/s", "ms_per_step": round(t, 3), "clips": n_clips, "frames_per_group": K,
           "how": f"mobi_batch_decode_gop: {K} consecutive P-frames of every clip per call, parsed side by side as {nv} virtual clips (mobi_gop.h), reconstructed as {K} steps; "
                  "wall time of the call / frames (host staging, H2D, parse, chain check, reconstruction, read-back, sync)",
           "clips_handed_to_the_host_parser": int(host_clips), "verified": verified}
    # Pipelined: what is PARSED side by side is not bound by the ring -- mobi_batch_gop_begin takes up to 32 frames, mobi_batch_gop_finish hands
    # them out six at a time -- so a batch too small to fill the parsers' lanes with six frames per clip begins more (Kp, above: 4096 clips x 32).
    Gp = G * K // Kp
    del packs
    packs = [pack(1 + Kp * g, Kp) for g in range(Gp)]
    nvp = n_clips * Kp
    offs, outo, rcs = (C.c_int32 * nvp)(), (C.c_int32 * nvp)(), (C.c_int * nvp)()
    b = m.MobiclipBatch(n_clips, W, H, version, device=device, device_parse=device_parse)
    lib, h = b._lib, b._h
    assert lib.mobi_batch_decode_gop(h, 1, iframe[1], iframe[2], offs, rcs) == 0
    C.memset(offs, 0, C.sizeof(offs))

    rc_view = np.frombuffer(rcs, dtype=np.int32)  # (looked at through numpy: a Python loop over 122 880 ctypes ints is 10 ms of the caller's turn)

    def finish():  # the oldest group, six frames per call
        pending = lib.mobi_batch_gop_frames_pending(h)
        assert pending == Kp
        while pending > 0:
            part = min(6, pending)
            assert lib.mobi_batch_gop_finish(h, outo, rcs) == 0 and not rc_view[:part * n_clips].any(), "stream error in the pipelined group leg"
            pending -= part

    # groups 0 and 1 begun and group 0 finished untimed: both slots' buffers exist before the clock starts
    assert lib.mobi_batch_gop_begin(h, Kp, packs[0][1], packs[0][2], offs) == 0
    assert lib.mobi_batch_gop_begin(h, Kp, packs[1][1], packs[1][2], offs) == 0
    finish()
    t0 = _t.perf_counter()
    for g in range(2, Gp):
        assert lib.mobi_batch_gop_begin(h, Kp, packs[g][1], packs[g][2], offs) == 0
        finish()
    tp = (_t.perf_counter() - t0) * 1e3 / ((Gp - 2) * Kp)
    finish()
    verified_p = verify_clips(b, streams, len(streams), n_clips, Kp * Gp, W, H)
    b.close()
    out["pipelined"] = {"value": round(n_clips * W * H / tp / 1e3, 1), "unit": "Mpixels/s", "ms_per_step": round(tp, 3), "groups_timed": Gp - 2, "frames_per_group": Kp,
                        "how": f"mobi_batch_gop_begin of group g + 1 ({Kp} frames of every clip: gather, upload) before the mobi_batch_gop_finish calls of group g (hand-overs, "
                               "reconstruction six frames per call, the parse of group g + 1 behind the last part's); wall time per frame step in the steady state (one group's parse is "
                               "under way when the clock starts and one when it stops)",
                        "verified": verified_p}
    return out


def single_stream_leg(m, stream, W, H, version, device):
    """What the drop-in boundary replaces is ONE MobiclipDecoder used by one thread (MobiConverter/Program.cs:57-71, Form1.cs:199-215):
    d.Data = frame; d.DecodeFrame().  One clip through mobi_decode (+ mobi_get_argb, the Bitmap DecodeFrame() returns), wall time per
    call, next to the oracle's time per frame on one host thread for the same stream.  Latency of a 1200-macroblock frame on a 256-CU part,
    not throughput: reported beside the headline value, never as it."""
    import ctypes as C
    from tests.oracle_binding import OracleDecoder  # the checker, here only as the reported CPU figure beside it
    p, data, fo = stream
    lib = m.decoder.load_library()
    lib.mobi_create.restype = C.c_void_p
    lib.mobi_create.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int]
    lib.mobi_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int32)]
    lib.mobi_get_argb.argtypes = [C.c_void_p, C.c_void_p]
    lib.mobi_destroy.argtypes = [C.c_void_p]
    argb = np.empty(W * H, np.uint32)
    res = {}
    for with_bitmap in (False, True):
        t_i, t_p = [], []
        for rep in range(10):  # the first pass pays the allocations; nine timed (a clip holds one I-frame: nine samples of it)
            h = lib.mobi_create(W, H, int(version), device)
            assert h
            for f in range(p.n_frames):
                buf = data[fo[f]:fo[f + 1]]
                off = C.c_int32(0)
                t0 = time.perf_counter()
                rc = lib.mobi_decode(h, buf.ctypes.data, buf.size, C.byref(off))
                if with_bitmap:
                    rc |= lib.mobi_get_argb(h, argb.ctypes.data)
                dt = (time.perf_counter() - t0) * 1e3
                assert rc == 0
                if rep:
                    (t_i if f == 0 else t_p).append(dt)
            lib.mobi_destroy(h)
        res["with_bitmap" if with_bitmap else "planes"] = {"p_frame_ms": round(float(np.median(t_p)), 4), "i_frame_ms": round(float(np.median(t_i)), 4),
                                                              "p_frame_ms_max": round(float(np.max(t_p)), 4), "i_frame_ms_max": round(float(np.max(t_i)), 4), "i_frame_samples": len(t_i)}
    o = OracleDecoder(W, H, p.version)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 1.0:
        assert o.decode_clip(data, fo, False) == p.n_frames
        n += 1
    cpu_ms = (time.perf_counter() - t0) * 1e3 / (n * p.n_frames)
    o.close()
    res.update({"workload": f"one {W}x{H} clip, {p.n_frames} frames, mobi_create / mobi_decode per frame (host parse, upload, one launch for a P-frame's step, sync)",
                "value": round(W * H / res["planes"]["p_frame_ms"] / 1e3, 1), "unit": "Mpixels/s (P-frames, planes)",
                "oracle_ms_per_frame_1_thread": round(cpu_ms, 4)})
    return res


def kernels_sha16():
    """identity of the reconstruction kernels a counter profile was taken with: sha256 of their sources"""
    import hashlib
    hsh = hashlib.sha256()
    for f in ("mobi_kernels.hip", "mobi_tile.h", "mobi_cmd.h", "mobi_recon_math.h"):
        hsh.update(open(os.path.join(ROOT, "mobiclipdecoder_amd", "csrc", f), "rb").read())
    return hsh.hexdigest()[:16]


def config4_leg(m, streams, W, H, version, device, n_clips, n_steps):
    """BASELINE config 4 ("64 clips sharded across 8 GPUs") as seen by ONE GPU: its share of 8 clips, replayed like the big batch.
    150 octet waves per clip fill a few per cent of the chip, so this is the latency of one short launch per step (r04: mobi_recon_step), not a
    throughput figure; it is reported next to the headline value, never as it."""
    b = m.MobiclipBatch(n_clips, W, H, version, device=device)
    for c in range(n_clips):
        p, data, fo = streams[c % len(streams)]
        assert all(r == 0 for r in b.preload(c, data, fo))
    b.commit()
    b.replay(0)
    for f in range(1, 5):
        b.replay(f)
    assert b.sync() == 0
    b.set_kernel_timing(0)
    t0 = time.perf_counter()
    b.time_begin()
    for i in range(n_steps):
        b.replay(5 + i)  # frames 5 .. 5 + n_steps - 1 <= 32: stream order
    stream_ms = b.time_end()
    assert b.sync() == 0
    wall = time.perf_counter() - t0
    b.close()
    return {"workload": f"{n_clips} clips of {W}x{H} on this GPU = the per-GPU share of 64 clips over 8 GPUs", "steps": n_steps,
            "ms_per_step": round(stream_ms / n_steps, 4), "value": round(n_clips * n_steps * W * H / (stream_ms * 1e-3) / 1e6, 1), "unit": "Mpixels/s",
            "wall_ms_per_step": round(wall * 1e3 / n_steps, 4),
            "note": "one launch per step (mobi_recon_step: 1200 octet waves of inter macroblocks and ~480 intra waves, one macroblock each at this size, which wait for the tags of the macroblocks their halo reads): launch latency and two or three dependency levels, not bandwidth"}


def bitmap_leg(m, streams, W, H, version, device, n_clips):
    """Row f1: the Bitmap DecodeFrame() returns (MD.cs:260-323) for every clip of a resident batch, mobi_yuv_to_argb alone: 1.5 bytes read
    and 4 written per pixel against the HBM roofline.  Beside the headline, never in it (the headline metric is planes)."""
    b = m.MobiclipBatch(n_clips, W, H, version, device=device)
    for c in range(n_clips):
        if c < len(streams):
            p, data, fo = streams[c]
            assert all(r == 0 for r in b.preload(c, data, fo))
        else:
            b.preload_clone(c, c % len(streams))
    b.commit()
    for f in range(3):
        b.replay(f)
    assert b.sync() == 0
    for _ in range(3):
        b.convert_argb()
    assert b.sync() == 0
    n = 20
    b.time_begin()
    for _ in range(n):
        b.convert_argb()
    ms = b.time_end() / n
    got = b.bitmap(n_clips - 1)
    from tests.oracle_binding import OracleDecoder  # (the checker)
    p, data, fo = streams[(n_clips - 1) % len(streams)]
    o = OracleDecoder(W, H, version)
    for f in range(3):
        o.Data, o.Offset = data, int(fo[f])
        assert o.DecodeFrame() is not None
    ok = bool((got == o.argb()).all())
    b.close()
    byts = n_clips * W * H * 5.5
    return {"kernel": "mobi_yuv_to_argb", "clips": n_clips, "ms": round(ms, 4), "value": round(n_clips * W * H / (ms * 1e-3) / 1e6, 1), "unit": "Mpixels/s",
            "roofline": {"bound": "hbm", "achieved": round(byts / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "algorithmic_bytes_per_launch": int(byts)},
            "verified": {"clip": n_clips - 1, "ok": ok, "against": "CPU oracle's ARGB of the same frame"}}


def rank_host_setup(rank, local, world, probe_gpu=True):
    """One rank per GPU on ONE host: the ranks share its cores and its memory controllers.  Each rank takes cpus / world parse threads
    (MOBI_PARSE_THREADS, unless the caller set it) and runs -- pool threads, pinned staging buffers (first touch) and all -- on the cpus of
    its GPU's NUMA node when sysfs names one, else on its 1 / world share of the cpus this process may use.  Returns what it did, for the line."""
    cpus = sorted(os.sched_getaffinity(0))
    info = {"cpus_visible": len(cpus), "numa_node": None}
    if world > 1:
        mine, node = None, None
        if probe_gpu:
            try:
                import torch
                pr = torch.cuda.get_device_properties(local)
                bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
                node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
                if node >= 0:
                    txt = open(f"/sys/devices/system/node/node{node}/cpulist").read().strip()
                    on_node = set()
                    for part in txt.split(","):
                        a, _, b = part.partition("-")
                        on_node.update(range(int(a), int(b or a) + 1))
                    sharing = max(1, world // max(1, len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])))
                    cand = sorted(on_node & set(cpus))
                    if cand:  # the ranks of one node split its cpus among themselves (by local rank order)
                        k = local % sharing
                        mine = cand[k * len(cand) // sharing:(k + 1) * len(cand) // sharing] or cand
                        info["numa_node"] = node
            except Exception as e:  # no sysfs entry, an older torch: the even split below
                info["numa_probe"] = f"{type(e).__name__}: {e}"
        if mine is None:
            mine = cpus[rank * len(cpus) // world:(rank + 1) * len(cpus) // world] or cpus
        try:
            os.sched_setaffinity(0, mine)
        except OSError as e:
            info["affinity_error"] = str(e)
        info["cpus_pinned"] = len(mine)
        # (the library's own rule: a thread per two hardware threads, 64 at most -- applied to this rank's share)
        os.environ.setdefault("MOBI_PARSE_THREADS", str(max(2, min(64, len(mine) // 2))))
    info["parse_threads"] = int(os.environ["MOBI_PARSE_THREADS"]) if "MOBI_PARSE_THREADS" in os.environ else None
    return info


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without an external launcher: start one worker per GPU (LOCAL_RANK = GPU index), each a copy of
    this command with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set -- exactly the environment `torch.distributed.run` would give
    them -- and wait for all.  Rank 0 prints the one JSON line on the stdout this process hands down.  There is no reference
    analogue beyond "independent decoder instances, one thread per open file" (MobiclipDecoder/Form1.cs:199-215)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [p.wait() for p in procs]
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc]
    if bad:
        raise SystemExit(f"bench.py: ranks failed: {bad}")


def verify_clips(b, streams, distinct, clips, frame, W, H):
    """After the timed region: EVERY clip's newest frame as it sits in HBM.  Clip c decodes the stream of clip c mod `distinct`: the
    `distinct` source clips are read back and compared with the oracle's frame at that stream position, every other clip is compared with
    its source clip on the device, byte for byte (mobi_batch_compare_clips) -- the bench line says whether the pixels it counted were the
    right ones, all of them.  (The checker, used as the checker.)"""
    from tests.oracle_binding import OracleDecoder
    from concurrent.futures import ThreadPoolExecutor
    n_src = min(distinct, clips)

    def oracle_frame(sidx):  # (the C oracle runs outside the GIL: the sources side by side on a few host threads)
        p, data, fo = streams[sidx]
        o = OracleDecoder(W, H, p.version)
        for f in range(frame + 1):
            o.Data, o.Offset = data[fo[f]:fo[f + 1]], 0
            assert o.DecodeFrame() is not None
        want = (np.array(o.y(0)[:, :W], copy=True), np.array(o.uv(0), copy=True))
        o.close()
        return want

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        wants = list(pool.map(oracle_frame, range(n_src)))
    bad_sources = []
    for sidx in range(n_src):
        got = b.planes(sidx)
        if got is None or not np.array_equal(got[0][:, :W], wants[sidx][0]) or not np.array_equal(got[1], wants[sidx][1]):
            bad_sources.append(sidx)
    differing_copies = b.compare_clips(n_src)
    return {"clips": int(clips), "frame": int(frame), "ok": not bad_sources and differing_copies == 0,
            "sources_against_oracle": n_src, "sources_that_differ": bad_sources,
            "copies_against_their_source_on_device": int(clips - n_src), "copies_that_differ": int(differing_copies),
            "against": "CPU oracle (Y and UV planes of ring slot 0 of the distinct source clips); every other clip byte for byte against its source clip"}


def content_leg(m, sharding, config, rank, local, clips, distinct, overrides, chains=2):
    """The same batch on OTHER content: generator overrides (e.g. lowfreq_prob=700: 70 % of the coded blocks carry one or two levels within
    the three lowest scan positions -- what DC-dominated video looks like, and what the reference's IDCT1P / IDCT3P classes take,
    MD.cs:2939-2940; the headline mix, 1..6 levels uniformly over 16 positions, is the transforms' worst case).  Reported BESIDE the
    headline value, never as it: `chains` x 32 P-frames in stream order, the I-frames outside the timed region."""
    streams = []
    for i in range(distinct):
        p = m.default_params(config, sharding.stream_seed(config, rank, i), n_frames=1 + N_PFRAMES, **overrides)
        streams.append((p,) + m.generate_clip(p))
    p0 = streams[0][0]
    W, H = p0.width, p0.height
    b = m.MobiclipBatch(clips, W, H, p0.version, device=local)
    for i, (p, data, fo) in enumerate(streams):
        assert all(r == 0 for r in b.preload(i, data, fo))
    for c in range(distinct, clips):
        b.preload_clone(c, c % distinct)
    b.commit()
    b.set_kernel_timing(0)
    b.replay(0)
    for f in range(1, 9):  # warm-up
        b.replay(f)
    assert b.sync() == 0
    stream_ms, acc = 0.0, {"inter_ms": 0.0, "intra_ms": 0.0, "inter_launches": 0, "intra_launches": 0}
    for _ in range(chains):
        b.set_kernel_timing(0)
        b.replay(0)
        assert b.sync() == 0
        b.set_kernel_timing(2)
        b.time_begin()
        for f in range(1, 1 + N_PFRAMES):
            b.replay(f)
        stream_ms += b.time_end()
        km = b.kernel_ms()
        for k in acc:
            acc[k] += km[k]
        assert b.sync() == 0
    steps = chains * N_PFRAMES
    frames = list(range(1, 1 + N_PFRAMES))
    cmd = sum(b.cmd_bytes(f) for f in frames) / N_PFRAMES
    st = [b.intra_stats(f) for f in frames]
    n_intra, intra_cmd = sum(x[0] for x in st) / N_PFRAMES, sum(x[1] for x in st) / N_PFRAMES
    verified = verify_clips(b, streams, distinct, clips, N_PFRAMES, W, H)
    b.close()
    n_mbs = (W // 16) * (H // 16)
    step_ms, inter_ms = stream_ms / steps, acc["inter_ms"] / max(1, acc["inter_launches"])
    algo = (clips * n_mbs - n_intra) * 768.0 + (cmd - intra_cmd)
    return {"generator_overrides": overrides, "clips": clips, "steps": steps, "ms_per_step": round(step_ms, 4),
            "value": round(clips * W * H / step_ms / 1e3, 1), "unit": "Mpixels/s",
            "inter_kernel_ms": round(inter_ms, 4), "inter_frac": round(algo / (inter_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "whole_step_frac": round((clips * 3.0 * W * H + cmd) / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "command_bytes_per_frame": round(cmd / clips, 1), "timed_region_s": round(stream_ms * 1e-3, 3), "verified": verified}


def dry_run(args, rank, world):
    """The N-rank plumbing with no HIP call: generate this rank's streams, build the process group, barrier, MAX-reduce a made-up
    time, gather the seeds, print the one line.  What can be wrong without a GPU is exactly this (tests/test_bench_launcher.py)."""
    import torch.distributed as dist
    import mobiclipdecoder_amd as m
    from mobiclipdecoder_amd import sharding
    seeds = [sharding.stream_seed(args.config, rank, i) for i in range(max(1, min(args.distinct, 2)))]
    sizes = []
    for sd in seeds:
        p = m.default_params(args.config, sd, n_frames=2, width=64, height=48)
        sizes.append(int(m.generate_clip(p)[0].size))
    group = world > 1
    if group:
        dist.init_process_group("gloo")
        dist.barrier()
    elapsed = sharding.max_over_ranks(dist if group else None, 1.0 + rank)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    host = rank_host_setup(rank, local, world, probe_gpu=False)
    mine = {"rank": rank, "local_rank": local, "device": local, "seeds": seeds, "bytes": sizes, "clips_per_gpu": int(args.clips), "verified_ok": None, "host": host,
            # world > 1: the product path too, all ranks at once (bitstreams in host memory -> planes, frame-parallel groups); None here: no HIP call
            "end_to_end_groups": None}
    gathered = [mine]
    if group:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "decoded Mpixels/s @ 640x480 P-frames", "value": 0.0, "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "dry_run": True, "elapsed_max_s": elapsed, "ranks": gathered, "scaling": "weak",
                          "note": "no HIP call was made: rank plumbing only"}), flush=True)
    if group:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=192, help="timed P-frame steps (default: six 32-frame P-chains = ~2 s)")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--clips", type=int, default=24576, help="independent clips resident per GPU (24576 x 640x480 = 109 GB of rings + 74 GB of command lists; halved if it does not fit)")
    ap.add_argument("--distinct", type=int, default=16, help="distinct generated streams per GPU (others are private HBM copies)")
    ap.add_argument("--config", default="B", choices=["A", "B", "C"])
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--e2e-clips", type=int, default=4096, help="clips of the end-to-end leg (bitstream in, device-side parse); 0 = skip")
    ap.add_argument("--e2e-steps", type=int, default=12)
    ap.add_argument("--e2e-large-clips", type=int, default=24576, help="clips of the second end-to-end leg, at the headline batch size with the lock-step parser in front; 0 = skip")
    ap.add_argument("--e2e-xl-clips", type=int, default=49152, help="clips of the third end-to-end leg: what 288 GB of HBM hold at 640x480 (40960 or 32768 if that does not fit); 0 = skip")
    ap.add_argument("--bitmap-clips", type=int, default=512, help="clips of the Bitmap leg (row f1: mobi_yuv_to_argb on a resident batch); 0 = skip")
    ap.add_argument("--config4-clips", type=int, default=8, help="clips of the config-4 leg (64 clips / 8 GPUs); 0 = skip")
    ap.add_argument("--single-stream", type=int, default=1, help="1: time one clip through mobi_decode / mobi_get_argb (the boundary's own shape); 0 = skip")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip per-launch HIP events (roofline becomes null)")
    ap.add_argument("--content-lowfreq", type=int, default=700, help="second content profile beside the headline: per mille of coded blocks that carry only 1..2 levels in the three lowest scan positions; 0 = skip")
    ap.add_argument("--dry-run", action="store_true", help="rank plumbing only: streams, sharding, barrier, reduction, the one line; no HIP call")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under a launcher: be one (the driver's 1-GPU command is `python3 bench.py --gpus 1 ...`; its N-GPU one may have that shape too)
        if not args.dry_run:
            import torch
            if torch.cuda.device_count() < args.gpus:
                raise SystemExit(f"bench.py: --gpus {args.gpus} but this node shows {torch.cuda.device_count()} HIP device(s)")
        launch_ranks(args.gpus, sys.argv[1:])
        return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU, and the line reports n_gpus = ranks")
    if args.dry_run:
        return dry_run(args, rank, world)
    import torch
    host_info = rank_host_setup(rank, local, world)  # (before the library creates its parse threads and pinned buffers)
    dist = None
    if world > 1:
        # no data-path collective exists (north_star: "no RCCL"): the process group only carries the barrier and the max of the
        # elapsed times, so it is a gloo group (TCP between the ranks of the node), not an RCCL one
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("gloo")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: there is no CPU reconstruction path to time")
    if local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} (LOCAL_RANK {local}) has no GPU: {torch.cuda.device_count()} HIP device(s) visible")

    import mobiclipdecoder_amd as m
    from mobiclipdecoder_amd import sharding

    # experiment hook: BENCH_GEN="pm_split1=0,pm_deep=0" overrides generator fields (non-default => not the headline workload)
    gen_over = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("BENCH_GEN", "").split(",") if kv)}
    distinct = max(1, min(args.distinct, args.clips))
    streams = []
    for i in range(distinct):
        p = m.default_params(args.config, sharding.stream_seed(args.config, rank, i), n_frames=1 + N_PFRAMES, **gen_over)
        streams.append((p,) + m.generate_clip(p))
    p0 = streams[0][0]
    W, H = p0.width, p0.height

    # the batch: as many clips as asked for; if a GPU cannot hold them (rings + command lists), halve until it can -- and every
    # rank settles on the same size (weak scaling: per-GPU work is fixed and equal)
    def build_batch(clips):
        b = m.MobiclipBatch(clips, W, H, p0.version, device=local)
        try:
            for i, (p, data, fo) in enumerate(streams):
                rcs = b.preload(i, data, fo)
                assert all(r == 0 for r in rcs), rcs
            for c in range(distinct, clips):
                b.preload_clone(c, c % distinct)
            b.commit()
            b.replay(0)  # the I-frame: the first replay allocates what is still missing, so it belongs to the "does it fit" test
            assert b.sync() == 0
        except m.MobiclipError:
            b.close()
            raise
        return b

    clips, b = args.clips, None
    while b is None:
        try:
            b = build_batch(clips)
        except m.MobiclipError as e:
            if clips <= 512:
                raise
            # three quarters, then half, of the size asked for, and so on down (24576 -> 18432 -> 12288 -> 9216 ...)
            nxt = clips * 3 // 4 if (clips & (clips - 1)) == 0 or clips % 3 else clips * 2 // 3
            print(f"bench.py: rank {rank}: {clips} clips do not fit ({e}); retrying with {nxt}", file=sys.stderr, flush=True)
            clips = nxt
    agreed = -int(sharding.max_over_ranks(dist, -clips))  # the smallest size any rank settled on
    if agreed != clips:
        b.close()
        clips = agreed
        b = build_batch(clips)
    args.clips = clips

    # Stream order.  The generated clips are 1 I-frame + 32 P-frames; a step s of the run decodes P-frame 1 + (s mod 32), and every
    # time the chain wraps the I-frame is decoded again first, so that each P-frame finds the ring history it was coded against
    # (r01 wrapped straight from frame 32 to frame 1: the same work on a wrong history).  The I-frame is not a P-frame step: it
    # runs OUTSIDE the timed region, which is therefore a sum of segments of up to 32 consecutive steps, each bracketed by the
    # barrier + synchronize pair; `ms_per_step` is that sum over `steps`.
    total = args.warmup + args.steps
    # A small step goes out as ONE launch (mobi_recon_step) and has no kernel of its own to time: the first warm-up step is run with the
    # per-launch events on to see which kind this batch gets; a one-launch batch then runs its timed region without them (two event
    # records per step are a fifth of a 40 us step) and its roofline is the whole step's.
    b.set_kernel_timing(2)
    one_launch = False
    s = 0
    while s < args.warmup:  # (the I-frame, frame 0, ran above)
        if s and s % N_PFRAMES == 0:
            b.replay(0)
        if s == 0:
            b.time_begin()  # (resets the per-launch accumulators)
        b.replay(1 + s % N_PFRAMES)
        if s == 0:
            b.time_end()
            km0 = b.kernel_ms()
            one_launch = km0["inter_launches"] >= 1 and km0["intra_launches"] == 0 and b.intra_stats(1)[0] > 0
            b.set_kernel_timing(0)
        s += 1
    assert b.sync() == 0, "clamp-domain fault during warm-up"

    kt = 0 if (args.no_kernel_events or one_launch) else 2
    b.set_kernel_timing(kt)
    elapsed, stream_ms, timed_frames, iframe_ms = 0.0, 0.0, [], []
    acc = {"inter_ms": 0.0, "intra_ms": 0.0, "inter_launches": 0, "intra_launches": 0}
    while s < total:
        if s % N_PFRAMES == 0:  # chain wrap: re-seed the ring, outside the timed region (its own time is reported as iframe_step_ms)
            b.set_kernel_timing(0)
            b.time_begin()
            b.replay(0)
            iframe_ms.append(b.time_end())
            assert b.sync() == 0
            b.set_kernel_timing(kt)
        seg = min(total - s, N_PFRAMES - s % N_PFRAMES)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        b.time_begin()
        t0 = time.perf_counter()
        for k in range(seg):
            b.replay(1 + (s + k) % N_PFRAMES)
        stream_ms += b.time_end()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed += time.perf_counter() - t0
        km = b.kernel_ms()
        for key in acc:
            acc[key] += km[key]
        timed_frames += [1 + (s + k) % N_PFRAMES for k in range(seg)]
        s += seg
        assert b.sync() == 0, "clamp-domain fault in the timed region"
    km = acc

    my_elapsed = elapsed
    elapsed = sharding.max_over_ranks(dist, elapsed)
    per_rank = [my_elapsed]
    steps = args.steps
    # every rank checks its own clips (all of them: verify_clips) and says how many it ran: a rank that silently ran half the batch, or the
    # wrong pictures, shows in the one line
    verified = verify_clips(b, streams, distinct, args.clips, timed_frames[-1], W, H) if timed_frames else None
    rank_report = [{"rank": rank, "device": local, "clips_per_gpu": int(args.clips), "verified_ok": bool(verified and verified["ok"]), "ms_per_step": round(my_elapsed * 1e3 / steps, 4), "host": host_info}]
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, my_elapsed)
        rank_report = [None] * world
        dist.all_gather_object(rank_report, {"rank": rank, "device": local, "clips_per_gpu": int(args.clips), "verified_ok": bool(verified and verified["ok"]),
                                             "ms_per_step": round(my_elapsed * 1e3 / steps, 4), "host": host_info})
    cmd_bytes = sum(b.cmd_bytes(f) for f in timed_frames) / steps          # per step, all clips of this GPU
    stats = [b.intra_stats(f) for f in timed_frames]
    n_intra = sum(x[0] for x in stats) / steps
    intra_cmd = sum(x[1] for x in stats) / steps
    b.close()
    e2e = e2e_large = c4 = single = None
    e2e_all = None
    if world > 1 and args.e2e_clips > 0 and args.config == "B":
        # N > 1: the PRODUCT path on every rank at once (the legs above only run at N = 1, so host staging x N ranks, N x PCIe and
        # N parse pools on one host went unmeasured): frame-parallel groups at the headline batch, each rank on its own streams and cpus,
        # behind the timed region's last barrier; per rank in ranks[], and the slowest rank's time for the whole job's rate.
        n_e2e = min(args.e2e_large_clips, args.clips)
        dist.barrier()
        try:
            mine = gop_leg(m, streams, W, H, p0.version, local, n_e2e, "lockstep")
        except Exception as e:
            mine = {"error": f"{type(e).__name__}: {e}"}
        every = [None] * world
        dist.all_gather_object(every, mine)
        if rank == 0:
            for r, g in enumerate(every):
                rank_report[r]["end_to_end_groups"] = g if "error" in g else {"ms_per_step": g["ms_per_step"], "pipelined_ms_per_step": g["pipelined"]["ms_per_step"], "clips": g["clips"],
                                                                                "frames_per_group": g["frames_per_group"], "verified_ok": bool(g["verified"]["ok"] and g["pipelined"]["verified"]["ok"])}
            if all("error" not in g for g in every):
                slow, slow_p = max(g["ms_per_step"] for g in every), max(g["pipelined"]["ms_per_step"] for g in every)
                e2e_all = {"value": round(world * n_e2e * W * H / slow / 1e3, 1), "unit": "Mpixels/s", "ms_per_step": slow, "n_gpus": world, "clips_per_gpu": n_e2e,
                           "pipelined": {"value": round(world * n_e2e * W * H / slow_p / 1e3, 1), "ms_per_step": slow_p},
                           "how": "every rank at once: mobi_batch_decode_gop / gop_begin + gop_finish on its own clips (bench.py gop_leg); the whole job's pixels per step / the slowest rank's ms per step"}
            else:
                e2e_all = {"error": [g.get("error") for g in every]}
    if world == 1 and args.single_stream and args.config == "B":
        single = single_stream_leg(m, streams[0], W, H, p0.version, local)
    if world == 1 and args.e2e_clips > 0 and args.config == "B":
        e2e = end_to_end(m, streams, W, H, p0.version, local, args.e2e_clips, args.e2e_steps)
    if world == 1 and args.e2e_clips > 0 and min(args.e2e_large_clips, args.clips) >= 8192 and args.config == "B":  # (--e2e-clips 0 skips both legs)
        try:
            # The lock-step parser's lanes (32 clips per wave) run until the slowest is done and every round costs what the lanes' different states need:
            # a wave of 4 x 16 copies is a quarter as diverse as a wave of 64 clips.  r04 measured 24.3 ms per step with 16 distinct streams
            # and 34.4 with 64 or 128 (tools/exp_dparse.py, DISTINCT=...): this leg takes 64, so that no wave holds two copies of one.
            wide = list(streams)
            for i in range(len(streams), 64):
                p = m.default_params(args.config, sharding.stream_seed(args.config, rank, i), n_frames=12, **gen_over)
                wide.append((p,) + m.generate_clip(p))
            e2e_large = end_to_end(m, wide, W, H, p0.version, local, min(args.e2e_large_clips, args.clips), 6, device_parse="lockstep")
        except Exception as e:  # (e.g. does not fit beside what the allocator still holds: reported beside the headline value, not fatal to it)
            e2e_large = {"error": f"{type(e).__name__}: {e}"}
    # The lock-step parser's cost per clip falls with the batch (more clips per wave share every round of the walk): the same leg once more
    # with as many clips as the device holds.  Skipped rather than tried when the memory is visibly not there (a failed allocation of tens
    # of GB is not a cheap way to find out).
    e2e_xl = None
    if world == 1 and args.e2e_xl_clips > max(args.e2e_large_clips, 0) and isinstance(e2e_large, dict) and "error" not in e2e_large and args.config == "B":
        for n_xl in sorted({args.e2e_xl_clips, min(args.e2e_xl_clips, 40960), min(args.e2e_xl_clips, 32768)}, reverse=True):
            free = torch.cuda.mem_get_info(local)[0]
            if free < n_xl * 4.9e6 + 8e9:  # rings 4.42 MB per clip, two steps' command lists, staging
                e2e_xl = {"error": f"{n_xl} clips need ~{(n_xl * 4.9e6 + 8e9) / 1e9:.0f} GB, {free / 1e9:.0f} GB free"}
                continue
            try:
                e2e_xl = end_to_end(m, wide, W, H, p0.version, local, n_xl, 6, device_parse="lockstep", groups=False)  # (rings of 218 GB leave no room for a second set of command lists)
                break
            except Exception as e:
                e2e_xl = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and args.config4_clips > 0 and args.config == "B":
        c4 = config4_leg(m, streams, W, H, p0.version, local, args.config4_clips, 24)
    bitmap = None
    if world == 1 and args.bitmap_clips > 0 and args.config == "B":
        try:
            bitmap = bitmap_leg(m, streams, W, H, p0.version, local, args.bitmap_clips)
        except Exception as e:
            bitmap = {"error": f"{type(e).__name__}: {e}"}
    content = None
    if world == 1 and args.content_lowfreq > 0 and not gen_over:
        try:
            content = content_leg(m, sharding, args.config, rank, local, args.clips, distinct, {"lowfreq_prob": args.content_lowfreq})
        except Exception as e:
            content = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        n_mbs = (W // 16) * (H // 16)
        step_bytes = args.clips * 3.0 * W * H + cmd_bytes                 # whole step: ref read 1.5WH + write 1.5WH + every command byte
        # the dominant kernel, mobi_recon_inter8: the inter macroblocks only (384 B read + 384 B written each) and their commands;
        # the intra macroblocks' pixels and records belong to mobi_recon_intra
        algo_bytes = (args.clips * n_mbs - n_intra) * 768.0 + (cmd_bytes - intra_cmd)
        roof = None
        if one_launch:
            step_ms = stream_ms / steps
            ach = step_bytes / (step_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "mobi_recon_step", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes_per_launch": int(step_bytes),
                    "avg_launch_ms": round(step_ms, 5), "launches": steps, "whole_step_frac": round(ach / HBM_PEAK_GBS, 4),
                    "whole_step_bytes": int(step_bytes), "whole_step_ms": round(step_ms, 5), "intra_macroblocks_per_step": round(n_intra, 1),
                    "note": "a step this small is one launch (octets of inter macroblocks and intra fours side by side): bytes of the whole step over the "
                            "stream time per step, launch gaps included; latency-bound at this size, not a bandwidth figure"}
        elif km["inter_launches"]:
            avg_ms = km["inter_ms"] / km["inter_launches"]
            ach = algo_bytes / (avg_ms * 1e-3) / 1e9
            step_ms = stream_ms / steps
            roof = {"bound": "hbm", "kernel": "mobi_recon_inter8", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                    "algorithmic_bytes_per_launch": int(algo_bytes), "avg_launch_ms": round(avg_ms, 5),
                    "launches": km["inter_launches"],
                    # pixel bytes only (the inter macroblocks' 384 B read + 384 B written, no command bytes): a fatter command list cannot flatter it
                    "frac_pixel_bytes": round((args.clips * n_mbs - n_intra) * 768.0 / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "command_kb_per_frame": round(cmd_bytes / args.clips / 1e3, 2),
                    "whole_step_frac": round(step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "whole_step_bytes": int(step_bytes), "whole_step_ms": round(step_ms, 5),
                    "intra_macroblocks_per_step": round(n_intra, 1),
                    "intra_kernel_ms_per_step": round(km["intra_ms"] / steps, 5) if km["intra_launches"] else None,
                    "intra_launches_per_step": round(km["intra_launches"] / steps, 2) if km["intra_launches"] else None,
                    # SURVEY's formula charges every macroblock a reference read (384 B); an intra macroblock has none:
                    "whole_step_frac_intra_without_reference_read": round((step_bytes - 384.0 * n_intra) / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            if km["intra_launches"]:  # mobi_recon_intra against its own bytes: SURVEY's (768 B per macroblock + commands), and what it can need (384 B written)
                ims = km["intra_ms"] / km["intra_launches"]
                roof["intra"] = {"kernel": "mobi_recon_intra", "avg_launch_ms": round(ims, 5),
                                 "algorithmic_bytes_per_launch": int(n_intra * 768.0 + intra_cmd), "frac": round((n_intra * 768.0 + intra_cmd) / (ims * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "frac_without_reference_read": round((n_intra * 384.0 + intra_cmd) / (ims * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None}
            if iframe_ms:  # an all-intra step (the I-frame that re-seeds the ring at every chain wrap), timed on its own
                ifr = float(np.median(iframe_ms))
                roof["iframe_step"] = {"ms": round(ifr, 4), "frac_of_bytes_written": round(args.clips * 1.5 * W * H / (ifr * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "steps": len(iframe_ms)}
            prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(prof):  # HBM bytes per launch from a separate rocprofv3 --pmc run (see profiles/README.md)
                try:
                    t = json.load(open(prof)).get(f"{args.config}:{args.clips}")
                    # a counter profile describes the kernels it was taken with: it is only quoted while their sources are unchanged
                    if t and t.get("kernels_sha16") == kernels_sha16() and not gen_over:
                        roof["traffic"] = t["hbm_bytes_per_launch"]
                        roof["traffic_source"] = t.get("source")
                        if "intra" in roof and t.get("intra_hbm_bytes_per_launch"):
                            roof["intra"]["traffic"] = t["intra_hbm_bytes_per_launch"]
                    elif t:
                        roof["traffic_source"] = "stale: profiles/pmc_traffic.json was taken with other kernel sources (" + str(t.get("kernels_sha16")) + ")"
                except Exception:
                    pass
        base = None
        if args.cpu_seconds > 0:  # rank 0, after the timed region's last barrier (the other ranks are done and idle)
            base = cpu_baseline(*streams[0], args.cpu_seconds)
        out = {
            "metric": "decoded Mpixels/s @ 640x480 P-frames" if args.config == "B" else f"decoded Mpixels/s P-frames (config {args.config})",
            "value": round(sharding.whole_job_mpix_per_s(world, args.clips, steps, W, H, elapsed), 1), "unit": "Mpixels/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / steps, 4),
            "ms_per_step_ranks": {"min": round(min(per_rank) * 1e3 / steps, 4), "max": round(max(per_rank) * 1e3 / steps, 4)},
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{W}x{H} {'Moflex3DS' if p0.version == 2 else 'ModsDS'} P-frame reconstruction "
                                   f"(SURVEY 8d generator mix), {args.clips} independent clips per GPU, "
                                   f"{distinct} distinct streams, command lists resident in HBM, frames in stream order "
                                   f"(the I-frame that re-seeds the ring every {N_PFRAMES} steps is outside the timed region)",
                       "generator_overrides": gen_over or None, "clips_per_gpu": args.clips, "parallelism": f"clips sharded over {world} GPU(s), no collective",
                       "stream_ms_per_step": round(stream_ms / steps, 4)},
            # the part settles at a lower clock after about a second of this load: a short timed region flatters the number
            "timed_region_s": round(elapsed, 3), "clock_state": "sustained" if elapsed >= 1.0 else "unsettled (timed region < 1 s)",
            "verified": verified, "ranks": rank_report, "roofline": roof, "cpu_baseline": base, "end_to_end": e2e, "end_to_end_large": e2e_large, "end_to_end_xl": e2e_xl, "end_to_end_all_ranks": e2e_all, "config4": c4, "single_stream": single, "bitmap": bitmap, "content_lowfreq": content,
        }
        # the end-to-end legs in one place (Mpixels/s from bitstreams in host memory; each leg's own entry says how it was measured)
        def _leg(e):
            if not e or "value" not in e:
                return None
            g = e.get("groups") or {}
            return {"clips": e.get("clips"), "call_per_frame": e.get("value"), "two_steps_in_flight": (e.get("async") or {}).get("value"),
                    "groups": g.get("value"), "groups_pipelined": (g.get("pipelined") or {}).get("value"), "frames_per_gop_begin": (g.get("pipelined") or {}).get("frames_per_group")}
        out["end_to_end_summary"] = {k: _leg(v) for k, v in (("end_to_end", e2e), ("end_to_end_large", e2e_large), ("end_to_end_xl", e2e_xl)) if _leg(v)} or None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
