#!/usr/bin/env python3
"""Scheduling of the lock-step parser's three parts (CPU, tests/tools/mobi_lsparse_host.cpp): 64 clips as the lanes of one wave, one P-frame and one
I-frame, how often each part runs under a schedule, and what that costs at the clocks measured for them on the GPU.
python tools/exp_lssched.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mobiclipdecoder_amd as m
from mobiclipdecoder_amd import sharding, build
L = C.CDLL(build.build_lshost())
L.mobi_lshost_create.restype = C.c_void_p; L.mobi_lshost_create.argtypes = [C.c_uint, C.c_uint, C.c_int]
L.mobi_lshost_destroy.argtypes = [C.c_void_p]
L.mobi_lshost_wave_sim.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
COST = dict(main=float(os.environ.get("C_MAIN", 5.5)), intra=float(os.environ.get("C_INTRA", 5.5)), cheap=float(os.environ.get("C_CHEAP", 1.6)))  # k clocks per run
streams = [m.generate_clip(m.default_params("B", sharding.stream_seed("B", 0, i), n_frames=4)) for i in range(16)]
def run(k, period, burst, kb, lanes=64):
    clips = [L.mobi_lshost_create(640, 480, 2) for _ in range(lanes)]
    out = []
    for f in range(3):
        bufs = [np.ascontiguousarray(streams[i % 16][0][streams[i % 16][1][f]:streams[i % 16][1][f + 1]]) for i in range(lanes)]
        cp = (C.c_void_p * lanes)(*clips); dp = (C.c_void_p * lanes)(*[b.ctypes.data for b in bufs]); lp = (C.c_size_t * lanes)(*[b.size for b in bufs])
        cnt = (C.c_long * 4)()
        rc = L.mobi_lshost_wave_sim(cp, dp, lp, lanes, k, period, burst, kb, cnt)
        assert rc == 0, rc
        out.append(list(cnt))
    for c in clips: L.mobi_lshost_destroy(c)
    def ms(c): return (c[1] * COST["main"] + c[2] * COST["intra"] + c[3] * COST["cheap"]) * 1e3 / 2.4e9 * 1e3
    i, p = out[0], out[2]
    print(f"k={k} intra every {period} x{burst} (+{kb} cheap): P-frame rounds {p[0]} main {p[1]} intra {p[2]} cheap {p[3]} -> {ms(p):5.1f} ms | I-frame rounds {i[0]} intra {i[2]} cheap {i[3]} -> {ms(i):5.1f} ms", flush=True)
L.mobi_lshost_wave_sched.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_void_p]
W = dict(M=float(os.environ.get("W_M", 1150)), I=float(os.environ.get("W_I", 900)), N=float(os.environ.get("W_N", 130)), T=float(os.environ.get("W_T", 150)))  # instructions per run
def sched(order, lanes=64):
    """any order of the four parts within a round; cost = instructions (vector + scalar) of the parts that ran with a lane in them"""
    clips = [L.mobi_lshost_create(640, 480, 2) for _ in range(lanes)]
    out = []
    for f in range(3):
        bufs = [np.ascontiguousarray(streams[i % 16][0][streams[i % 16][1][f]:streams[i % 16][1][f + 1]]) for i in range(lanes)]
        cp = (C.c_void_p * lanes)(*clips); dp = (C.c_void_p * lanes)(*[b.ctypes.data for b in bufs]); lp = (C.c_size_t * lanes)(*[b.size for b in bufs])
        cnt = (C.c_long * 5)()
        assert L.mobi_lshost_wave_sched(cp, dp, lp, lanes, order.encode(), cnt) == 0
        out.append(list(cnt))
    for c in clips: L.mobi_lshost_destroy(c)
    def mi(c): return (c[1] * W["M"] + c[2] * W["I"] + c[3] * W["N"] + c[4] * W["T"]) / 1e6
    i, p = out[0], out[2]
    print(f"{order:28s} P: rounds {p[0]:5d} M {p[1]:5d} I {p[2]:5d} N {p[3]:5d} T {p[4]:5d} -> {mi(p):5.2f} M instr | I-frame: rounds {i[0]:5d} I {i[2]:5d} N {i[3]:5d} T {i[4]:5d} -> {mi(i):5.2f} M", flush=True)
L.mobi_lshost_wave_greedy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
def greedy(cost, lanes=12, label=""):
    """every iteration ONE part: cost = [M, I, N, T] instructions -> the part with the most waiting lanes per instruction; cost[0] < 0: M (and I, if
    cost[1] < 0 in the same go) only when that share of the live lanes waits for them, the fuller of N / T otherwise"""
    clips = [L.mobi_lshost_create(640, 480, 2) for _ in range(lanes)]
    out = []
    for f in range(3):
        bufs = [np.ascontiguousarray(streams[i % 16][0][streams[i % 16][1][f]:streams[i % 16][1][f + 1]]) for i in range(lanes)]
        cp = (C.c_void_p * lanes)(*clips); dp = (C.c_void_p * lanes)(*[b.ctypes.data for b in bufs]); lp = (C.c_size_t * lanes)(*[b.size for b in bufs])
        cnt = (C.c_long * 5)()
        assert L.mobi_lshost_wave_greedy(cp, dp, lp, lanes, (C.c_double * 4)(*cost), cnt) == 0
        out.append(list(cnt))
    for c in clips: L.mobi_lshost_destroy(c)
    def mi(c): return (c[1] * W["M"] + c[2] * W["I"] + c[3] * W["N"] + c[4] * W["T"]) / 1e6
    i, p = out[0], out[2]
    print(f"{label:28s} P: iterations {p[0]:5d} M {p[1]:5d} I {p[2]:5d} N {p[3]:5d} T {p[4]:5d} -> {mi(p):5.2f} M instr | I-frame: {i[0]:5d} M {i[1]:5d} I {i[2]:5d} T {i[4]:5d} -> {mi(i):5.2f} M", flush=True)
if __name__ == "__main__" and "--greedy" in sys.argv:  # r05: does choosing the part by the lanes waiting for it beat the fixed round?  (12 clips per wave)
    sched("MINTNTNTNT", 12)
    greedy([W["M"], W["I"], W["N"], W["T"]], 12, "most lanes per instruction")
    for th in (0.25, 0.5, 0.75, 1.0): greedy([-th, -1, W["N"], W["T"]], 12, f"M + I when {th:.2f} of lanes wait")
    sys.exit(0)
if __name__ == "__main__" and "--orders" in sys.argv:
    for o in ["MINTNTNTNTNT", "MINTTTNTTT", "MINTTNTTNTT", "MINTTTTNTTTT", "MINTTTTTT", "MINTNTTTNTTT", "MNTINTNTTNTT", "MINTTNTTTNTTT", "MINTTTNTTTNTTT"]:
        sched(o)
    sys.exit(0)
if __name__ == "__main__":
    run(4, 1, 1, 0)
    for period, burst, kb in [(2, 1, 0), (2, 2, 2), (3, 2, 2), (4, 3, 2), (4, 4, 4), (6, 4, 4), (8, 6, 4), (8, 8, 4)]:
        run(4, period, burst, kb)
