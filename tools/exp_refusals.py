"""How often does the product refuse (MOBI_E_UNSUPPORTED) a frame the reference decodes -- and does the answer depend on where the parse runs?

  python tools/exp_refusals.py [trials]            CPU only: product parser + kernel arithmetic (tests/tools/mobi_cmd_interp.cpp) against the oracle
  python tools/exp_refusals.py [trials] --gpu      the C ABI on a GPU, once per parse mode (MOBI_DEVICE_PARSE = 0 host, 1 device, 2 hybrid, 3 lock-step)

Corrupted streams (1..7 bit flips in a rich 4-frame stream).  A refusal counts as a result difference only when the oracle decoded the same
frame without throwing; frames both decode are compared (Offset, Quantizer, planes)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mobiclipdecoder_amd import default_params, generate_clip
from mobiclipdecoder_amd.streamgen import BASE_SEED
from tests.oracle_binding import OracleDecoder

args = [a for a in sys.argv[1:] if not a.startswith("--")]
trials = int(args[0]) if args else 1500


def run(make, label):
    rng = np.random.default_rng(2026)
    frames = refused = refused_ref_ok = both_ok = both_err = differ = 0
    for trial in range(trials):
        ver = 1 + trial % 2
        p = default_params("AB"[trial % 2], BASE_SEED + 9000 + trial, n_frames=4, width=96, height=64, version=ver, pm_intra=120, pm_deep=150,
                           pm_multiref=250, qdelta_prob=250, escape_prob=80, table1_prob=400)
        data, fo = generate_clip(p)
        d = data.copy()
        for _ in range(int(rng.integers(1, 8))):
            d[int(rng.integers(0, d.size))] ^= 1 << int(rng.integers(0, 8))
        a, o = make(p), OracleDecoder(p.width, p.height, p.version)
        for f in range(p.n_frames):
            a.Data = o.Data = d[: fo[f + 1]]
            a.Offset = o.Offset = int(fo[f])
            ra, ro = a.DecodeFrame(), o.DecodeFrame()
            frames += 1
            if a.last_error == -6:
                refused += 1
                refused_ref_ok += o.last_error == 0
                break
            if a.last_error != 0 or o.last_error != 0:
                both_err += 1
                if (a.last_error == 0) != (o.last_error == 0) and a.last_error != -5:
                    differ += 1
                break
            both_ok += 1
            if a.Offset != o.Offset or a.Quantizer != o.Quantizer or not np.array_equal(ra[0], ro[0]) or not np.array_equal(ra[1], ro[1]):
                differ += 1
        if hasattr(a, "close"):
            a.close()
        o.close()
    print(f"{label}: {frames} corrupted frames: {both_ok} decoded by both, {both_err} rejected by both, {refused} refused by the product only "
          f"({refused_ref_ok} of them decoded by the reference's restatement = {100.0 * refused_ref_ok / frames:.2f} % of frames); "
          f"{differ} frames with a different result")
    return refused_ref_ok + differ


bad = 0
if "--gpu" in sys.argv:
    from mobiclipdecoder_amd import MobiclipDecoder
    for mode in "0123":
        os.environ["MOBI_DEVICE_PARSE"] = mode
        bad += run(lambda p: MobiclipDecoder(p.width, p.height, p.version), "GPU, MOBI_DEVICE_PARSE=" + mode)
    # r06: the same corrupted streams, each clip's four frames as ONE frame-parallel group (mobi_batch_decode_gop: parsed side by side from
    # predicted start states; a frame the device parsers do not finish hands the rest of the group to the host parser inside the call)
    os.environ.pop("MOBI_DEVICE_PARSE", None)
    from mobiclipdecoder_amd import MobiclipBatch

    class GroupDecoder:
        """DecodeFrame() by DecodeFrame() on top of ONE decode_gop call per clip: the first call decodes all the frames it will be asked for"""

        def __init__(self, p, frames_of):
            self.b = MobiclipBatch(1, p.width, p.height, p.version, device_parse="lockstep")
            self.frames_of, self.k, self.res, self.K = frames_of, 0, None, p.n_frames
            self.Data, self.Offset, self.last_error, self.Quantizer = None, 0, 0, 0

        def DecodeFrame(self):
            if self.res is None:
                datas, offs = self.frames_of()
                self.res = self.b.decode_gop([[d] for d in datas], [[o] for o in offs])
            k = self.k
            self.k += 1
            self.last_error, self.Offset = self.res[0][k][0], self.res[1][k][0]
            if self.last_error != 0:
                return None
            self.Quantizer = self.want_q  # (the decoder's field describes the group's last frame: compared there)
            return self.b.planes(0, self.K - 1 - k)

        def close(self):
            self.b.close()

    def run_groups():
        rng = np.random.default_rng(2026)
        frames = both_ok = both_err = differ = 0
        for trial in range(trials):
            ver = 1 + trial % 2
            p = default_params("AB"[trial % 2], BASE_SEED + 9000 + trial, n_frames=4, width=96, height=64, version=ver, pm_intra=120, pm_deep=150,
                               pm_multiref=250, qdelta_prob=250, escape_prob=80, table1_prob=400)
            data, fo = generate_clip(p)
            d = data.copy()
            for _ in range(int(rng.integers(1, 8))):
                d[int(rng.integers(0, d.size))] ^= 1 << int(rng.integers(0, 8))
            g = GroupDecoder(p, lambda: ([d[: fo[f + 1]] for f in range(4)], [int(fo[f]) for f in range(4)]))
            o = OracleDecoder(p.width, p.height, p.version)
            for f in range(4):
                o.Data, o.Offset = d[: fo[f + 1]], int(fo[f])
                ro = o.DecodeFrame()
                g.want_q = o.Quantizer
                rg = g.DecodeFrame()
                frames += 1
                if g.last_error != 0 or o.last_error != 0:
                    both_err += 1
                    if (g.last_error == 0) != (o.last_error == 0) and g.last_error != -5:
                        differ += 1
                    break
                both_ok += 1
                if g.Offset != o.Offset or not np.array_equal(rg[0], ro[0]) or not np.array_equal(rg[1], ro[1]) or (f == 3 and g.b.quantizer(0) != o.Quantizer):
                    differ += 1
            g.close()
            o.close()
        print(f"GPU, frame-parallel groups of four (mobi_batch_decode_gop): {frames} corrupted frames: {both_ok} decoded by both, {both_err} rejected by both; {differ} frames with a different result")
        return differ

    bad += run_groups()
else:
    import ctypes as C
    from tests.interp_binding import InterpDecoder, lib as interp_lib
    bad += run(lambda p: InterpDecoder(p.width, p.height, p.version), "CPU (parser + command-list interpreter)")
    cnt = (C.c_ulong * 4)()
    interp_lib().mobi_cmdinterp_refusals(cnt)
    print("refusals by cause (all refusing frames, whether or not the reference decodes them): |MV| > 8191: %d, ModsDS quantiser < 12: %d, "
          "run past the block: %d, plane parameter outside int16: %d" % tuple(cnt))
sys.exit(1 if bad else 0)
