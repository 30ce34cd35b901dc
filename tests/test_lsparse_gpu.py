"""The lock-step parser on the GPU (mobi_parse_frames_ls, mobi_lsparse.hip: clips in lock step, one per lane, parse mode 3): the device-parse tests of
tests/test_device_parse.py once more with it in front -- oracle parity of planes, rc, Offset, Quantizer on good streams, on the streams the
reference throws on (where it must hand the clip to mobi_parse_frames), on fuzzed streams against the host parser, and asynchronous steps --
plus a check that it really finishes the intact frames itself instead of handing everything over."""
import numpy as np
import pytest

import tests.test_device_parse as T
from mobiclipdecoder_amd import MobiclipBatch, default_params, generate_clip
from mobiclipdecoder_amd.streamgen import BASE_SEED
from tests.oracle_binding import OracleDecoder

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _lockstep_mode(monkeypatch):
    monkeypatch.setattr(T, "DEVICE_PARSE", "lockstep")


test_default_streams = T.test_default_streams
test_rich_streams_whole_file = T.test_rich_streams_whole_file
test_long_gop_640x480_uses_every_reference_slot = T.test_long_gop_640x480_uses_every_reference_slot
test_edge_motion_vectors = T.test_edge_motion_vectors
test_uneven_clip_count_and_wide_picture = T.test_uneven_clip_count_and_wide_picture
test_streams_the_reference_throws_on = T.test_streams_the_reference_throws_on
test_fuzzed_streams_device_parse_equals_host_parse = T.test_fuzzed_streams_device_parse_equals_host_parse
test_host_and_device_parse_agree_on_a_larger_batch = T.test_host_and_device_parse_agree_on_a_larger_batch
test_asynchronous_steps_equal_the_oracle = T.test_asynchronous_steps_equal_the_oracle
test_asynchronous_steps_every_frame_checked = T.test_asynchronous_steps_every_frame_checked


def test_intact_frames_are_finished_by_the_lock_step_parser_itself():
    """70 clips (two waves, the second one sparse), one of them damaged: every intact frame is the lock-step parser's, the damaged clip's
    frames go to mobi_parse_frames from the damage on, and the planes are the oracle's either way."""
    n, nfr = 70, 5
    ps = [default_params("A", BASE_SEED + 1700 + (i % 7), n_frames=nfr, pm_intra=100, iframe_interval=3) for i in range(n)]
    clips = [generate_clip(p) for p in ps]
    bad = 33
    d = np.array(clips[bad][0], copy=True)
    d[int(clips[bad][1][2]) + 11] ^= 0x3C  # inside frame 2
    clips[bad] = (d, clips[bad][1])
    b = MobiclipBatch(n, ps[0].width, ps[0].height, ps[0].version, device_parse="lockstep")
    oras = [OracleDecoder(ps[0].width, ps[0].height, ps[0].version) for _ in range(n)]
    bad_hist = []
    for f in range(nfr):
        datas = [c[0][c[1][f]:c[1][f + 1]] for c in clips]
        rcs, offs = b.decode(datas, [0] * n)
        fin = b.lockstep_finished()
        assert fin >= n - 1, (f, fin)
        if f < 2:
            assert fin == n, (f, fin)
        for i in range(n):
            oras[i].Data, oras[i].Offset = datas[i], 0
            o = oras[i].DecodeFrame()
            if i == bad and f >= 2:
                if rcs[i] in (-5, -6) or oras[i].last_error == -5:
                    continue
                assert rcs[i] == oras[i].last_error, (f, rcs[i], oras[i].last_error)
                if rcs[i] != 0:
                    continue
                if any(x != 0 for x in bad_hist):
                    continue  # after a frame that threw, the reference keeps a partial picture
            else:
                assert rcs[i] == 0 and oras[i].last_error == 0, (f, i, rcs[i])
            assert offs[i] == oras[i].Offset, (f, i)
            y, uv = b.planes(i)
            assert np.array_equal(y, o[0]) and np.array_equal(uv, o[1]), (f, i)
        bad_hist.append(rcs[bad])
    b.close()
    for o in oras:
        o.close()


@pytest.mark.parametrize("clips_per_wave, waves", [(1, 4), (3, 8), (12, 8), (24, 8), (27, 8), (32, 4), (64, 2), (7, 1)])
def test_every_shape_of_the_launch(clips_per_wave, waves, monkeypatch, profiling_library):
    """mobi_launch_parse_ls deals the clips by batch size (1 .. 64 per wave, four or eight waves per workgroup: one workgroup per CU for a full
    machine); a test batch is small and would only ever see one clip per wave.  The profiling twin takes both numbers from the environment:
    every shape the product chooses somewhere, on 150 clips (so that the last wave and the last workgroup are partly empty), every clip
    against the oracle at every frame, and every frame finished by the lock-step parser itself."""
    monkeypatch.setenv("MOBI_LS_CLIPS", str(clips_per_wave))
    monkeypatch.setenv("MOBI_LS_WG_WAVES", str(waves))
    n, nfr = 150, 4
    ps = [default_params("A", BASE_SEED + 5100 + (i % 37), n_frames=nfr, pm_intra=120, pm_deep=200, iframe_interval=3) for i in range(n)]
    clips = [generate_clip(p) for p in ps[:37]]
    b = MobiclipBatch(n, ps[0].width, ps[0].height, ps[0].version, device_parse="lockstep")
    oras = [OracleDecoder(ps[0].width, ps[0].height, ps[0].version) for _ in range(37)]
    for f in range(nfr):
        datas = [clips[i % 37][0][clips[i % 37][1][f]:clips[i % 37][1][f + 1]] for i in range(n)]
        rcs, offs = b.decode(datas, [0] * n)
        assert b.lockstep_finished() == n, (f, b.lockstep_finished())
        want = []
        for k in range(37):
            oras[k].Data, oras[k].Offset = datas[k], 0
            want.append(oras[k].DecodeFrame())
            assert oras[k].last_error == 0
        for i in range(n):
            assert rcs[i] == 0 and offs[i] == oras[i % 37].Offset, (f, i, rcs[i])
            y, uv = b.planes(i)
            assert np.array_equal(y, want[i % 37][0]) and np.array_equal(uv, want[i % 37][1]), (f, i)
    b.close()
    for o in oras:
        o.close()
