"""The C ABI driven from plain C (tests/tools/abi_caller.c: include/mobiclip_hip.h + libc, no Python in the data path) and on
more than one device.  The reference's own binding would be C# P/Invoke; no .NET exists in this image, so a C
caller is the closest stand-in for "a host in another language"."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
MAN = json.load(open(os.path.join(GOLD, "golden.json")))
CALLER = os.path.join(ROOT, "tests", "tools", "abi_caller")


def test_c_caller_is_built_and_links_only_the_product_library():
    """build.py compiles it with plain gcc against include/mobiclip_hip.h; it must not link the oracle."""
    assert os.path.exists(CALLER), "run python -m mobiclipdecoder_amd.build"
    needed = subprocess.run(["readelf", "-d", CALLER], capture_output=True, text=True).stdout
    assert "libmobiclip_hip.so" in needed and "oracle" not in needed


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in MAN["cases"] if c["name"] in ("mods_256x192_A", "moflex_640x480_B", "mods_64x48_rich", "mods_256x192_edge_wrap")],
                         ids=lambda c: c["name"])
def test_c_caller_reproduces_golden(case):
    args = [CALLER, os.path.join(GOLD, case["name"] + ".bin"), str(case["width"]), str(case["height"]), str(int(case["version"])),
            str(len(case["frames"]))] + [str(o) for o in case["frame_off"][: len(case["frames"]) + 1]]
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "mobiclipdecoder_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run(args, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    lines = [l.split() for l in r.stdout.strip().splitlines()]
    assert len(lines) == len(case["frames"])
    for l, exp in zip(lines, case["frames"]):
        assert int(l[1]) == 0 and int(l[2]) == exp["offset_after"] and int(l[3]) == exp["quantizer"], (case["name"], l)
        assert l[4] == exp["y_sha256"] and l[5] == exp["uv_sha256"], (case["name"], l[0])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mods_256x192_A", "moflex_640x480_B"])
def test_c_caller_asynchronous_batch_reproduces_golden(name):
    """The same golden streams through mobi_batch_submit / mobi_batch_wait from C: three clips, parse on the GPU, two frame steps in
    flight; frame f is read back as ring position 1 while frame f + 1 is already under way."""
    case = [c for c in MAN["cases"] if c["name"] == name][0]
    args = [CALLER, "--batch-async", os.path.join(GOLD, name + ".bin"), str(case["width"]), str(case["height"]), str(int(case["version"])),
            str(len(case["frames"]))] + [str(o) for o in case["frame_off"][: len(case["frames"]) + 1]]
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "mobiclipdecoder_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run(args, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    lines = [l.split() for l in r.stdout.strip().splitlines()]
    assert len(lines) == len(case["frames"])
    for l, exp in zip(lines, case["frames"]):
        assert int(l[1]) == 0 and int(l[2]) == exp["offset_after"] and int(l[3]) == exp["quantizer"], (name, l)
        assert l[4] == exp["y_sha256"] and l[5] == exp["uv_sha256"], (name, l[0])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mods_256x192_A", "moflex_640x480_B", "mods_64x48_rich", "moflex_64x48_rich_iint"])
def test_c_caller_frame_parallel_groups_reproduce_golden(name):
    """The golden streams through mobi_batch_decode_gop from C (r06): three clips, groups of 4, 1, 6, 2 frames parsed side by side on the
    GPU, every frame of a group read back from the ring afterwards."""
    case = [c for c in MAN["cases"] if c["name"] == name][0]
    args = [CALLER, "--batch-gop", os.path.join(GOLD, name + ".bin"), str(case["width"]), str(case["height"]), str(int(case["version"])),
            str(len(case["frames"]))] + [str(o) for o in case["frame_off"][: len(case["frames"]) + 1]]
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "mobiclipdecoder_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run(args, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    lines = [l.split() for l in r.stdout.strip().splitlines()]
    assert len(lines) == len(case["frames"])
    for l, exp in zip(lines, case["frames"]):
        assert int(l[1]) == 0 and int(l[2]) == exp["offset_after"] and int(l[3]) in (0, exp["quantizer"]), (name, l)
        assert l[4] == exp["y_sha256"] and l[5] == exp["uv_sha256"], (name, l[0])
    assert any(int(l[3]) for l in lines)


@pytest.mark.gpu
def test_batches_on_two_devices_in_one_process():
    """mobi_batch_create(..., device) with device != 0: a batch on GPU 0 and one on GPU 1, driven alternately from one thread
    (config 4 of BASELINE.json shards clips over the GPUs of a node; one process may own several).  Skipped on a 1-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices")
    from mobiclipdecoder_amd import MobiclipBatch, default_params, generate_clip
    from mobiclipdecoder_amd.streamgen import BASE_SEED
    from tests.oracle_binding import OracleDecoder
    ps = [default_params("A", BASE_SEED + 900 + i, n_frames=5, pm_intra=150) for i in range(2)]
    clips = [generate_clip(p) for p in ps]
    bs = [MobiclipBatch(3, ps[0].width, ps[0].height, ps[0].version, device=dev) for dev in (0, 1)]
    oras = [OracleDecoder(p.width, p.height, p.version) for p in ps]
    for f in range(5):
        for dev in (1, 0):  # interleaved: every entry point must select its own device
            data, fo = clips[dev]
            rcs, offs = bs[dev].decode([data[fo[f]:fo[f + 1]]] * 3, [0, 0, 0])
            assert rcs == [0, 0, 0]
        for dev in (0, 1):
            data, fo = clips[dev]
            oras[dev].Data, oras[dev].Offset = data[fo[f]:fo[f + 1]], 0
            o = oras[dev].DecodeFrame()
            for c in range(3):
                y, uv = bs[dev].planes(c)
                assert np.array_equal(y, o[0]) and np.array_equal(uv, o[1]), (dev, f, c)
    for b in bs:
        b.close()
