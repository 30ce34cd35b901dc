import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build (or reuse) every in-tree native library once per session.  On the GPU box the .so
    files travel with the snapshot; rebuilding there is a no-op unless sources are newer."""
    from mobiclipdecoder_amd import build
    build.build_all()
    yield


@pytest.fixture
def profiling_library(monkeypatch):
    """The -DMOBI_PROFILING twin of the product library (libmobiclip_hip_prof.so: same sources plus the mobi_debug_* test hooks, which the
    product library does not export) as the library behind MobiclipDecoder / MobiclipBatch for the duration of one test."""
    from mobiclipdecoder_amd import build, decoder
    lib = decoder.bind_library(build.build_hip(profiling=True))
    monkeypatch.setattr(decoder, "_LIB", lib)
    yield lib


# Small frame steps go out as ONE launch (mobi_recon_step: the octets of inter macroblocks and the intra fours side by side), large ones as
# two (mobi_recon_inter8, mobi_recon_intra); almost every test batch is small.  The parity modules therefore run twice on the GPU, once
# per kind of step (MOBI_FUSED_STEP_MBS is read when a batch is created).
_BOTH_KINDS_OF_STEP = {"test_gpu_parity", "test_golden", "test_internal_walk", "test_demux", "test_abi_c_caller", "test_unit_vectors"}


def pytest_generate_tests(metafunc):
    if metafunc.module.__name__.split(".")[-1] in _BOTH_KINDS_OF_STEP and metafunc.definition.get_closest_marker("gpu"):
        metafunc.fixturenames.append("step_launches")
        metafunc.parametrize("step_launches", ["one_launch", "two_launches"], indirect=True)


@pytest.fixture
def step_launches(request, monkeypatch):
    if request.param == "two_launches":
        monkeypatch.setenv("MOBI_FUSED_STEP_MBS", "0")
    else:
        monkeypatch.delenv("MOBI_FUSED_STEP_MBS", raising=False)
    yield request.param
