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
