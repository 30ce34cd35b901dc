"""Imported first by the tools that call mobi_debug_* hooks or read MOBI_DEBUG / MOBI_STOP_STAGE / MOBI_*_PAD: they need the
-DMOBI_PROFILING twin of the library (python -m mobiclipdecoder_amd.build --profiling), which the product package loads when MOBI_LIB names it."""
import os

_P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mobiclipdecoder_amd", "libmobiclip_hip_prof.so")
os.environ.setdefault("MOBI_LIB", _P)
