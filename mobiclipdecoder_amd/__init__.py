"""mobiclipdecoder_amd -- MI355X-native Mobiclip frame reconstruction.

Host-side mirror of ``LibMobiclip.Codec.Mobiclip.MobiclipDecoder`` (MobiclipDecoder.cs:13-61) over the
C ABI of ``libmobiclip_hip.so`` (include/mobiclip_hip.h).  Pixels are only ever produced by the HIP
kernels; importing the decoder classes without the built library raises.
"""
from .decoder import (  # noqa: F401
    MobiclipDecoder,
    MobiclipBatch,
    MobiclipVersion,
    MobiclipError,
    load_library,
    unpack_motion_search,
)
from .streamgen import GenParams, generate_clip, default_params  # noqa: F401
