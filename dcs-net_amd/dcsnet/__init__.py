"""dcsnet — MI355X-native DCS-Net hot path (complex encoder/decoder + subtractive mask).

Host side is Python on PyTorch-ROCm mirroring the reference's module surface; the arithmetic is
hand-written HIP for gfx950 in lib/libdcsnet_hip.so (C ABI: include/dcsnet_hip.h).
There is no CPU or eager fallback: importing the ops without the built library raises.
"""
from ._lib import DcsHipError, LIB_PATH, load as load_library  # noqa: F401

__all__ = ['DcsHipError', 'LIB_PATH', 'load_library']
