"""HIP-backed functions with the surface of ``complexPyTorch.complexFunctions`` (0.3) that the
reference imports at c_network.py:7 (``complex_upsample``, ``complex_relu``)."""
from . import functional as F
from ._lib import DcsHipError


def complex_relu(input):
    return F.complex_relu(input)


def complex_upsample(input, size=None, scale_factor=None, mode='nearest', align_corners=None,
                     recompute_scale_factor=None):
    if size is not None or mode != 'nearest' or scale_factor is None:
        raise DcsHipError("complex_upsample: the HIP path implements mode='nearest' with an integer "
                          'scale_factor (config.py:105-106)')
    return F.complex_upsample(input, scale_factor)
