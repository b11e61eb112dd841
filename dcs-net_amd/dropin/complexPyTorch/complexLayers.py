from dcsnet.complexLayers import *  # noqa: F401,F403
from dcsnet.complexLayers import ComplexConv2d, ComplexConvTranspose2d, ComplexBatchNorm2d, ComplexLinear, ComplexReLU  # noqa: F401
from torch.nn import Module as ComplexAvgPool2d  # placeholder name: the reference deletes it right after import (c_network.py:6)
