from dcsnet.c_network import *  # noqa: F401,F403
from dcsnet.c_network import C_NETWORK, ComplexLSTM, ComplexChannelAttention, ComplexSpatialAttention  # noqa: F401
