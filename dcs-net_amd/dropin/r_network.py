from dcsnet.r_network import *  # noqa: F401,F403
from dcsnet.r_network import R_NETWORK, RealChannelAttention, RealSpatialAttention  # noqa: F401
