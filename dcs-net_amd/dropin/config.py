from dcsnet.config import *  # noqa: F401,F403
from dcsnet.config import hparams, Config, config  # noqa: F401
