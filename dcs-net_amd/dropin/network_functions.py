from dcsnet.network_functions import *  # noqa: F401,F403
import sys, torch  # noqa: F401,E401  (the reference star-imports these names from here)
