"""Minimal `gym` stand-in used only when the real package is absent and compat/ is on PYTHONPATH:
`gym.make("PlenWalkEnv-v1", render=False)` as called by plen_bullet/src/plen_td3.py:43."""
from plen_ml_walk_amd.gym_compat import Box, Env, TimeLimit, register, make  # noqa: F401
import plen_ml_walk_amd.plen_env  # noqa: F401  (registers PlenWalkEnv-v1)


class spaces(object):   # noqa: N801
    Box = Box
