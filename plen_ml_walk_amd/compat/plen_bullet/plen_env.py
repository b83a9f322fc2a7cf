"""Drop-in for the reference module plen_bullet.plen_env (plen_bullet/src/plen_bullet/plen_env.py)."""
from plen_ml_walk_amd.plen_env import PlenWalkEnv, ENV_RANGES, REAL_RANGES, JOINT_NAMES, MOVING_JOINTS  # noqa: F401
