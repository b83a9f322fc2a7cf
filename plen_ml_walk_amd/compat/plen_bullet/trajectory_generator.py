"""Import shim: `from plen_bullet.trajectory_generator import TrajectoryGenerator` (reference trajectory_eval.py:9)."""
from plen_ml_walk_amd.trajectory_generator import *          # noqa: F401,F403
from plen_ml_walk_amd.trajectory_generator import TrajectoryGenerator    # noqa: F401
