#!/usr/bin/env python3
"""Summary fixture of the reference's DATA file plen_bullet/results/plen_walk_gazebo_.npy (episode returns of its own PyBullet training
run, 24 832 episodes): block means per 1000 episodes, extremes -> tests/golden/ref_training_log_summary.npz.  Only statistics that the
documentation quotes (tests/test_env_gpu.py).  Run in the build container only (needs /root/reference)."""
import os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PLEN_REFERENCE", "/root/reference")
r = np.load(os.path.join(REF, "plen_bullet/results/plen_walk_gazebo_.npy"))
blocks = np.array([r[i:i + 1000].mean() for i in range(0, len(r) - len(r) % 1000, 1000)])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_training_log_summary.npz"), episodes=len(r), block_means_1000=blocks,
                    first100_mean=r[:100].mean(), last1000_mean=r[-1000:].mean(), max_return=r.max(), min_return=r.min(),
                    last1000_returns=r[-1000:].astype(np.float32))     # round 4: the tail's distribution (scripts/pin/closed_loop_stats.py)
print(len(r), blocks.round(1), r[-1000:].mean(), r.max())
