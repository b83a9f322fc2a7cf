"""Thin re-export for the scripts in this directory: the pin lives in tests/pybullet_pin.py (test infrastructure)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pybullet_pin import ACTS, SD, HYP, pre, target, jac, make_oracle, residuals as _residuals


def make_env(hyp=None, urdf_inertia=False, inertia=None):
    return make_oracle(hyp=hyp, urdf_inertia=urdf_inertia, inertia=inertia)


def residuals(e, K=12):
    def step(a):
        o, _, d, _ = e.step(np.asarray(a, dtype=np.float64))
        return o, d
    return _residuals(e.reset, step, K)


def survive(e, maxT=500):
    e.reset()
    for t in range(maxT):
        _, _, done, _ = e.step(ACTS[t].astype(np.float64))
        if done:
            return t + 1
    return maxT
