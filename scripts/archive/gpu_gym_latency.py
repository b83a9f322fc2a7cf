import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from plen_ml_walk_amd.plen_env import PlenWalkEnv
for kw in (dict(), ):
    env = PlenWalkEnv()
    obs = env.reset()
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, size=(2000, 18)).astype(np.float32)
    for i in range(100): 
        o, r, d, _ = env.step(acts[i])
        if d: env.reset()
    t0 = time.perf_counter(); n = 0; resets = 0
    for i in range(100, 2000):
        o, r, d, _ = env.step(acts[i]); n += 1
        if d: env.reset(); resets += 1
    dt = time.perf_counter() - t0
    print("gym facade single env: %.0f steps/s (%.3f ms per step, %d resets)" % (n / dt, dt / n * 1e3, resets), type(o), o.dtype)
