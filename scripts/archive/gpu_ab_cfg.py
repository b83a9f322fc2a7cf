"""A/B of cfg switches on ONE binary inside one gpurun call: ms per 4096-env step (single launch per step) for body_contacts on/off, both dtypes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from plen_ml_walk_amd.vec_env import PlenVecEnv
n = 4096
g = torch.Generator(device="cuda").manual_seed(0)
acts = torch.rand(64, n, 18, generator=g, device="cuda") * 2 - 1
for dtype in (torch.float32, torch.float64):
    for rep in range(2):
        for bc in (0, 1):
            env = PlenVecEnv(n, dtype=dtype, cfg_overrides=dict(body_contacts=bc)); env.reset()
            for t in range(30): env.step(acts[t % 64])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            K = 150
            for t in range(K): env.step(acts[(30 + t) % 64])
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
            print("%s body_contacts=%d  %.4f ms/step  %.2f M env-steps/s" % (str(dtype)[6:], bc, dt * 1e3, n / dt / 1e6), flush=True)
            env.close()
