"""Exploration behind tests/test_full_size_gpu.py: 4096 envs x T steps, kernel vs oracle on every env (error quantiles per step)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import oracle
from plen_ml_walk_amd.vec_env import PlenVecEnv
n, T = 4096, 12
g = torch.Generator(device="cuda").manual_seed(0)
acts = (torch.rand(T, n, 18, generator=g, device="cuda") * 2 - 1)
ms = 0.8 + 0.4 * torch.rand(n, generator=g, device="cuda"); mu = 0.4 + 0.6 * torch.rand(n, generator=g, device="cuda")
for name, dtype, rolling, dr in (("f64 ref", torch.float64, None, False), ("f64 rolling0", torch.float64, 0.0, False), ("f64 DR", torch.float64, None, True),
                                 ("f32 ref", torch.float32, None, False), ("f32 rolling0", torch.float32, 0.0, False)):
    ov = {} if rolling is None else dict(rolling_friction=rolling)
    env = PlenVecEnv(n, dtype=dtype, cfg_overrides=ov)
    if dr: env.set_params(ms.to(dtype), mu.to(dtype))
    env.reset()
    O, R, D = [], [], []
    for t in range(T):
        o, r, d, _ = env.step(acts[t]); O.append(o.cpu().numpy().astype(np.float64)); R.append(r.cpu().numpy().astype(np.float64)); D.append(d.cpu().numpy())
    env.close()
    t0 = time.time()
    oo, rr, ff = oracle.batch_rollout(acts.cpu().numpy(), ms.cpu().numpy() if dr else None, mu.cpu().numpy() if dr else None, -1.0 if rolling is None else rolling)
    dt = time.time() - t0
    O, R, D = np.array(O), np.array(R), np.array(D)
    err = np.abs(O - oo).max(2)
    print(name, "oracle %.1fs" % dt)
    for t in range(T):
        e = err[t]
        print("  t=%2d med %.1e p90 %.1e p99 %.1e max %.1e | <=1e-4: %.3f | flags equal %.4f contact flags equal %.4f | rew err med %.1e" % (
            t, np.median(e), np.quantile(e, .9), np.quantile(e, .99), e.max(), (e <= 1e-4).mean(), (D[t] == ff[t]).mean(), (O[t][:, 24:26] == oo[t][:, 24:26]).all(1).mean(),
            np.median(np.abs(R[t] - rr[t]))))
