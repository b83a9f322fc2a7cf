import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plen_ml_walk_amd.walk_eval import load_policy
from plen_ml_walk_amd.vec_env import PlenVecEnv
pol = load_policy(os.path.join(ROOT, "tests/golden/policy_3229999.npz"))
def run(dtype, sigma, n=256, seed=0, **ov):
    env = PlenVecEnv(n, dtype=dtype, cfg_overrides=ov)
    obs = env.reset().float().clone()
    g = torch.Generator(device="cuda").manual_seed(seed)
    ret = torch.zeros(n, device="cuda"); ln = torch.zeros(n, device="cuda"); alive = torch.ones(n, dtype=torch.bool, device="cuda")
    for t in range(500):
        a = pol.select_action_batch(obs)
        a = (a + sigma * torch.randn(a.shape, device="cuda", generator=g)).clamp(-1, 1)
        _, r, d, info = env.step(a)
        ret += r.float() * alive; ln += alive
        alive &= (d == 0)
        obs = info["obs"].float().clone()
    env.close()
    return ret.cpu().numpy(), ln.cpu().numpy()
for dtype in (torch.float32, torch.float64):
    for sigma in (0.0, 0.01, 0.05, 0.1):
        r, l = run(dtype, sigma)
        print(dtype, "sigma", sigma, "mean return %.1f" % r.mean(), "median length %.0f" % np.median(l), "mean length %.1f" % l.mean(), "full-length frac %.2f" % (l >= 500).mean())
