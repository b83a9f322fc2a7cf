"""Diagnostic build (-DPLEN_SLIDE_STATS, f64): how often is a lateral-friction pair OUTSIDE its friction circle (the cone projection's slow path) -- per solver iteration, in the
two state distributions bench.py times: random actions and the shipped walking policy.  One debug substep (with dump) is taken from every env's state after a rollout.
usage: python scripts/gpu_slide_stats.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from plen_ml_walk_amd.build import build_variant
os.environ["PLENVEC_LIB"] = os.environ.get("STATS_LIB") or build_variant("slide_stats", ["-DPLEN_SLIDE_STATS", "-DPLENVEC_CONE_STRAIGHT=0"])
import numpy as np, torch
import pybullet_pin as P
from plen_ml_walk_amd.vec_env import PlenVecEnv
dev = torch.device("cuda:0")
W = {k: torch.from_numpy(v).to(dev).to(torch.float32) for k, v in P.SD.items()}
def actor(o):
    h = torch.relu(o @ W["fc1.weight"].T + W["fc1.bias"]); h = torch.relu(h @ W["fc2.weight"].T + W["fc2.bias"]); return torch.tanh(h @ W["fc3.weight"].T + W["fc3.bias"])
lo = np.array([-1.57, -0.15, -0.95, -0.9, -0.95, -0.8, -1.57, -1.5, -0.75, -0.3, -1.2, -0.4, -1.57, -0.15, -0.2, -1.57, -0.15, -0.2])
hi = np.array([1.57, 1.5, 0.75, 0.3, 1.2, 0.4, 1.57, 0.15, 0.95, 0.9, 0.95, 0.8, 1.57, 1.57, 0.35, 1.57, 1.57, 0.35])
n = 4096
for what in ("random", "walking"):
    env = PlenVecEnv(n, device=dev, dtype=torch.float64); obs = env.reset().to(torch.float32).clone()
    g = torch.Generator(device=dev).manual_seed(3)
    rows = []
    for rep in range(4):
        for _ in range(60 if rep else 150):
            a = (torch.rand(n, 18, generator=g, device=dev) * 2 - 1) if what == "random" else torch.clamp(actor(obs) + 0.01 * torch.randn(n, 18, generator=g, device=dev), -1, 1)
            _, _, _, info = env.step(a); obs = info["obs"].to(torch.float32)
        a = (torch.rand(n, 18, generator=g, device=dev) * 2 - 1) if what == "random" else torch.clamp(actor(obs), -1, 1)
        tg = torch.from_numpy((hi - lo) / 2 * a.double().cpu().numpy() + (hi + lo) / 2)
        d = env.debug_substeps(tg, nsub=1, dump=True)
        aux = env.get_aux().cpu().numpy()
        act = (aux[:, 7] >> 8) & 0xff
        npt = np.array([bin(x).count("1") for x in act])
        rows.append(np.stack([npt, d[:, 3700].cpu().numpy(), d[:, 3702].cpu().numpy(), d[:, 3703].cpu().numpy()], 1))
    r = np.concatenate(rows)
    npt, its, sl, fl = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
    c = npt > 0
    print("%s: %.1f %% of substeps have contact points (mean %.2f points); of those: %.1f %% have NO sliding iteration, %.1f %% slide in EVERY iteration; "
          "sliding iterations %.1f %% of all; mean changes of state (sliding <-> sticking) per substep %.2f" % (
              what, 100 * c.mean(), npt[c].mean(), 100 * (sl[c] == 0).mean(), 100 * (sl[c] >= its[c]).mean(), 100 * sl[c].sum() / its[c].sum(), fl[c].mean()))
    for k in range(1, 9):
        m = npt == k
        if m.sum() > 20: print("   %d points: %5.1f %% of substeps, sliding iterations %.1f %%, none %.1f %%" % (k, 100 * m.mean(), 100 * sl[m].sum() / its[m].sum(), 100 * (sl[m] == 0).mean()))
    env.close()
