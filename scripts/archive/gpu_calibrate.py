"""Measure GPU-vs-oracle error levels for the configurations the GPU tests assert on."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle.oracle import OracleEnv
from plen_ml_walk_amd.vec_env import PlenVecEnv

def rollout_cmp(dtype, n, T, rolling=None, joint_act=False, amp=1.0, seed=0):
    ov = {} if rolling is None else {"rolling_friction": rolling}
    env = PlenVecEnv(n, dtype=dtype, joint_act=joint_act, cfg_overrides=ov)
    ors = []
    for i in range(n):
        o = OracleEnv(joint_act=joint_act)
        if rolling is not None: o.set_friction(rolling=rolling)
        o.reset(); ors.append(o)
    env.reset()
    g = torch.Generator().manual_seed(seed)
    if joint_act:
        t = torch.arange(T)[:, None, None].float()
        ph = torch.rand(1, n, 18, generator=g) * 6.28
        acts = amp * torch.sin(2 * 3.14159 * t / 80.0 + ph)
    else:
        acts = (torch.rand(T, n, 18, generator=g) * 2 - 1) * amp
    acts = acts.float()
    alive = np.ones(n, bool); worst_o = 0; worst_r = 0; mism = 0; hist = []
    for t in range(T):
        nobs, rew, done, info = env.step(acts[t].cuda())
        nobs = nobs.cpu().numpy().astype(np.float64); rew = rew.cpu().numpy().astype(np.float64); fl = done.cpu().numpy()
        eo = 0
        for i in range(n):
            if not alive[i]: continue
            ob, r, d, _ = ors[i].step(acts[t, i].numpy().astype(np.float64))
            eo = max(eo, np.abs(ob - nobs[i]).max()); worst_r = max(worst_r, abs(r - rew[i]) / max(1, abs(r)))
            if bool(fl[i] & 1) != d: mism += 1
            if d or fl[i]: alive[i] = False
        worst_o = max(worst_o, eo); hist.append(eo)
    aux = env.get_aux().cpu().numpy()
    env.close()
    return worst_o, worst_r, mism, int(alive.sum()), ["%.1e" % h for h in hist[::max(1, T // 8)]], aux[:, 0].max()

if __name__ == "__main__":
    for dtype in (torch.float64, torch.float32):
        print(dtype, "rolling=0 random actions   ", rollout_cmp(dtype, 24, 12, rolling=0.0))
        print(dtype, "rolling=0 amp 0.3          ", rollout_cmp(dtype, 16, 40, rolling=0.0, amp=0.3))
        print(dtype, "joint_act sin amp 0.15     ", rollout_cmp(dtype, 8, 200, joint_act=True, amp=0.15))
        print(dtype, "joint_act sin amp 0.3      ", rollout_cmp(dtype, 8, 200, joint_act=True, amp=0.3))
        print(dtype, "reference cfg 1 step       ", rollout_cmp(dtype, 64, 1))
        print(dtype, "reference cfg amp0.2 8 step", rollout_cmp(dtype, 32, 8, amp=0.2))
