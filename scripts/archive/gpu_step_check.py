"""GPU bring-up 2: full env steps (action map, 4 substeps, obs/reward/done, auto-reset) vs the C oracle,
then a quick throughput probe at 4096 envs."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle.oracle import OracleEnv
from plen_ml_walk_amd.vec_env import PlenVecEnv

def parity(dtype, n=32, T=40):
    g = torch.Generator(device="cpu").manual_seed(0)
    acts = (torch.rand(T, n, 18, generator=g) * 2 - 1).float()
    env = PlenVecEnv(n, dtype=dtype)
    obs0 = env.reset().cpu().numpy()
    ors = [OracleEnv() for _ in range(n)]
    o0 = np.array([o.reset() for o in ors])
    print(dtype, "reset obs diff", np.abs(obs0 - o0).max())
    errs = []
    alive = np.ones(n, bool)       # envs still on their first episode in both worlds
    for t in range(T):
        nobs, rew, done, info = env.step(acts[t].cuda())
        nobs = nobs.cpu().numpy().astype(np.float64); rew = rew.cpu().numpy().astype(np.float64); fl = info["flags"].cpu().numpy()
        e_obs = np.zeros(n); e_rew = np.zeros(n); mism = 0
        for i in range(n):
            if not alive[i]: continue
            ob, r, d, _ = ors[i].step(acts[t, i].numpy().astype(np.float64))
            e_obs[i] = np.abs(ob - nobs[i]).max(); e_rew[i] = abs(r - rew[i])
            if bool(fl[i] & 1) != d: mism += 1
            if d or (fl[i] != 0): alive[i] = False
        errs.append((t, alive.sum(), e_obs.max(), np.median(e_obs[e_obs > 0]) if (e_obs > 0).any() else 0, e_rew.max(), mism))
    for e in errs[:12] + errs[-3:]:
        print("  t=%d alive=%d obs max err %.3e median %.3e reward err %.3e done mismatches %d" % e)
    env.close()

def throughput(dtype, n=4096, steps=50):
    env = PlenVecEnv(n, dtype=dtype)
    env.reset()
    acts = (torch.rand(steps, n, 18, device="cuda") * 2 - 1)
    for t in range(5): env.step(acts[t])
    torch.cuda.synchronize()
    env.timing_begin(); t0 = time.time()
    for t in range(steps): env.step(acts[t])
    ms, nl = env.timing_end(); torch.cuda.synchronize(); wall = time.time() - t0
    print(dtype, "N=%d: %.3f ms/step (events, %d launches) wall %.3f ms/step -> %.3f M env-steps/s" % (n, ms / nl, nl, wall / steps * 1e3, n / (ms / nl) / 1e3))
    env.close()

if __name__ == "__main__":
    parity(torch.float64); parity(torch.float32)
    for n in (4096, 16384):
        throughput(torch.float32, n); throughput(torch.float64, n)
