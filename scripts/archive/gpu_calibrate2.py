"""Distribution of one-step errors from injected states in the reference configuration, next to the
oracle's own sensitivity to input perturbations of f32 size."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle.oracle import OracleEnv
from plen_ml_walk_amd.vec_env import PlenVecEnv

def collect(n, seed=3, amp=1.0):
    rng = np.random.default_rng(seed)
    e = OracleEnv(); e.reset(); S = []
    t = 0
    while len(S) < n:
        a = rng.uniform(-1, 1, 18) * amp
        ob, r, d, _ = e.step(a); t += 1
        if d or t % 40 == 0: e.reset(); continue
        if t % 2 == 0: S.append(e.get_state())
    return np.array(S)

def one_step_oracle(s, a, pert=0.0, rng=None):
    o = OracleEnv()
    if pert: s = s * (1 + pert * rng.standard_normal(s.shape))
    o.set_state(s); o.lib.oracle_script_reset(o.h)
    return o.step(a)[0]

if __name__ == "__main__":
    n = 128
    for amp in (1.0, 0.3):
        S = collect(n, amp=amp)
        rng = np.random.default_rng(9)
        A = (rng.uniform(-1, 1, (n, 18)) * amp).astype(np.float32)
        ref = np.array([one_step_oracle(S[i], A[i].astype(np.float64)) for i in range(n)])
        sens = np.array([np.abs(one_step_oracle(S[i], A[i].astype(np.float64), 6e-8, rng) - ref[i]).max() for i in range(n)])
        print("amp", amp, "oracle self-sensitivity to 6e-8 relative state perturbation: quantiles 10/50/90/99/max", np.quantile(sens, [.1, .5, .9, .99, 1]))
        for dtype in (torch.float64, torch.float32):
            env = PlenVecEnv(n, dtype=dtype)
            env.set_state(torch.tensor(S))
            nobs, rew, done, info = env.step(torch.tensor(A).cuda())
            err = np.abs(nobs.cpu().numpy().astype(np.float64) - ref)[:, :24].max(1)     # contact flags excluded
            cf = np.abs(nobs.cpu().numpy().astype(np.float64) - ref)[:, 24:].max(1)
            print("  ", dtype, "err quantiles 10/50/90/99/max", np.quantile(err, [.1, .5, .9, .99, 1]), "frac<=1e-4", (err <= 1e-4).mean(), "contact flag mismatches", int((cf > 0).sum()))
            if dtype == torch.float32:
                ratio = err / np.maximum(sens, 1e-6)
                print("     f32 err / max(sens,1e-6): quantiles 50/90/99/max", np.quantile(ratio, [.5, .9, .99, 1]))
            env.close()
