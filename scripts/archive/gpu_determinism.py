import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from plen_ml_walk_amd.walk_eval import load_policy
from plen_ml_walk_amd.vec_env import PlenVecEnv
pol = load_policy(os.path.join(ROOT, "tests/golden/policy_3229999.npz"))
def run(dtype, n, T=40):
    env = PlenVecEnv(n, dtype=dtype)
    obs = env.reset().float().clone(); out = []
    for t in range(T):
        a = pol.select_action_batch(obs).clamp(-1, 1)
        o, r, d, info = env.step(a)
        out.append(torch.cat([o.double(), r.double()[:, None], d.double()[:, None]], 1).cpu().numpy().copy())
        obs = info["obs"].float().clone()
    env.close()
    return np.array(out)
for dtype in (torch.float64, torch.float32):
    for n in (64, 256, 4096):
        a = run(dtype, n); b = run(dtype, n)
        same_envs = (a == a[:, :1]).all()
        first_bad = np.argwhere(~(a == a[:, :1]).all(axis=(1, 2)))
        print(dtype, "n", n, "run-to-run identical:", np.array_equal(a, b), "| all envs identical within run:", same_envs, "| first step with env mismatch:", first_bad[:1].ravel().tolist(),
              "| episode end step env0:", int(np.argmax(a[:, 0, 27] != 0)))
