"""Experiment: 4096 envs as G independent groups stepped on G streams (no cross-group sync) vs one 4096-env launch per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from plen_ml_walk_amd.vec_env import PlenVecEnv

def run(groups, n_total=4096, steps=200, warm=20):
    n = n_total // groups
    envs = [PlenVecEnv(n) for _ in range(groups)]
    streams = [torch.cuda.Stream() for _ in range(groups)]
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    acts = torch.rand(64, n_total, 18, device="cuda", generator=g) * 2 - 1
    for e in envs: e.reset()
    torch.cuda.synchronize()
    def loop(k0, k):
        for t in range(k0, k0 + k):
            a = acts[t % 64]
            for i, (e, s) in enumerate(zip(envs, streams)):
                with torch.cuda.stream(s):
                    e.step(a[i * n:(i + 1) * n])
    loop(0, warm); torch.cuda.synchronize()
    t0 = time.perf_counter(); loop(warm, steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    for e in envs: e.close()
    return dt / steps * 1e3

for groups in (1, 2, 4, 8):
    ms = run(groups)
    print("groups %d x %4d envs: %.4f ms per 4096 env-steps -> %.2f M env-steps/s" % (groups, 4096 // groups, ms, 4096 / ms / 1e3), flush=True)
