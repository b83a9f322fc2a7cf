"""Contact points in range per foot in random-action rollouts of 4096 envs, and the per-point branch count that follows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from plen_ml_walk_amd.vec_env import PlenVecEnv
n = 4096
env = PlenVecEnv(n); env.reset()
g = torch.Generator(device="cuda").manual_seed(0)
hist = np.zeros((5,), dtype=np.int64); feet = 0; taken = 0; compact_taken = 0; tot_foot_passes = 0
for t in range(120):
    a = torch.rand(n, 18, generator=g, device="cuda") * 2 - 1
    env.step(a)
    if t % 10 == 9:
        tg = torch.zeros(n, 18)
        env.debug_substeps(tg, nsub=1, dump=False)
        aux = env.get_aux().cpu().numpy()
        occ = (aux[:, 7] >> 8) & 0xff
        for f in range(2):
            nib = (occ >> (4 * f)) & 0xf
            k = np.array([bin(x).count("1") for x in nib])
            for kk in range(5): hist[kk] += (k == kk).sum()
            touching = k > 0
            taken += ((4 - k) * touching).sum()
            compact_taken += ((k < 4) * touching).sum()
            feet += touching.sum()
print("active points per foot histogram (0..4):", hist, " touching feet:", feet)
print("taken point-skips per touching foot per pass: now %.2f, compacted %.2f" % (taken / max(feet, 1), compact_taken / max(feet, 1)))
