"""Contact points in range per foot under a freshly initialised actor + N(0, 0.1) noise (what the TD3 leg plays) and under the shipped walking policy, like gpu_slot_distribution.py does for random actions."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np
from plen_ml_walk_amd.vec_env import PlenVecEnv
from plen_ml_walk_amd.td3 import TD3Agent
n = 4096
dev = torch.device("cuda:0")
torch.manual_seed(0)
fresh = TD3Agent(26, 18, 1.0, device=dev)
shipped = TD3Agent(26, 18, 1.0, device=dev)
w = np.load(os.path.join(ROOT, "tests", "golden", "policy_3229999.npz"))
sd = shipped.actor.state_dict()
for k in sd:
    sd[k].copy_(torch.from_numpy(w["actor." + k]))
for name, ag, sig in (("fresh actor + N(0, 0.1)", fresh, 0.1), ("shipped policy + N(0, 0.01)", shipped, 0.01)):
    env = PlenVecEnv(n, device=dev); obs = env.reset().to(torch.float32)
    g = torch.Generator(device=dev).manual_seed(0)
    hist = np.zeros(5, dtype=np.int64); both_full = 0; samples = 0
    for t in range(200):
        with torch.no_grad():
            a = (ag.actor(obs) + sig * torch.randn(n, 18, generator=g, device=dev)).clamp(-1, 1)
        o, r, d, info = env.step(a)
        obs = info["obs"].to(torch.float32).clone()
        if t % 10 == 9:
            env.debug_substeps(torch.zeros(n, 18), nsub=1, dump=False)
            aux = env.get_aux().cpu().numpy()
            occ = (aux[:, 7] >> 8) & 0xff
            for f in range(2):
                k = np.array([bin(x).count("1") for x in (occ >> (4 * f)) & 0xf])
                for kk in range(5): hist[kk] += (k == kk).sum()
            both_full += (occ == 0xff).sum(); samples += n
    print("%-28s points per foot histogram (0..4): %s   both feet with 4 points: %.3f of the env-substeps" % (name, (hist / hist.sum()).round(3), both_full / samples), flush=True)
    env.close()
