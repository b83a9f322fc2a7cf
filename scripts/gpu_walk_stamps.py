"""Per-iteration solver cycles (shader clock, -DPGS_STAMPS builds) in the WALKING state distribution, by foot point counts: the shipped policy drives 4096 envs for 150 steps,
then one debug substep with stamps is taken from every env's state.  usage: STAMPS_LIB=<stamps build> python scripts/gpu_walk_stamps.py [f32|f64]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["PLENVEC_LIB"] = os.environ["STAMPS_LIB"]
import numpy as np, torch
import pybullet_pin as P
from plen_ml_walk_amd.vec_env import PlenVecEnv
dt = torch.float64 if len(sys.argv) > 1 and sys.argv[1] == "f64" else torch.float32
dev = torch.device("cuda:0")
W = {k: torch.from_numpy(v).to(dev).to(torch.float32) for k, v in P.SD.items()}
def actor(o):
    h = torch.relu(o @ W["fc1.weight"].T + W["fc1.bias"]); h = torch.relu(h @ W["fc2.weight"].T + W["fc2.bias"]); return torch.tanh(h @ W["fc3.weight"].T + W["fc3.bias"])
n = 4096
env = PlenVecEnv(n, device=dev, dtype=dt); obs = env.reset().to(torch.float32).clone()
g = torch.Generator(device=dev).manual_seed(3)
for _ in range(150):
    a = torch.clamp(actor(obs) + 0.01 * torch.randn(n, 18, generator=g, device=dev), -1, 1)
    _, _, _, info = env.step(a); obs = info["obs"].to(torch.float32)
a = torch.clamp(actor(obs), -1, 1).double().cpu().numpy()
lo = np.array([-1.57, -0.15, -0.95, -0.9, -0.95, -0.8, -1.57, -1.5, -0.75, -0.3, -1.2, -0.4, -1.57, -0.15, -0.2, -1.57, -0.15, -0.2])
hi = np.array([1.57, 1.5, 0.75, 0.3, 1.2, 0.4, 1.57, 0.15, 0.95, 0.9, 0.95, 0.8, 1.57, 1.57, 0.35, 1.57, 1.57, 0.35])
tg = torch.from_numpy((hi - lo) / 2 * a + (hi + lo) / 2)
d = env.debug_substeps(tg, nsub=1, dump=True)
aux = env.get_aux().cpu().numpy()
act = (aux[:, 7] >> 8) & 0xff
nr = np.array([bin(x & 0xf).count("1") for x in act]); nl = np.array([bin(x >> 4).count("1") for x in act])
ph = d[:, 3800:3811].double().cpu().numpy(); pd = np.diff(ph, axis=1).mean(0)
print("phases (mean cycles): " + ", ".join("%s %.0f" % (nm, v) for nm, v in zip(["kin", "body", "S,M,tau", "chol", "v*", "coll+J+Y", "A", "rows", "PGS+apply", "integrate"], pd)) + "  | substep total %.0f" % pd.sum())
if os.environ.get("PHASES_ONLY"):
    env.close(); sys.exit(0)
tot = np.zeros(n)
for itn in (3, 4):
    st = d[:, 3820 + 10 * (itn - 3): 3828 + 10 * (itn - 3)].double().cpu().numpy()
    tot += st[:, 7] - st[:, 0]
tot /= 2
print("mean cycles per iteration over all envs: %.0f" % tot.mean())
for key in sorted(set(zip(nr, nl)), key=lambda k: -((nr == k[0]) & (nl == k[1])).sum())[:12]:
    m = (nr == key[0]) & (nl == key[1])
    print("  (%d, %d): %5.1f %% of envs, %6.0f cycles" % (key[0], key[1], 100 * m.mean(), tot[m].mean()))
env.close()
