"""Where a substep's cycles go in the state distributions bench.py times (not in a standing robot): 4096 envs are driven for 150 steps by random actions or by the shipped
walking policy, then ONE debug substep with shader-clock stamps (-DPGS_STAMPS build) is taken from every env's state: mean cycles per phase, and per section of solver
iterations 3 and 4, over all envs (a launch lasts as long as its slowest waves, but the mean is what the SIMDs' time is spent on).
usage: python scripts/gpu_rollout_stamps.py [f32|f64] [random|walking]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from plen_ml_walk_amd.build import build_variant
os.environ["PLENVEC_LIB"] = os.environ.get("STAMPS_LIB") or build_variant("stamps", ["-DPGS_STAMPS"])
import numpy as np, torch
import pybullet_pin as P
from plen_ml_walk_amd.vec_env import PlenVecEnv
dt = torch.float64 if len(sys.argv) > 1 and sys.argv[1] == "f64" else torch.float32
what = sys.argv[2] if len(sys.argv) > 2 else "random"
dev = torch.device("cuda:0")
W = {k: torch.from_numpy(v).to(dev).to(torch.float32) for k, v in P.SD.items()}
def actor(o):
    h = torch.relu(o @ W["fc1.weight"].T + W["fc1.bias"]); h = torch.relu(h @ W["fc2.weight"].T + W["fc2.bias"]); return torch.tanh(h @ W["fc3.weight"].T + W["fc3.bias"])
n = 4096
env = PlenVecEnv(n, device=dev, dtype=dt); obs = env.reset().to(torch.float32).clone()
g = torch.Generator(device=dev).manual_seed(3)
def act():
    return (torch.rand(n, 18, generator=g, device=dev) * 2 - 1) if what == "random" else torch.clamp(actor(obs) + 0.01 * torch.randn(n, 18, generator=g, device=dev), -1, 1)
for _ in range(150):
    _, _, _, info = env.step(act()); obs = info["obs"].to(torch.float32)
a = act().double().cpu().numpy()
lo = np.array([-1.57, -0.15, -0.95, -0.9, -0.95, -0.8, -1.57, -1.5, -0.75, -0.3, -1.2, -0.4, -1.57, -0.15, -0.2, -1.57, -0.15, -0.2])
hi = np.array([1.57, 1.5, 0.75, 0.3, 1.2, 0.4, 1.57, 0.15, 0.95, 0.9, 0.95, 0.8, 1.57, 1.57, 0.35, 1.57, 1.57, 0.35])
d = env.debug_substeps(torch.from_numpy((hi - lo) / 2 * a + (hi + lo) / 2), nsub=1, dump=True)
aux = env.get_aux().cpu().numpy()
occ = (aux[:, 7] >> 8) & 0xff
npt = np.array([bin(x).count("1") for x in occ])
ph = np.diff(d[:, 3800:3811].double().cpu().numpy(), axis=1)
names = ["kinematics", "body_dyn+subtree", "S,M,tau", "cholesky", "v* solve", "collision+J+Y", "A build", "rows setup", "PGS+apply", "integrate"]
print("%s, %s actions: mean contact points %.2f; substep total %.0f cycles (mean over envs), slowest env %.0f" % (dt, what, npt.mean(), ph.sum(1).mean(), ph.sum(1).max()))
for k, nme in enumerate(names): print("   %-18s %8.0f  %5.1f %%" % (nme, ph[:, k].mean(), 100 * ph[:, k].mean() / ph.sum(1).mean()))
es = np.diff(d[:, 3850:3857].double().cpu().numpy(), axis=1)
print("inside collision+J+Y (mean over envs):")
for k, nme in enumerate(["foot manifolds", "foot Jacobians", "box near test (+ rare path)", "J rows -> regs", "port velocities", "back subst + store"]): print("   %-28s %8.0f" % (nme, es[:, k].mean()))
lent = (aux[:, 7] & 0xff) != 0
print("   envs with a contact slot lent to a box corner: %.1f %%; their box section %.0f cycles, the others' %.0f" % (100 * lent.mean(), es[lent, 2].mean() if lent.any() else 0, es[~lent, 2].mean()))
its = d[:, 3700].cpu().numpy()
print("solver iterations per substep: mean %.1f, %.1f %% run all %d" % (its.mean(), 100 * (its >= its.max()).mean(), its.max()))
sec = ["motors(+limits)", "normals", "tors bounds", "spin rows", "roll rows", "cone pairs", "wave max"]
tot = np.zeros((n, 7))
for itn in (3, 4):
    st = d[:, 3820 + 10 * (itn - 3): 3828 + 10 * (itn - 3)].double().cpu().numpy()
    tot += np.diff(st, axis=1) / 2
print("solver iteration (mean of iterations 3 and 4 over envs): %.0f cycles" % tot.sum(1).mean())
for k, nme in enumerate(sec): print("   %-18s %8.0f  %5.1f %%" % (nme, tot[:, k].mean(), 100 * tot[:, k].mean() / tot.sum(1).mean()))
env.close()
