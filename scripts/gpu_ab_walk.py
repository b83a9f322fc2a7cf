"""A/B of libplenvec builds on a WALKING workload in either arithmetic (the shipped policy + N(0, 0.01) in the loop, 4096 envs, auto-reset on, one launch per step; float64
actor in torch on the device): the contact-rich counterpart of scripts/gpu_ab64.py's random flailing.  usage: [AB_DTYPE=f32] python scripts/gpu_ab_walk.py libA.so libB.so ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COST = os.environ.get("AB_COSTS", "").split(";")
CHILD = r'''
import sys, time, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import pybullet_pin as P
from plen_ml_walk_amd.vec_env import PlenVecEnv
dt = torch.float64 if sys.argv[1] == "f64" else torch.float32
dev = torch.device("cuda:0")
W = {k: torch.from_numpy(v).to(dev).to(torch.float32) for k, v in P.SD.items()}
def actor(o):
    h = torch.relu(o @ W["fc1.weight"].T + W["fc1.bias"]); h = torch.relu(h @ W["fc2.weight"].T + W["fc2.bias"]); return torch.tanh(h @ W["fc3.weight"].T + W["fc3.bias"])
n = 4096
env = PlenVecEnv(n, device=dev, dtype=dt); obs = env.reset().to(torch.float32).clone()
g = torch.Generator(device=dev).manual_seed(3)
def step():
    global obs
    a = torch.clamp(actor(obs) + 0.01 * torch.randn(n, 18, generator=g, device=dev), -1, 1)
    _, _, _, info = env.step(a); obs = info["obs"].to(torch.float32)
for _ in range(150): step()
torch.cuda.synchronize(); env.timing_begin(); t0 = time.perf_counter()
for _ in range(300): step()
ms, nl = env.timing_end(); torch.cuda.synchronize(); wall = time.perf_counter() - t0
print("kernel %%.4f ms per 4096-env launch, loop %%.4f ms per step" %% (ms / nl, wall / 300 * 1e3))
''' % (ROOT, os.path.join(ROOT, "tests"))
if __name__ == "__main__":
    dtype = os.environ.get("AB_DTYPE", "f64")
    for rnd in range(2):
        for lib in sys.argv[1:]:
            for cost in COST:                  # AB_COSTS="4100,78,50,37;4800,78,41,26": PLENVEC_COST settings to compare for every build ("" = the build's default)
                env = dict(os.environ, PLENVEC_LIB=os.path.join(ROOT, "plen_ml_walk_amd", "csrc", "variants", lib))
                if cost:
                    env["PLENVEC_COST"] = cost
                out = subprocess.run([sys.executable, "-c", CHILD, dtype], env=env, capture_output=True, text=True, timeout=300)
                print("%-12s %-18s %s %s %s" % (lib, cost, dtype, out.stdout.strip(), out.stderr.strip()[-300:] if out.returncode else ""), flush=True)
