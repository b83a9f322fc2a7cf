"""PMC / trace target: 4096 envs driven by the shipped policy + N(0, 0.01) (walking robots), 150 warm-up + 60 steps.  usage: python3 scripts/gpu_walk_target.py f32|f64"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import pybullet_pin as P
from plen_ml_walk_amd.vec_env import PlenVecEnv
dt = torch.float64 if (len(sys.argv) > 1 and sys.argv[1] == "f64") else torch.float32
dev = torch.device("cuda:0")
W = {k: torch.from_numpy(v).to(dev).to(torch.float32) for k, v in P.SD.items()}
def actor(o):
    h = torch.relu(o @ W["fc1.weight"].T + W["fc1.bias"]); h = torch.relu(h @ W["fc2.weight"].T + W["fc2.bias"]); return torch.tanh(h @ W["fc3.weight"].T + W["fc3.bias"])
n = 4096
env = PlenVecEnv(n, device=dev, dtype=dt); obs = env.reset().to(torch.float32).clone()
g = torch.Generator(device=dev).manual_seed(3)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 210):
    a = torch.clamp(actor(obs) + 0.01 * torch.randn(n, 18, generator=g, device=dev), -1, 1)
    _, _, _, info = env.step(a); obs = info["obs"].to(torch.float32)
torch.cuda.synchronize(); env.close()
