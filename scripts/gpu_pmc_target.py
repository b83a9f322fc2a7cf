"""PMC target: 20 steps of 4096 envs, random actions, solver iterations from argv (rocprofv3 --pmc -- python3 this N_ITER)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from plen_ml_walk_amd.vec_env import PlenVecEnv
it = int(sys.argv[1]) if len(sys.argv) > 1 else 50
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
env = PlenVecEnv(n, cfg_overrides=dict(num_iterations=it)); env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
acts = torch.rand(30, n, 18, device="cuda", generator=g) * 2 - 1
for t in range(30): env.step(acts[t])
torch.cuda.synchronize(); env.close()
