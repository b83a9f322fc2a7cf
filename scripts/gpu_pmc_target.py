"""PMC target: 30 steps of N envs (one launch per step), random actions.  rocprofv3 --pmc ... -- python3 scripts/gpu_pmc_target.py N_ITER N DTYPE"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from plen_ml_walk_amd.vec_env import PlenVecEnv
it = int(sys.argv[1]) if len(sys.argv) > 1 else 50
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dtype = torch.float64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else torch.float32
env = PlenVecEnv(n, dtype=dtype, cfg_overrides=dict(num_iterations=it)); env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
acts = torch.rand(30, n, 18, device="cuda", generator=g) * 2 - 1
for t in range(30): env.step(acts[t])
torch.cuda.synchronize(); env.close()
