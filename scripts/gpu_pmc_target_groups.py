"""PMC target for the HEADLINE mode: N envs as G sub-batches on G HIP streams (PlenVecEnvPipelined), 40 steps of random actions.
rocprofv3 --pmc ... -- python3 scripts/gpu_pmc_target_groups.py N G DTYPE"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from plen_ml_walk_amd.vec_env import PlenVecEnvPipelined
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dtype = torch.float64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else torch.float32
env = PlenVecEnvPipelined(n, groups=g, dtype=dtype); env.reset()
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
acts = torch.rand(40, n, 18, device="cuda", generator=gen) * 2 - 1
for t in range(40):
    env.step_async(acts[t])
env.sync(); torch.cuda.synchronize(); env.close()
