"""Occupancy cliff probe: ms/step around N = 16 blocks x CU count (one wave per env, 16 waves per CU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from plen_ml_walk_amd.vec_env import PlenVecEnv
p = torch.cuda.get_device_properties(0)
print(p.name, "CUs", p.multi_processor_count, "clock", getattr(p, "clock_rate", None))
def run(n, steps=40):
    env = PlenVecEnv(n); env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    acts = torch.rand(steps + 10, n, 18, device="cuda", generator=g) * 2 - 1
    for t in range(10): env.step(acts[t])
    torch.cuda.synchronize(); env.timing_begin()
    for t in range(steps): env.step(acts[10 + t])
    ms, nl = env.timing_end(); env.close(); return ms / nl
for n in (2048, 3072, 3584, 3840, 3968, 4032, 4096, 4160, 4352, 5120):
    print("N=%5d %.4f ms" % (n, run(n)), flush=True)
