"""A few large-batch TD3 updates through the block kernels (csrc/td3_block.hip), nothing else: the workload of scripts/gpu_td3_block_pmc.sh / kernel traces.
usage: python scripts/gpu_td3_block_run.py [batch] [updates]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from plen_ml_walk_amd import td3 as T
from plen_ml_walk_amd.td3_fused import FusedTD3
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
torch.manual_seed(0)
ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
fz = FusedTD3(ag, seed=1, rows=os.environ.get("PLEN_TD3_ROWS", "0") == "1", team=False, block=True)
fz.enable_flat_adam()
data = torch.randn(100000, 72, device="cuda"); data[:, 70] = torch.rand(100000, device="cuda"); data[:, 71] = (torch.rand(100000, device="cuda") > 0.02).float()
tot = torch.tensor(100000, dtype=torch.long, device="cuda")
for k in range(n):
    fz.update(data, B, with_policy=(k % 2 == 1), all_reduce=False, total=tot)
torch.cuda.synchronize()
print("loss", float(ag.last_critic_loss))
