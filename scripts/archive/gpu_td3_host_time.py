"""Is the pipelined TD3 loop bound by the host's enqueue rate?  Host time of tr.step() (graph replays + event bookkeeping, no sync) against the
wall time per step including the final synchronise.  usage: python scripts/gpu_td3_host_time.py [batch] [steps]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from plen_ml_walk_amd.vec_env import PlenVecEnv
from plen_ml_walk_amd.td3 import ReplayBuffer, TD3Agent
from plen_ml_walk_amd.train_vec import PipelinedVecTD3Trainer

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = torch.device("cuda:0")
torch.manual_seed(0)
agent = TD3Agent(26, 18, 1.0, device=dev)
replay = ReplayBuffer(1000000, device=dev)
envs = [PlenVecEnv(2048, device=dev) for _ in range(2)]
tr = PipelinedVecTD3Trainer(envs, agent, replay, start_timesteps=10000, expl_noise=0.1, batch_size=batch, seed=1000)
for _ in range(40):
    tr.step()
torch.cuda.synchronize()
for rep in range(3):
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        h0 = time.perf_counter()
        tr.step()
        host += time.perf_counter() - h0
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("batch %d: host enqueue %.3f ms per step (sum of step() calls %.3f), wall incl. final sync %.3f ms per step -> the queues were %.1f ms ahead at the end"
          % (batch, t_enq / steps * 1e3, host / steps * 1e3, t_all / steps * 1e3, (t_all - t_enq) * 1e3), flush=True)
