"""What the pipelined TD3 loop (BASELINE.json configs[2]) costs WITHOUT its learner: the two collectors acting with the freshly initialised policy + N(0, 0.1) -- robots
that stand, a heavier env workload than uniform random actions -- against the same loop with the learner on.   usage: python scripts/gpu_td3_collectors_only.py"""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from plen_ml_walk_amd.vec_env import PlenVecEnv
from plen_ml_walk_amd.td3 import ReplayBuffer, TD3Agent
from plen_ml_walk_amd.train_vec import PipelinedVecTD3Trainer
dev = torch.device("cuda", 0)
out = {}
for learning in ((True,) if os.environ.get("ONLY_LEARNING") == "1" else (True, False)):
    n, H = 4096, 2
    torch.manual_seed(0)
    agent = TD3Agent(26, 18, 1.0, device=dev)
    replay = ReplayBuffer(1000000, device=dev)
    envs = [PlenVecEnv(n // H, device=dev) for _ in range(H)]
    tr = PipelinedVecTD3Trainer(envs, agent, replay, start_timesteps=10000, expl_noise=0.1, batch_size=4096, seed=1000)
    tr.learning = learning
    for _ in range(60):
        tr.step()
    tr.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 1500
    for _ in range(steps):
        tr.step()
    tr.sync(); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    out["learning" if learning else "collectors_only"] = {"ms_per_step": ms, "env_steps_per_s": n / ms * 1e3, "episodes": tr.episode_stats()}
    print("learner %-3s: %.3f ms per vector step of %d envs = %.2f M env-steps/s   %s" % ("on" if learning else "off", ms, n, n / ms / 1e3, out["learning" if learning else "collectors_only"]["episodes"]), flush=True)
    for e in envs:
        e.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
if os.environ.get("ONLY_LEARNING") != "1":
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r05_td3_collectors_only.json"), "w"), indent=1)
