"""Pipelined TD3 trainer with its H collectors on ONE stream (never more than one sub-batch's waves resident, so the update's kernels find
free wave slots) against the default of one stream per collector.   usage: python scripts/gpu_collectors_one_stream_probe.py H same|own [batch]"""
import os
import sys
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch


def main():
    H = int(sys.argv[1])
    mode = sys.argv[2]
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    from plen_ml_walk_amd import vec_env
    from plen_ml_walk_amd.vec_env import PlenVecEnv
    from plen_ml_walk_amd.td3 import ReplayBuffer, TD3Agent
    from plen_ml_walk_amd.train_vec import PipelinedVecTD3Trainer
    s0 = vec_env.worker_stream(dev, 0)
    if mode == "same":
        for h in range(1, H):
            vec_env._WORKER_STREAMS[(0, str(h))] = s0
    n = 4096
    torch.manual_seed(0)
    agent = TD3Agent(26, 18, 1.0, device=dev)
    replay = ReplayBuffer(1000000, device=dev)
    envs = [PlenVecEnv(n // H, device=dev) for _ in range(H)]
    tr = PipelinedVecTD3Trainer(envs, agent, replay, start_timesteps=10000, expl_noise=0.1, batch_size=batch, seed=1000)
    for _ in range(40):
        tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 200
    for _ in range(steps):
        tr.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print("H %d collectors on %s stream(s), batch %d: %.3f ms/step = %.2f M env-steps/s, %.0f grad steps/s" % (H, mode, batch, ms, n / ms / 1e3, 1e3 / ms))


if __name__ == "__main__":
    main()
