"""Steady-state schedule of the pipelined TD3 trainer read from device-clock stamps inside its graphs (no profiler: rocprofv3's kernel trace
serialises dispatches and slows the host enough to change the picture).   usage: python scripts/gpu_td3_timeline.py [batch] [rows] [start_timesteps]"""
import os
import sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    start = int(sys.argv[3]) if len(sys.argv) > 3 else 10000          # start_timesteps: huge = collectors only (uniform random actions, no update)
    dev = torch.device("cuda", 0)
    from plen_ml_walk_amd.vec_env import PlenVecEnv
    from plen_ml_walk_amd.td3 import ReplayBuffer, TD3Agent
    from plen_ml_walk_amd.train_vec import PipelinedVecTD3Trainer
    n, H = 4096, 2
    torch.manual_seed(0)
    agent = TD3Agent(26, 18, 1.0, device=dev)
    replay = ReplayBuffer(1000000, device=dev)
    envs = [PlenVecEnv(n // H, device=dev) for _ in range(H)]
    tr = PipelinedVecTD3Trainer(envs, agent, replay, start_timesteps=start, expl_noise=0.1, batch_size=batch, seed=1000)
    tl = tr.enable_timeline(64)
    for _ in range(64 * 3 + 7):
        tr.step()
    tr.sync()
    torch.cuda.synchronize()
    t = tl.cpu().numpy().astype("int64")
    order = t[:, 4 * H].argsort()
    t = t[order]
    t = t[-rows - 2:-2]                      # the last complete steps (the update of the final rows may not have run)
    t0 = t[0].min()
    us = (t - t0) / 100.0
    names = ["c%d %s" % (h, k) for h in range(H) for k in ("start", "env>", "env<", "end")] + ["upd start", "sampled", "targets", "critic>", "critic<", "upd end"]
    print("device-clock timeline (us from the first stamp shown); one row per vector step")
    print("  ".join("%9s" % x for x in names))
    for r in us:
        print("  ".join("%9.1f" % x for x in r))
    d = us[1:] - us[:-1]
    print("period (update start to update start): %s" % " ".join("%.0f" % x for x in d[:, 4 * H]))
    u = us[:, 4 * H:]
    print("update phases (means, us): sample %.0f, targets %.0f, critic forward %.0f, critic backward %.0f, optimiser (+ policy) %.0f" % tuple((u[:, k + 1] - u[:, k]).mean() for k in range(5)))
    print("env kernel c0 %.0f us, c1 %.0f us, update %.0f us (means)" % ((us[:, 2] - us[:, 1]).mean(), (us[:, 6] - us[:, 5]).mean(), (us[:, 4 * H + 5] - us[:, 4 * H]).mean()))


if __name__ == "__main__":
    main()
