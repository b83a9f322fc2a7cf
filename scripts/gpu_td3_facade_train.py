"""TD3Agent.train(replay_buffer, 100) -- the reference's call (plen_td3.py:119-120) on the reference's surface -- timed eagerly (no hipGraph, one Python call
per iteration, as a reference user's loop issues it): the fused iteration train() takes on a HIP device vs the autograd iteration (fused_train = False).
usage: python scripts/gpu_td3_facade_train.py [batch]   -> gpurun_out/r04_td3_facade_train.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from plen_ml_walk_amd import td3 as T

B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
out = {}
for name, fused in (("autograd", False), ("fused", None)):
    torch.manual_seed(0)
    ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
    ag.fused_train = fused
    buf = T.ReplayBuffer(100000)
    buf.add_batch(torch.randn(50000, 26), torch.rand(50000, 18) * 2 - 1, torch.randn(50000, 26), torch.randn(50000), (torch.rand(50000) < 0.02).float())
    for _ in range(50):
        ag.train(buf, B)
    torch.cuda.synchronize()
    n = 1000
    t0 = time.perf_counter()
    for _ in range(n):
        ag.train(buf, B)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / n * 1e6
    out[name] = {"us_per_train_call": us, "calls_per_s": 1e6 / us, "batch": B, "critic_loss": float(ag.last_critic_loss)}
    print("%-9s batch %d  %8.1f us per train() call  %8.0f calls/s   loss %.4f" % (name, B, us, 1e6 / us, float(ag.last_critic_loss)), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_td3_facade_train.json"), "w"), indent=1)
