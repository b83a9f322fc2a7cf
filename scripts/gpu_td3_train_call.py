import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from plen_ml_walk_amd.td3 import TD3Agent, ReplayBuffer
for cache in ("1", "0"):
    os.environ["PLEN_TD3_EAGER_CACHE"] = cache
    torch.manual_seed(1)
    ag = TD3Agent(26, 18, 1.0, device="cuda", data_parallel=False)
    buf = ReplayBuffer(20000, device="cuda")
    buf.add_batch(torch.randn(10000, 26), torch.rand(10000, 18) * 2 - 1, torch.randn(10000, 26), torch.randn(10000), (torch.rand(10000) < 0.02).float())
    for _ in range(30):
        ag.train(buf, 100)
    torch.cuda.synchronize()
    t1, calls = time.perf_counter(), 2000
    for _ in range(calls):
        ag.train(buf, 100)
    torch.cuda.synchronize()
    print("eager cache", cache, ": %.1f us per TD3Agent.train(buf, 100)" % ((time.perf_counter() - t1) / calls * 1e6), "fused", ag._fused is not None and ag._fused.eager_cache)
