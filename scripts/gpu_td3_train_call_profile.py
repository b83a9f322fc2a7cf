"""Where the HOST time of the reference's own call goes: cProfile over TD3Agent.train(buf, 100) (plen_td3.py:119-120), eager, cached small-batch path; and the same loop
timed without the profiler, with and without a device synchronisation per call (host-bound or GPU-bound).
usage: python scripts/gpu_td3_train_call_profile.py -> stdout"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.getcwd())
import torch
from plen_ml_walk_amd.td3 import TD3Agent, ReplayBuffer
torch.manual_seed(1)
ag = TD3Agent(26, 18, 1.0, device="cuda", data_parallel=False)
buf = ReplayBuffer(20000, device="cuda")
buf.add_batch(torch.randn(10000, 26), torch.rand(10000, 18) * 2 - 1, torch.randn(10000, 26), torch.randn(10000), (torch.rand(10000) < 0.02).float())
for _ in range(200):
    ag.train(buf, 100)
torch.cuda.synchronize()
calls = 4000
t0 = time.perf_counter()
for _ in range(calls):
    ag.train(buf, 100)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("issue %.1f us per call (host), %.1f us per call until the GPU is done" % (t_issue / calls * 1e6, t_all / calls * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(calls):
    ag.train(buf, 100)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print("\n".join(l[:170] for l in s.getvalue().splitlines()))
