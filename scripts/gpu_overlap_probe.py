"""Can the fused TD3 update run BESIDE an env launch?  Time (a) N-env steps alone, (b) updates alone, (c) both on two streams (hipGraphs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from plen_ml_walk_amd.vec_env import PlenVecEnv
from plen_ml_walk_amd.td3 import TD3Agent, ReplayBuffer
from plen_ml_walk_amd.td3_fused import FusedTD3
dev = torch.device("cuda", 0)
B = 4096
agent = TD3Agent(26, 18, 1.0); replay = ReplayBuffer(200000); fz = FusedTD3(agent)
agent.actor_optimizer = torch.optim.Adam(agent.actor.parameters(), lr=3e-4, capturable=True, fused=True)
agent.critic_optimizer = torch.optim.Adam(agent.critic.parameters(), lr=3e-4, capturable=True, fused=True)
replay.data[:100000].normal_(); replay.data[:100000, 71].fill_(1.0)
total = torch.tensor(100000, device=dev)
def graph_of(fn, stream):
    with torch.cuda.stream(stream):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream): fn()
    return g
for n in (2048, 4096):
    env = PlenVecEnv(n); env.reset()
    acts = torch.rand(n, 18, device=dev) * 2 - 1
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ga = graph_of(lambda: env.step(acts), sa)
    it = [0]
    def upd():
        fz.update(replay.data, B, True, all_reduce=False, total=total)
    gb = graph_of(upd, sb)
    def run(do_a, do_b, K=200):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(K):
            if do_a:
                with torch.cuda.stream(sa): ga.replay()
            if do_b:
                with torch.cuda.stream(sb): gb.replay()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
    a, b, c = run(True, False), run(False, True), run(True, True)
    print("envs %d: env step alone %.3f ms | policy update alone %.3f ms | both on two streams %.3f ms per pair (sum %.3f, max %.3f)" % (n, a, b, c, a + b, max(a, b)))
    env.close()
