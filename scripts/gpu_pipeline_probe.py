"""Where does the pipelined trainer's step time go?  Collectors only (learning off) vs the full loop, per block of 50 steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from plen_ml_walk_amd.vec_env import PlenVecEnv
from plen_ml_walk_amd.td3 import TD3Agent, ReplayBuffer
from plen_ml_walk_amd.train_vec import PipelinedVecTD3Trainer
def run(start, blocks=8, warm=40, seed=0):
    torch.manual_seed(0)
    envs = [PlenVecEnv(2048), PlenVecEnv(2048)]
    agent = TD3Agent(26, 18, 1.0); replay = ReplayBuffer(1000000)
    tr = PipelinedVecTD3Trainer(envs, agent, replay, start_timesteps=start, batch_size=4096, seed=seed)
    for _ in range(warm): tr.step()
    out = []
    for b in range(blocks):
        tr.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): tr.step()
        tr.sync(); torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 50 * 1e3)
    a0, a1 = envs[0].get_aux().float(), envs[1].get_aux().float()
    print("   mean episode step of the envs at the end: %.1f, mean solver iterations %.1f, critic loss %.3g" % (float(a0[:, 2].mean()), float(a0[:, 6].mean()), float(agent.last_critic_loss)))
    for e in envs: e.close()
    return " ".join("%.3f" % x for x in out)
if len(sys.argv) == 1: print("collectors only (uniform actions, no update), ms/step per 50-step block:", run(10 ** 12))
if len(sys.argv) == 1: print("full pipelined loop seed 0:   ", run(10000))
if len(sys.argv) == 1: print("full pipelined loop seed 1000:", run(10000, seed=1000))
def run_b(batch, blocks=4, warm=40):
    torch.manual_seed(0)
    envs = [PlenVecEnv(2048), PlenVecEnv(2048)]
    agent = TD3Agent(26, 18, 1.0); replay = ReplayBuffer(1000000)
    tr = PipelinedVecTD3Trainer(envs, agent, replay, start_timesteps=10000, batch_size=batch, seed=1000)
    for _ in range(warm): tr.step()
    out = []
    for b in range(blocks):
        tr.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): tr.step()
        tr.sync(); torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 50 * 1e3)
    for e in envs: e.close()
    return " ".join("%.3f" % x for x in out)
if len(sys.argv) > 1:
    for b in [int(x) for x in sys.argv[1:]]:
        print("batch", b, run_b(b))
