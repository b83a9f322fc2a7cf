"""End-to-end demonstration: TD3 from scratch on 4096 vectorised envs (hipGraph-captured loop) for a fixed wall-clock budget,
evaluating the deterministic actor every few thousand iterations.  Writes gpurun_out/train_curve.json.
usage: python scripts/gpu_train_demo.py [seconds] [updates_per_step] [init.npz] [sync|pipelined]   (init.npz: actor./critic. arrays, e.g. tests/golden/policy_3229999.npz;
pipelined = train_vec.PipelinedVecTD3Trainer, the actor / learner overlap on three streams)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from plen_ml_walk_amd.vec_env import PlenVecEnv
from plen_ml_walk_amd.td3 import ReplayBuffer, TD3Agent
from plen_ml_walk_amd.train_vec import GraphedVecTD3Trainer, PipelinedVecTD3Trainer
from plen_ml_walk_amd.walk_eval import evaluate

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
ups = int(sys.argv[2]) if len(sys.argv) > 2 else 1
init = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else None
sched = sys.argv[4] if len(sys.argv) > 4 else "sync"
torch.manual_seed(0)
envs = [PlenVecEnv(2048), PlenVecEnv(2048)] if sched == "pipelined" else [PlenVecEnv(4096)]
env = envs[0]
agent = TD3Agent(26, 18, 1.0)
replay = ReplayBuffer(1000000)
if init:
    agent.load_arrays(np.load(os.path.join(ROOT, init)))
    agent.actor_target.load_state_dict(agent.actor.state_dict()); agent.critic_target.load_state_dict(agent.critic.state_dict())
if sched == "pipelined":
    tr = PipelinedVecTD3Trainer(envs, agent, replay, start_timesteps=(4096 if init else 100000), expl_noise=0.1, batch_size=4096, seed=0)
else:
    tr = GraphedVecTD3Trainer(env, agent, replay, start_timesteps=(4096 if init else 100000), expl_noise=0.1, batch_size=4096, updates_per_step=ups, seed=0)
curve = []
best = dict(mean=-1e9)
def ev(tag):
    torch.cuda.synchronize()
    r = evaluate(agent, num_envs=128, episodes_per_env=1, action_noise=0.01, seed=1)
    ret, ln = np.array(r["returns"]), np.array(r["lengths"])
    row = dict(wall_s=round(time.time() - t0, 1), env_steps=int(tr.env_steps), grad_steps=int(tr.grad_steps), mean_return=float(ret.mean()),
               median_return=float(np.median(ret)), mean_length=float(ln.mean()), full_length_fraction=float((ln >= 500).mean()))
    curve.append(row); print(tag, json.dumps(row), flush=True)
    if row["mean_return"] > best["mean"]:          # keep the best actor seen (TD3 with this schedule does not stay at its peak)
        best.update(mean=row["mean_return"], row=row, actor={"actor." + k: v.detach().cpu().numpy().copy() for k, v in agent.actor.state_dict().items()},
                    critic={"critic." + k: v.detach().cpu().numpy().copy() for k, v in agent.critic.state_dict().items()})
t0 = time.time()
ev("init")
it = 0
while time.time() - t0 < budget:
    for _ in range(2000):
        tr.step()
    it += 2000
    ev("it%d" % it)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(dict(envs=4096, batch=4096, updates_per_step=ups, schedule=sched, start_timesteps=(4096 if init else 100000), budget_s=budget, curve=curve), open(os.path.join(ROOT, "gpurun_out", ("train_curve_finetune.json" if init else "train_curve.json")), "w"), indent=1)
if best.get("actor"):
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "best_policy.npz"), **best["actor"], **best["critic"])
    print("best", json.dumps(best["row"]))
for e in envs:
    e.close()
