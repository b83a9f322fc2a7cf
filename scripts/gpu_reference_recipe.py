"""TD3 at the REFERENCE'S update-to-data recipe in this simulator: plen_td3.py:21-30, 83-157 -- start_timesteps 1e4 of
uniform random actions, exploration N(0, 0.1), ONE train() of batch 100 per env-step (td3.py:259-356), replay 1e6, seeds 0 -- and the history
of TRAINING-episode returns the reference logs (plen_td3.py:122-133 `evaluations.append(episode_reward)`, saved as results/plen_walk_gazebo_.npy),
compared block by block with tests/golden/ref_training_log_summary.npz (the reference's own 24 832-episode history: first 1000 episodes
-190.8, last 1000 +50.4, max 328).

n envs step together and n updates follow (same update-to-data ratio, same batch; the reference interleaves them one by one).
usage: python scripts/gpu_reference_recipe.py [max_env_steps] [wall_budget_s] [n_envs] [f32|f64] [seed]
writes gpurun_out/r04_reference_recipe_curve[_seed<k>_n<envs>].json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from plen_ml_walk_amd.vec_env import PlenVecEnv
from plen_ml_walk_amd.td3 import ReplayBuffer, TD3Agent
from plen_ml_walk_amd.train_vec import GraphedVecTD3Trainer
from plen_ml_walk_amd.walk_eval import evaluate

max_steps = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3250000
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 1500.0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dtype = torch.float64 if (len(sys.argv) > 4 and sys.argv[4] == "f64") else torch.float32
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 0
torch.manual_seed(seed); np.random.seed(seed)
env = PlenVecEnv(n, dtype=dtype)
agent = TD3Agent(26, 18, 1.0)
replay = ReplayBuffer(1000000)
tr = GraphedVecTD3Trainer(env, agent, replay, start_timesteps=10000, expl_noise=0.1, batch_size=100, updates_per_step=n, seed=seed)
ep_ret = np.zeros(n); ep_len = np.zeros(n, dtype=np.int64)
returns, lengths, at_step = [], [], []
evals = []
t0 = time.time()
next_eval = 0
out_path = os.path.join(ROOT, "gpurun_out", "r04_reference_recipe_curve.json" if (seed == 0 and n == 16) else "r04_reference_recipe_curve_seed%d_n%d.json" % (seed, n))
os.makedirs(os.path.dirname(out_path), exist_ok=True)
ref = np.load(os.path.join(ROOT, "tests", "golden", "ref_training_log_summary.npz"))


def dump(final=False):
    r = np.array(returns); ln = np.array(lengths)
    blocks = [float(r[i:i + 1000].mean()) for i in range(0, len(r) - 999, 1000)]
    res = dict(recipe=dict(envs=n, seed=seed, env_dtype=str(dtype), start_timesteps=10000, expl_noise=0.1, batch=100, updates_per_env_step=1.0, replay=1000000,
                           policy_freq=2, note="n envs step together, then n updates of batch 100 (same update-to-data ratio as plen_td3.py:119-120)"),
               wall_s=round(time.time() - t0, 1), env_steps=int(tr.env_steps), grad_steps=int(tr.grad_steps), episodes=len(returns),
               training_episode_return_block_means_1000=blocks,
               training_episode_length_block_means_1000=[float(ln[i:i + 1000].mean()) for i in range(0, len(ln) - 999, 1000)],
               first1000_mean=float(r[:1000].mean()) if len(r) >= 1000 else None, last1000_mean=float(r[-1000:].mean()) if len(r) >= 1000 else None,
               max_return=float(r.max()) if len(r) else None, min_return=float(r.min()) if len(r) else None,
               quantiles_last1000={q: float(np.percentile(r[-1000:], q)) for q in (5, 25, 50, 75, 95)} if len(r) >= 1000 else None,
               reference=dict(episodes=int(ref["episodes"]), env_steps="~3.25 M (checkpoints 3179999..3249999)", block_means_1000=[float(x) for x in ref["block_means_1000"]],
                              first1000_mean=float(ref["block_means_1000"][0]), last1000_mean=float(ref["last1000_mean"]), max_return=float(ref["max_return"]), min_return=float(ref["min_return"])),
               deterministic_evaluations=evals, final=final)
    json.dump(res, open(out_path, "w"), indent=1)


while tr.env_steps < max_steps and time.time() - t0 < budget:
    tr.step()
    r = env._reward.detach().to("cpu", torch.float64).numpy(); d = env._done.cpu().numpy()
    ep_ret += r; ep_len += 1
    for e in np.nonzero(d)[0]:
        returns.append(float(ep_ret[e])); lengths.append(int(ep_len[e])); at_step.append(int(tr.env_steps))
        ep_ret[e] = 0.0; ep_len[e] = 0
    if tr.env_steps >= next_eval:
        torch.cuda.synchronize()
        ev = evaluate(agent, num_envs=128, episodes_per_env=1, action_noise=0.0, seed=1)
        row = dict(env_steps=int(tr.env_steps), grad_steps=int(tr.grad_steps), wall_s=round(time.time() - t0, 1), mean_return=float(np.mean(ev["returns"])),
                   mean_length=float(np.mean(ev["lengths"])), full_length_fraction=float((np.array(ev["lengths"]) >= 500).mean()),
                   train_episodes=len(returns), train_return_last200=float(np.mean(returns[-200:])) if returns else None)
        evals.append(row); print(json.dumps(row), flush=True)
        next_eval += 100000
        dump()
dump(final=True)
print("done", tr.env_steps, tr.grad_steps, len(returns), round(time.time() - t0, 1))
env.close()
