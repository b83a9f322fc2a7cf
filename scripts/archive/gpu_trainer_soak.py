"""Soak of the pipelined TD3 trainer (row-block update, flat Adam): N seconds from the shipped policy, checking every 2000 steps that all parameters,
moments, targets and the critic loss are finite, that the step counters agree, and that no env needed a non-finite reset.
usage: python scripts/gpu_trainer_soak.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from plen_ml_walk_amd.vec_env import PlenVecEnv
from plen_ml_walk_amd.td3 import ReplayBuffer, TD3Agent
from plen_ml_walk_amd.train_vec import PipelinedVecTD3Trainer

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
torch.manual_seed(0)
envs = [PlenVecEnv(2048), PlenVecEnv(2048)]
agent = TD3Agent(26, 18, 1.0)
agent.load_arrays(np.load(os.path.join(ROOT, "tests/golden/policy_3229999.npz")))
agent.actor_target.load_state_dict(agent.actor.state_dict()); agent.critic_target.load_state_dict(agent.critic.state_dict())
replay = ReplayBuffer(1000000)
tr = PipelinedVecTD3Trainer(envs, agent, replay, start_timesteps=4096, expl_noise=0.1, batch_size=4096, seed=0)
t0, bad, checks = time.time(), 0, 0
while time.time() - t0 < budget:
    for _ in range(2000):
        tr.step()
    tr.sync(); torch.cuda.synchronize()
    fz = tr.fused
    tensors = [agent._actor_flat.flat, agent._critic_flat.flat, agent._actor_target_flat.flat, agent._critic_target_flat.flat,
               fz._critic_adam.m, fz._critic_adam.v, fz._actor_adam.m, fz._actor_adam.v, replay.data[:min(replay.size, 200000)]]
    ok = all(bool(torch.isfinite(t).all()) for t in tensors) and bool(torch.isfinite(agent.last_critic_loss))
    ok = ok and float(fz._critic_adam.step_t) == tr.grad_steps and float(fz._actor_adam.step_t) == tr.grad_steps // agent.policy_freq
    nonfinite = sum(int(e.nonfinite_count()) for e in envs)
    checks += 1; bad += 0 if (ok and nonfinite == 0) else 1
    print("t=%5.1fs steps %d grad steps %d critic loss %.3f nonfinite resets %d %s" % (time.time() - t0, tr.env_steps, tr.grad_steps, float(agent.last_critic_loss), nonfinite,
                                                                                     "OK" if ok else "BAD"), flush=True)
stats = tr.episode_stats()
print("trainer soak: %d checks, %d bad; %.2f M env-steps/s; episodes %d, mean return %.1f" % (checks, bad, tr.env_steps / (time.time() - t0) / 1e6, stats["episodes"], stats["mean_return"]))
sys.exit(1 if bad else 0)
