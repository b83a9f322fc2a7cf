#!/usr/bin/env python
"""Deterministic-policy evaluation, the GPU form of the reference's plen_bullet/src/walk_eval.py:
load a TD3 checkpoint (the reference's four-file layout, e.g. plen_bullet/models/plen_walk_gazebo_3229999,
or a plain .npz of actor./critic. arrays) and roll the actor -- no exploration noise, actions clipped to
[-max_action, max_action] (walk_eval.py:83-85) -- through N vectorised environments for whole episodes.
No GUI and no 20 Hz sleep (walk_eval.py:78): this reports episode returns / lengths instead."""
import argparse
import json
import os

import numpy as np
import torch

from .td3 import TD3Agent
from .vec_env import PlenVecEnv


def evaluate(policy, num_envs=64, episodes_per_env=1, dtype=torch.float32, device=None, max_steps=None, cfg_overrides=None,
             action_noise=0.0, seed=0):
    """Returns dict(returns=[...], lengths=[...]) over num_envs * episodes_per_env episodes.
    `action_noise` adds N(0, sigma) to the actions: with sigma=0 every env sees the same deterministic episode
    (the env has no randomness of its own), so a small sigma is what gives a distribution."""
    env = PlenVecEnv(num_envs, device=device or policy.device, dtype=dtype, cfg_overrides=cfg_overrides)
    obs = env.reset().to(torch.float32).clone()
    ret = torch.zeros(num_envs, device=env.device)
    length = torch.zeros(num_envs, device=env.device)
    done_eps = torch.zeros(num_envs, dtype=torch.long, device=env.device)
    returns, lengths = [], []
    max_steps = max_steps or env.max_episode_steps * episodes_per_env + 1
    gen = torch.Generator(device=env.device)
    gen.manual_seed(seed)
    for _ in range(max_steps):
        action = policy.select_action_batch(obs)
        if action_noise:
            action = action + action_noise * torch.randn(action.shape, device=env.device, generator=gen)
        action = action.clamp(-policy.max_action, policy.max_action)
        _, reward, done, info = env.step(action)
        counting = done_eps < episodes_per_env
        ret += reward.to(torch.float32) * counting
        length += counting
        ended = (done != 0) & counting
        if bool(ended.any()):
            returns += ret[ended].tolist(); lengths += length[ended].tolist()
            ret = ret * (~ended); length = length * (~ended)
            done_eps += ended.long()
        obs = info["obs"].to(torch.float32).clone()
        if bool((done_eps >= episodes_per_env).all()):
            break
    env.close()
    return dict(returns=returns, lengths=lengths)


def load_policy(path, device=None):
    policy = TD3Agent(26, 18, 1.0, device=device, data_parallel=False)
    if path.endswith(".npz"):
        policy.load_arrays(np.load(path))
    else:
        policy.load(path, load_optimizers=False)
    return policy


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("checkpoint", help="prefix of <prefix>_actor/_critic files, or an .npz with actor./critic. arrays")
    ap.add_argument("--envs", type=int, default=64)
    ap.add_argument("--episodes-per-env", type=int, default=1)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--action-noise", type=float, default=0.0)
    a = ap.parse_args(argv)
    policy = load_policy(a.checkpoint)
    res = evaluate(policy, a.envs, a.episodes_per_env, torch.float32 if a.dtype == "f32" else torch.float64, action_noise=a.action_noise)
    r, l = np.array(res["returns"]), np.array(res["lengths"])
    print(json.dumps({"episodes": len(r), "mean_return": float(r.mean()), "min_return": float(r.min()), "max_return": float(r.max()),
                      "mean_length": float(l.mean()), "full_length_fraction": float((l >= 500).mean())}))


if __name__ == "__main__":
    main()
