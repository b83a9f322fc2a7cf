#!/usr/bin/env python
"""Single-environment TD3 driver with the reference's control flow (plen_bullet/src/plen_td3.py:16-158):
same hyper-parameters, warm-up, exploration noise, done_bool masking of the time limit, checkpoint
cadence and file names.  It exists to show the surfaces are drop-in; train_vec.py is the fast path."""
import os

import numpy as np
import torch

from .td3 import ReplayBuffer, TD3Agent
from . import plen_env  # noqa: F401  (registers PlenWalkEnv-v1)
from . import gym_compat as gym


def main(max_timesteps=4e6, start_timesteps=1e4, eval_freq=1e4, out_dir=None, seed=0, expl_noise=0.1, batch_size=100,
         save_model=True, quiet=False, policy_num=0, buffer_number=0):
    env_name = "PlenWalkEnv-v1"
    file_name = "plen_walk_gazebo_"
    my_path = out_dir or os.path.abspath(os.path.dirname(__file__))
    results_path = os.path.join(my_path, "../results")
    models_path = os.path.join(my_path, "../models")
    for p in (results_path, models_path):
        if not os.path.exists(p):
            os.makedirs(p)
    env = gym.make(env_name, render=False)
    env.seed(seed)
    torch.manual_seed(seed)
    np.random.seed(seed)
    state_dim = env.observation_space.shape[0]
    action_dim = env.action_space.shape[0]
    max_action = float(env.action_space.high[0])
    policy = TD3Agent(state_dim, action_dim, max_action)
    if os.path.exists(models_path + "/" + "plen_walk_gazebo_" + str(policy_num) + "_critic"):     # plen_td3.py:57-62
        if not quiet:
            print("Loading Existing Policy")
        policy.load(models_path + "/" + "plen_walk_gazebo_" + str(policy_num))
    replay_buffer = ReplayBuffer()
    if os.path.exists(replay_buffer.buffer_path + "/" + "replay_buffer_" + str(buffer_number) + '.data'):   # plen_td3.py:64-69
        if not quiet:
            print("Loading Replay Buffer " + str(buffer_number))
        replay_buffer.load(buffer_number)
    evaluations = []
    state = env.reset()
    done = False
    episode_reward = 0
    episode_timesteps = 0
    episode_num = 0
    for t in range(int(max_timesteps)):
        episode_timesteps += 1
        if t < start_timesteps:
            action = env.action_space.sample()
        else:
            action = np.clip((policy.select_action(np.array(state)) + np.random.normal(0, max_action * expl_noise, size=action_dim)),
                             -max_action, max_action)
        next_state, reward, done, _ = env.step(action)
        done_bool = float(done) if episode_timesteps < env._max_episode_steps else 0
        replay_buffer.add((state, action, next_state, reward, done_bool))
        state = next_state
        episode_reward += reward
        if t >= start_timesteps:
            policy.train(replay_buffer, batch_size)
        if done:
            state, done = env.reset(), False
            evaluations.append(episode_reward)
            if not quiet:
                print("Total T: {} Episode Num: {} Episode T: {} Reward: {}".format(t + 1, episode_num, episode_timesteps, episode_reward))
            episode_reward = 0
            episode_timesteps = 0
            episode_num += 1
        if (t + 1) % eval_freq == 0:
            np.save(results_path + "/" + str(file_name), evaluations)
            if save_model:
                policy.save(models_path + "/" + str(file_name) + str(t))
    env.close()
    return evaluations


if __name__ == '__main__':
    main()
