#!/usr/bin/env python
"""Single-environment TD3 training, drop-in for the reference's plen_bullet/src/plen_td3.py.

Behaviour kept (reference line numbers): hyper-parameters and seeds (:19-48), optional pick-up of an existing policy / replay
buffer (:57-69), uniform random actions during the first `start_timesteps` steps, then actor + N(0, 0.1) noise clipped to the
action range (:96-104), the replay `done` flag masking the 500-step time limit (:109-110), one TD3 update per environment step
once warmed up (:118-119), episode bookkeeping and reset (:122-133), `results/plen_walk_gazebo_.npy` and the four-file
checkpoint `models/plen_walk_gazebo_<t>_*` every `eval_freq` steps (:136-155).  It exists to show that the surfaces are drop-in;
`train_vec.py` is the fast path (thousands of environments per GPU)."""
import os

import numpy as np
import torch

from . import gym_compat as gym
from . import plen_env  # noqa: F401  (registers PlenWalkEnv-v1)
from .td3 import ReplayBuffer, TD3Agent

ENV_ID = "PlenWalkEnv-v1"
RUN_NAME = "plen_walk_gazebo_"


class _Run(object):
    """Directories, environment, agent and buffer of one training run."""

    def __init__(self, out_dir, seed, policy_num, buffer_number, quiet):
        root = out_dir or os.path.abspath(os.path.dirname(__file__))
        self.results_dir = os.path.join(root, "../results")
        self.models_dir = os.path.join(root, "../models")
        for d in (self.results_dir, self.models_dir):
            os.makedirs(d, exist_ok=True)
        self.quiet = quiet
        self.env = gym.make(ENV_ID, render=False)
        self.env.seed(seed)
        torch.manual_seed(seed)
        np.random.seed(seed)
        self.n_obs = self.env.observation_space.shape[0]
        self.n_act = self.env.action_space.shape[0]
        self.act_limit = float(self.env.action_space.high[0])
        self.agent = TD3Agent(self.n_obs, self.n_act, self.act_limit)
        self.buffer = ReplayBuffer()
        self._pick_up(policy_num, buffer_number)

    def checkpoint_prefix(self, tag):
        return self.models_dir + "/" + RUN_NAME + str(tag)

    def _pick_up(self, policy_num, buffer_number):
        if os.path.exists(self.checkpoint_prefix(policy_num) + "_critic"):
            self.say("Loading Existing Policy")
            self.agent.load(self.checkpoint_prefix(policy_num))
        if os.path.exists(self.buffer.buffer_path + "/replay_buffer_" + str(buffer_number) + ".data"):
            self.say("Loading Replay Buffer " + str(buffer_number))
            self.buffer.load(buffer_number)

    def say(self, msg):
        if not self.quiet:
            print(msg)

    def behaviour_action(self, obs, exploring, noise_scale):
        if exploring:
            return self.env.action_space.sample()
        greedy = self.agent.select_action(np.array(obs))
        noisy = greedy + np.random.normal(0, self.act_limit * noise_scale, size=self.n_act)
        return np.clip(noisy, -self.act_limit, self.act_limit)


def main(max_timesteps=4e6, start_timesteps=1e4, eval_freq=1e4, out_dir=None, seed=0, expl_noise=0.1, batch_size=100,
         save_model=True, quiet=False, policy_num=0, buffer_number=0):
    run = _Run(out_dir, seed, policy_num, buffer_number, quiet)
    env, horizon = run.env, run.env._max_episode_steps
    returns = []                       # one entry per finished episode: what the reference stores as "evaluations"
    obs = env.reset()
    ep_return, ep_len, ep_index = 0, 0, 0
    for t in range(int(max_timesteps)):
        ep_len += 1
        action = run.behaviour_action(obs, exploring=t < start_timesteps, noise_scale=expl_noise)
        obs_next, reward, done, _ = env.step(action)
        terminal_for_replay = float(done) if ep_len < horizon else 0      # a time-limit ending is not a terminal state
        run.buffer.add((obs, action, obs_next, reward, terminal_for_replay))
        obs = obs_next
        ep_return += reward
        if t >= start_timesteps:
            run.agent.train(run.buffer, batch_size)
        if done:
            returns.append(ep_return)
            run.say("Total T: {} Episode Num: {} Episode T: {} Reward: {}".format(t + 1, ep_index, ep_len, ep_return))
            obs = env.reset()
            ep_return, ep_len, ep_index = 0, 0, ep_index + 1
        if (t + 1) % eval_freq == 0:
            np.save(run.results_dir + "/" + RUN_NAME, returns)
            if save_model:
                run.agent.save(run.checkpoint_prefix(t))
    env.close()
    return returns


if __name__ == '__main__':
    main()
