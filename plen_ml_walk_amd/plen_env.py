"""PlenWalkEnv: the reference's single-environment gym surface on top of the HIP library.

Mirrors plen_bullet/src/plen_bullet/plen_env.py of the reference: registry id "PlenWalkEnv-v1" with
max_episode_steps=500 (:15-19), `PlenWalkEnv(render=False, realtime=False, joint_act=False)` (:34),
`reset() -> float64 (26,)` (:558), `step(action) -> (obs float64 (26,), np.float64, bool, {})` (:638),
`action_space` / `observation_space` (:142-144, :246-263), `env_ranges` / `real_ranges` /
`joint_names` / `movingJoints`, `close()` (:1095).  One env = one wavefront of libplenvec's kernel,
computed in float64 (the reference and PyBullet are double precision); use PlenVecEnv for throughput.
"""
import os
import numpy as np
import torch

from . import gym_compat
from .gym_compat import Box, Env, register
from .vec_env import PlenVecEnv

ENV_RANGES = [   # plen_env.py:148-167
    [-1.57, 1.57], [-0.15, 1.5], [-0.95, 0.75], [-0.9, 0.3], [-0.95, 1.2], [-0.8, 0.4],
    [-1.57, 1.57], [-1.5, 0.15], [-0.75, 0.95], [-0.3, 0.9], [-1.2, 0.95], [-0.4, 0.8],
    [-1.57, 1.57], [-0.15, 1.57], [-0.2, 0.35], [-1.57, 1.57], [-0.15, 1.57], [-0.2, 0.35]]
REAL_RANGES = [  # plen_env.py:170-189
    [-1.57, 1.57], [-0.15, 1.5], [-0.95, 1.2], [-1.0, 1.57], [-0.95, 1.2], [-0.8, 0.4],
    [-1.57, 1.57], [-1.5, 0.15], [-1.2, 0.95], [-1.0, 1.57], [-1.2, 1.2], [-0.4, 0.8],
    [-1.57, 1.57], [-0.15, 1.57], [-0.2, 0.35], [-1.57, 1.57], [-0.15, 1.57], [-0.2, 0.35]]
JOINT_NAMES = [  # plen_env.py:547-554
    'rb_servo_r_hip', 'r_hip_r_thigh', 'r_thigh_r_knee', 'r_knee_r_shin', 'r_shin_r_ankle', 'r_ankle_r_foot',
    'lb_servo_l_hip', 'l_hip_l_thigh', 'l_thigh_l_knee', 'l_knee_l_shin', 'l_shin_l_ankle', 'l_ankle_l_foot',
    'torso_r_shoulder', 'r_shoulder_rs_servo', 're_servo_r_elbow', 'torso_l_shoulder', 'l_shoulder_ls_servo', 'le_servo_l_elbow']
MOVING_JOINTS = [5, 6, 7, 9, 10, 11, 13, 14, 15, 17, 18, 19, 20, 21, 24, 26, 27, 30]   # plen_env.py:318-320

register(id="PlenWalkEnv-v1", entry_point="plen_ml_walk_amd.plen_env:PlenWalkEnv", max_episode_steps=500)
# the Gazebo environment's contract (plen_ros_helpers/plen_walk.py:20-24: same id, 26-dim obs, 18 actions) on this simulator
register(id="PlenWalkEnv-v0", entry_point="plen_ml_walk_amd.plen_env:PlenWalkEnvV0", max_episode_steps=500)
try:                                   # also visible to a real gym, when there is one
    import gym as _gym
    _gym.envs.registration.register(id="PlenWalkEnv-v1", entry_point="plen_ml_walk_amd.plen_env:PlenWalkEnv", max_episode_steps=500)
except Exception:                      # noqa: BLE001 -- gym absent or id already registered
    pass


class PlenWalkEnv(Env):
    metadata = {'render.modes': ['human', 'rgb_array'], 'video.frames_per_second': 50}

    def __init__(self, render=False, realtime=False, joint_act=False, device=None, dtype=torch.float64, reward_head=0, quiet=False):
        if render or realtime:
            raise NotImplementedError("the GPU environment has no GUI / wall-clock mode (plen_env.py:275-292 are PyBullet GUI features)")
        self.joint_act = joint_act
        self.running_step = 1. / 60.
        self.timestep = 1. / 240.
        self.sim_stepsize = int(self.running_step / self.timestep)
        self.max_episode_steps = 500
        self.env_ranges = [list(r) for r in ENV_RANGES]
        self.real_ranges = [list(r) for r in REAL_RANGES]
        self.joint_names = list(JOINT_NAMES)
        self.movingJoints = list(MOVING_JOINTS)
        self.action_space = Box(np.ones(18) * -1, np.ones(18), dtype=np.float32)
        lo = [r[0] for r in ENV_RANGES] + [0, -np.inf, -np.pi, -np.pi, -np.pi, -np.inf, 0, 0]
        hi = [r[1] for r in ENV_RANGES] + [0.25, np.inf, np.pi, np.pi, np.pi, np.inf, 1, 1]
        self.observation_space = Box(np.array(lo), np.array(hi), dtype=np.float32)
        self.reward_range = (-np.inf, np.inf)
        self.episode_num = 0
        self.cumulated_episode_reward = 0
        self.episode_timestep = 0
        self.total_timesteps = 0
        # per-episode reward line + moving average over the last 1000 episodes, printed at reset like plen_env.py:52-55, 575-577, 616-636
        # (quiet=True, or PLEN_QUIET=1, switches the print off; the bookkeeping stays)
        self.moving_avg_buffer_size = 1000
        self.moving_avg_buffer = np.zeros(self.moving_avg_buffer_size)
        self.moving_avg_counter = 0
        self.quiet = bool(quiet) or os.environ.get("PLEN_QUIET") == "1"
        # the caller owns resets (plen_td3.py:122-133), the TimeLimit wrapper owns the 500-step limit
        self._vec = PlenVecEnv(1, device=device, dtype=dtype, joint_act=joint_act, auto_reset=False,
                               cfg_overrides={"max_episode_steps": 2 ** 30, "reward_head": int(reward_head)})
        self._dtype = dtype

    def _seed(self, seed=None):        # the reference defines _seed, not seed (plen_env.py:28); the env has no RNG
        return [seed]

    def _publish_reward(self, reward, episode_number):
        """plen_env.py:616-636: this episode's reward and the moving average over the last 1000 episodes (NaN until 1000 exist)."""
        if self.moving_avg_counter >= self.moving_avg_buffer_size:
            self.moving_avg_counter = 0
        self.moving_avg_buffer[self.moving_avg_counter] = self.cumulated_episode_reward
        moving_avg_reward = np.average(self.moving_avg_buffer) if self.episode_num >= self.moving_avg_buffer_size else np.nan
        if not self.quiet:
            print("Episode #{} \tTotal Timesteps: {} \nReward: {} \tMA Reward: {}\n".format(episode_number, self.total_timesteps, reward, moving_avg_reward))
        return moving_avg_reward

    def reset(self):
        obs = self._vec.reset()
        self._publish_reward(self.cumulated_episode_reward, self.episode_num)          # same order as plen_env.py:574-579
        self.episode_num += 1
        self.moving_avg_counter += 1
        self.cumulated_episode_reward = 0
        self.episode_timestep = 0
        return obs[0].to(torch.float64).cpu().numpy()

    def step(self, action):
        a = torch.as_tensor(np.asarray(action, dtype=np.float32).reshape(1, 18))
        obs, reward, done, _ = self._vec.step(a)
        out = torch.cat([obs[0].to(torch.float64), reward.to(torch.float64), done.to(torch.float64)]).cpu().numpy()
        r = np.float64(out[26])
        self.cumulated_episode_reward += r
        self.episode_timestep += 1
        self.total_timesteps += 1
        return out[:26].copy(), r, bool(int(out[27]) & 1), {}

    def agent_to_env(self, env_range, agent_val):
        """plen_env.py:694-714 (host copy, for callers such as trajectory tools)."""
        m = (env_range[1] - env_range[0]) / 2.0
        b = env_range[1] - m
        v = m * agent_val + b
        if v >= env_range[1]:
            v = env_range[1] - 0.001
        elif v <= env_range[0]:
            v = env_range[0] + 0.001
        return v

    def close(self):
        self._vec.close()


class PlenWalkEnvV0(PlenWalkEnv):
    """`PlenWalkEnv-v0` (plen_ros_helpers/plen_walk.py) without ROS/Gazebo: its contact rule (force > weight/3, :346-396), `_is_done`
    (:597-618) and `_compute_reward` (:620-650) evaluated in the kernel on this simulator's state."""

    def __init__(self, device=None, dtype=torch.float64):
        super().__init__(device=device, dtype=dtype, reward_head=1)


def make(id="PlenWalkEnv-v1", **kwargs):   # noqa: A002
    return gym_compat.make(id, **kwargs)
