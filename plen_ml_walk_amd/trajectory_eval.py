#!/usr/bin/env python
"""Open-loop gait playback, the GPU form of the reference's plen_bullet/src/trajectory_eval.py: assemble the joint-space walking
trajectory from the gait generator exactly as the reference does (trajectory_eval.py:180-271: leg sign conventions, fixed arm pose,
20 double steps; 20 bend steps first, :282-286), and play it through `joint_act=True` environments (:37).  No GUI, no 20 Hz sleep;
reports reward and distance instead.  The reference also saves the arrays as trajectories/<joint>_traj.npy: `--save DIR` does that."""
import argparse
import json
import os

import numpy as np

from .plen_env import JOINT_NAMES
from .trajectory_generator import TrajectoryGenerator


def assemble_joint_trajectories(traj=None, repeats=20):
    """-> (walk [repeats*(n_rfwd+n_lfwd), 18], bend [18]) in the reference's joint order; trajectory_eval.py:180-271."""
    if traj is None:
        traj = TrajectoryGenerator()
        traj.main()
    sign = np.array([-1, -1, -1, -1, 1, 1, 1, 1, 1, 1, -1, 1], dtype=np.float64)      # :202-213 / :222-233
    arms = np.array([np.pi / 5, np.pi / 8, 0, -np.pi / 5, np.pi / 8, 0])              # :214-219
    rows = []
    for _ in range(repeats):
        for src in (traj.foot_walk_rfwd, traj.foot_walk_lfwd):
            for i in range(np.size(src, 0)):
                rows.append(np.concatenate([sign * np.asarray(src[i][:12], dtype=np.float64), arms]))
    walk = np.array(rows)
    bend = np.append(np.asarray(traj.bend[:][0], dtype=np.float64), np.zeros(6))      # :252-254
    bend[13] = 0.5; bend[16] = 0.5                                                     # :256-257
    bend[:4] = -bend[:4]                                                               # :259-260
    bend[10] = -bend[10]                                                               # :262
    return walk, bend


def play(num_envs=1, dtype=None, bend_steps=20):
    """Bend for `bend_steps` control steps, then walk open loop; returns per-step reward [T, N], done flags and final x."""
    import torch
    from .vec_env import PlenVecEnv
    walk, bend = assemble_joint_trajectories()
    env = PlenVecEnv(num_envs, joint_act=True, auto_reset=False, dtype=dtype or torch.float32, cfg_overrides={"max_episode_steps": 2 ** 30})
    env.reset()
    acts = np.concatenate([np.tile(bend, (bend_steps, 1)), walk], 0)
    a = torch.tensor(acts, dtype=torch.float32, device=env.device)
    rew, done = [], []
    for t in range(a.shape[0]):
        _, r, d, _ = env.step(a[t][None].expand(num_envs, 18).contiguous())
        rew.append(r.clone()); done.append(d.clone())
    st = env.get_state()
    env.close()
    return torch.stack(rew).cpu().numpy(), torch.stack(done).cpu().numpy(), st[:, 0].cpu().numpy()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--save", default=None, help="directory for <joint>_traj.npy / bend_traj.npy, as the reference writes them")
    ap.add_argument("--envs", type=int, default=1)
    a = ap.parse_args(argv)
    walk, bend = assemble_joint_trajectories()
    if a.save:
        os.makedirs(a.save, exist_ok=True)
        for i, name in enumerate(JOINT_NAMES):
            np.save(os.path.join(a.save, name + "_traj"), walk[:, i])
        np.save(os.path.join(a.save, "bend_traj"), bend)
    rew, done, x = play(a.envs)
    fell = np.argmax(done[:, 0] & 1) if (done[:, 0] & 1).any() else -1
    print(json.dumps({"steps": int(rew.shape[0]), "return": float(rew[:, 0].sum()), "first_fall_step": int(fell), "final_x": float(x[0])}))


if __name__ == "__main__":
    main()
