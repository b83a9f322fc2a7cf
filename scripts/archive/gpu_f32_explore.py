"""Exploration behind the f32 variants of the recorded-policy / trajectory tests: per-step error of the f32 kernel vs the f64 oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle.oracle import OracleEnv
from plen_ml_walk_amd.vec_env import PlenVecEnv
a = np.load(os.path.join(ROOT, "tests", "golden", "policy_cmd_sequence.npz"))["actions"]
for dtype in (torch.float32, torch.float64):
    for rolling in (0.0, None):
        ov = {} if rolling is None else {"rolling_friction": rolling}
        env = PlenVecEnv(1, dtype=dtype, cfg_overrides=ov); env.reset()
        o = OracleEnv()
        if rolling is not None: o.set_friction(rolling=rolling)
        o.reset(); errs = []
        for t in range(30):
            nobs, rew, done, _ = env.step(torch.tensor(a[t:t + 1]).cuda())
            ob, r, d, _ = o.step(a[t].astype(np.float64))
            errs.append(np.abs(ob - nobs[0].cpu().numpy().astype(np.float64)).max())
        print("policy cmds", dtype, rolling, " ".join("%.0e" % e for e in errs))
        env.close()
from plen_ml_walk_amd.trajectory_eval import assemble_joint_trajectories
from plen_ml_walk_amd.trajectory_generator import TrajectoryGenerator
walk, bend = assemble_joint_trajectories()
seqs = {"ref trajectory": np.concatenate([np.tile(bend, (20, 1)), walk[:100]], 0),
        "generated gait": TrajectoryGenerator(num_DoubleSupport=20, num_SingleSupport=20, height=20.0, stride=20.0).walk_cycle_actions(cycles=2)[:100]}
for name, acts in seqs.items():
    for dtype in (torch.float32, torch.float64):
        env = PlenVecEnv(1, dtype=dtype, joint_act=True); env.reset()
        o = OracleEnv(joint_act=True); o.reset(); errs = []
        for t in range(len(acts)):
            nobs, rew, done, _ = env.step(torch.tensor(acts[t:t + 1], dtype=torch.float32).cuda())
            ob, r, d, _ = o.step(acts[t].astype(np.float32).astype(np.float64))
            errs.append(np.abs(ob - nobs[0].cpu().numpy().astype(np.float64)).max())
        print(name, dtype, " ".join("%.0e" % e for e in errs[::4]))
        env.close()
