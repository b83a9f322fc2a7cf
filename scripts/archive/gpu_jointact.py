import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import OracleEnv
from plen_ml_walk_amd.vec_env import PlenVecEnv
from plen_ml_walk_amd.trajectory_generator import TrajectoryGenerator
acts = TrajectoryGenerator(num_DoubleSupport=20, num_SingleSupport=20, height=20.0, stride=20.0).walk_cycle_actions(cycles=2)
print("trajectory", acts.shape)
for dtype in (torch.float64, torch.float32):
    env = PlenVecEnv(2, dtype=dtype, joint_act=True); env.reset()
    o = OracleEnv(joint_act=True); o.reset()
    errs = []; rerr = []; alive = True
    for t in range(acts.shape[0]):
        nobs, rew, done, _ = env.step(torch.tensor(np.stack([acts[t], acts[t]]), dtype=torch.float32).cuda())
        ob, r, d, _ = o.step(acts[t].astype(np.float32).astype(np.float64))
        errs.append(np.abs(nobs[0].cpu().numpy().astype(np.float64) - ob).max()); rerr.append(abs(float(rew[0]) - r))
        if d or int(done[0]): print("episode ended at", t, d, int(done[0])); break
    print(dtype, "steps", len(errs), "obs err", ["%.1e" % e for e in errs[::20]], "max", max(errs), "reward err max", max(rerr), "final z", ob[18], "x progress n/a")
    env.close()
