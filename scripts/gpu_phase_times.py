"""Per-phase latency of one substep (shader-clock stamps from the debug dump), idle chip."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from plen_ml_walk_amd.vec_env import PlenVecEnv
names = ["kinematics", "body_dyn+subtree", "S,M,tau", "cholesky", "v* solve", "collision+J+Y", "A build", "rows setup", "PGS", "apply", "integrate"]
for dtype in (torch.float32, torch.float64):
    for n in (64, 4096):
        env = PlenVecEnv(n, dtype=dtype); env.reset()
        tg = torch.zeros(n, 18)
        for rep in range(3):
            d = env.debug_substeps(tg, nsub=1, dump=True)
        st = d[:, 3800:3812].double().cpu().numpy()
        dt = np.diff(st, axis=1)
        med = np.median(dt, axis=0)
        print(dtype, "n=%d" % n, "total %.0f ticks" % med.sum())
        for k, nme in enumerate(names): print("   %-18s %8.0f" % (nme, med[k]))
        env.close()
