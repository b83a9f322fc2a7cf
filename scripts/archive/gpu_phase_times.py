"""Per-phase latency of one substep (shader-clock stamps from the debug dump), idle chip."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from plen_ml_walk_amd.vec_env import PlenVecEnv
names = ["kinematics", "body_dyn+subtree", "S,M,tau", "cholesky", "v* solve", "collision+J+Y", "A build", "rows setup", "PGS+apply", "integrate"]
for dtype in (torch.float32, torch.float64):
    for n, air in ((64, False), (64, True), (4096, False)):
        env = PlenVecEnv(n, dtype=dtype); env.reset()
        if air:
            s = env.get_state(); s[:, 2] = 1.0; env.set_state(s)
        tg = torch.zeros(n, 18)
        d = env.debug_substeps(tg, nsub=1, dump=True)
        st = d[:, 3800:3811].double().cpu().numpy()
        dt = np.diff(st, axis=1)
        med = np.median(dt, axis=0)
        aux = env.get_aux().cpu().numpy()
        print(dtype, "n=%d airborne=%s" % (n, air), "total %.0f ticks" % med.sum(), "iters", aux[0, 6], "contacts", aux[0, 4], aux[0, 5])
        for k, nme in enumerate(names): print("   %-18s %8.0f" % (nme, med[k]))
        env.close()
