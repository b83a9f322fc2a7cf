"""Profiling build (-DPGS_STAMPS): finer shader-clock stamps inside phase E (collision + port Jacobians + Y) of one substep, idle chip and loaded chip.
usage: python scripts/gpu_phase_e_stamps.py [f32|f64]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from plen_ml_walk_amd.build import build_variant
os.environ["PLENVEC_LIB"] = os.environ.get("STAMPS_LIB") or build_variant("stamps", ["-DPGS_STAMPS"])
from plen_ml_walk_amd.vec_env import PlenVecEnv
names = ["foot manifolds", "foot Jacobians", "box near test", "J rows -> regs", "port velocities", "back subst + store"]
dtype = torch.float64 if len(sys.argv) > 1 and sys.argv[1] == "f64" else torch.float32
for n in (64, 4096):
    env = PlenVecEnv(n, dtype=dtype); env.reset()
    d = env.debug_substeps(torch.zeros(n, 18), nsub=1, dump=True)
    st = d[:, 3850:3857].double().cpu().numpy()
    dt = np.median(np.diff(st, axis=1), axis=0)
    print(dtype, "n=%d: phase E total %.0f" % (n, dt.sum()))
    for k, nme in enumerate(names): print("   %-20s %7.0f" % (nme, dt[k]))
    env.close()
