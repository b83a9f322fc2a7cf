"""Profiling build (-DPGS_STAMPS) only: shader-clock stamps inside PGS iterations 3 (odd) and 4 (even).  usage: python scripts/gpu_pgs_stamps.py [f32|f64]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
sys.path.insert(0, ROOT)
from plen_ml_walk_amd.build import build_variant
os.environ["PLENVEC_LIB"] = os.environ.get("STAMPS_LIB") or build_variant("stamps", ["-DPGS_STAMPS"])      # STAMPS_LIB: a -DPGS_STAMPS build of another source (A/B)
from plen_ml_walk_amd.vec_env import PlenVecEnv
names = ["motors(+limits)", "normals", "tors bounds", "spin rows", "roll rows", "cone pairs", "wave max"]
dtype = torch.float64 if len(sys.argv) > 1 and sys.argv[1] == "f64" else torch.float32
for n in (64, 4096):
    env = PlenVecEnv(n, dtype=dtype); env.reset()
    tg = torch.zeros(n, 18)
    for _ in range(40): env.debug_substeps(tg, nsub=1, dump=False)     # settle onto the ground
    d = env.debug_substeps(tg, nsub=1, dump=True)
    aux = env.get_aux().cpu().numpy()
    for itn in (3, 4):
        st = d[:, 3820 + 10 * (itn - 3): 3828 + 10 * (itn - 3)].double().cpu().numpy()
        dt = np.median(np.diff(st, axis=1), axis=0)
        print("n=%d iteration %d: total %.0f  aux %s" % (n, itn, dt.sum(), aux[0]))
        for k, nme in enumerate(names): print("   %-16s %7.0f" % (nme, dt[k]))
    env.close()
