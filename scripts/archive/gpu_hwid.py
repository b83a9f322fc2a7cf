"""Profiling build (-DPGS_STAMPS -DPGS_HWID, variants/hwid.so): which (XCC, SE, CU, SIMD, slot) each block of a 4096-block launch ran on."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from plen_ml_walk_amd.build import build_variant
os.environ["PLENVEC_LIB"] = build_variant("hwid", ["-DPGS_STAMPS", "-DPGS_HWID"])
sys.path.insert(0, ROOT)
import numpy as np, torch
from plen_ml_walk_amd.vec_env import PlenVecEnv
n = 4096
env = PlenVecEnv(n, auto_reset=False); env.reset()
acts = torch.zeros(n, 18, device="cuda")
maps = []
for t in range(3):
    env.step(acts)
    a = env.get_aux().cpu().numpy()[:, 7]
    maps.append(a.copy())
a = maps[-1]
wave, simd, cu, sh, se, xcc = a & 0xf, (a >> 4) & 3, (a >> 8) & 0xf, (a >> 12) & 1, (a >> 13) & 7, (a >> 16) & 0xf
print("stable across launches:", [bool((maps[i] == maps[-1]).all()) for i in range(2)], " changed:", [int((maps[i] != maps[-1]).sum()) for i in range(2)])
print("first 40 blocks (xcc, se, sh, cu, simd, wave):")
for b in range(40): print(b, xcc[b], se[b], sh[b], cu[b], simd[b], wave[b])
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
simdkey = key * 4 + simd
u, cnt = np.unique(simdkey, return_counts=True)
print("distinct SIMDs", len(u), "waves per SIMD min/max", cnt.min(), cnt.max())
# which blocks share a SIMD with block b?
for b in (0, 1, 2, 100):
    print("SIMD mates of block", b, ":", np.nonzero(simdkey == simdkey[b])[0].tolist())
np.save(os.path.join(ROOT, "gpurun_out", "hwid_map.npy"), np.stack(maps))
env.close()
