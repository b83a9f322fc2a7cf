"""Development: the mass matrix of the matrix-core build against the vector build, entry by entry (debug dump of one substep), f32 and f64.
usage: python scripts/gpu_twin_check.py   (builds csrc/variants/tw_mass0.so = -DPLENVEC_COUNT_SPECIALISED=0 -DPLENVEC_MFMA_MASS=0 when missing; build it here, the GPU box takes minutes)"""
import os, subprocess, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plen_ml_walk_amd.build import build_variant
build_variant("tw_mass0", ["-DPLENVEC_COUNT_SPECIALISED=0", "-DPLENVEC_MFMA_MASS=0"])
code = ("import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from plen_ml_walk_amd.vec_env import PlenVecEnv\n"
        "env = PlenVecEnv(64, dtype=getattr(torch, sys.argv[2])); env.reset()\n"
        "g = torch.Generator().manual_seed(5); tg = (torch.rand(64, 18, generator=g) * 0.6 - 0.3)\n"
        "d = env.debug_substeps(tg.to(getattr(torch, sys.argv[2])), nsub=1, dump=True)\n"
        "np.save(sys.argv[1], d[:, :640].double().cpu().numpy())\n" % ROOT)
for dt in ("float32", "float64"):
    res = {}
    for tag in ("-", "tw_mass0.so"):
        env = dict(os.environ)
        if tag != "-": env["PLENVEC_LIB"] = os.path.join(ROOT, "plen_ml_walk_amd/csrc/variants", tag)
        p = "/tmp/twm_%s_%s.npy" % (dt, tag.replace(".so", ""))
        subprocess.run([sys.executable, "-c", code, p, dt], check=True, env=env)
        res[tag] = np.load(p)
    a, b = res["-"][:, :576], res["tw_mass0.so"][:, :576]
    rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
    nz = b != 0
    print(dt, "M: max abs diff / max |M| = %.3e" % (np.abs(a - b).max() / np.abs(b).max()))
    print(dt, "M: entries compared", nz.sum(), "bitwise equal", (a == b).sum(), "of", a.size, "max rel diff", rel[nz].max() if nz.any() else 0, "zeros agree", bool(((a == 0) == (b == 0)).all()),
          "| tau max abs diff", np.abs(res["-"][:, 576:600] - res["tw_mass0.so"][:, 576:600]).max())
