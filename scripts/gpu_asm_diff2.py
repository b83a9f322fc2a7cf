import os, sys, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def run(tag):
    import torch
    from plen_ml_walk_amd.vec_env import PlenVecEnv
    env = PlenVecEnv(64, dtype=torch.float32, cfg_overrides={"num_iterations": 33, "reset_substeps": 0}); env.reset()
    d = env.debug_substeps(torch.full((64, 18), 0.2), nsub=1, dump=True).cpu().numpy()
    np.save(os.path.join(ROOT, "gpurun_out", "asmdiff2_%s.npy" % tag), d[0, 3900:3900 + 120])
if __name__ == "__main__":
    if len(sys.argv) > 1: run(sys.argv[1]); sys.exit(0)
    env = dict(os.environ); subprocess.check_call([sys.executable, __file__, "asm"], env=env)
    env["PLENVEC_NO_ASM"] = "1"; subprocess.check_call([sys.executable, __file__, "noasm"], env=env)
    a = np.load(os.path.join(ROOT, "gpurun_out", "asmdiff2_asm.npy")).reshape(5, 3, 8); b = np.load(os.path.join(ROOT, "gpurun_out", "asmdiff2_noasm.npy")).reshape(5, 3, 8)
    np.set_printoptions(precision=9, linewidth=220)
    for it in range(5):
        for li, ln in enumerate((33, 47, 39)):
            print("it", 28 + it, "lane", ln, "asm  ", a[it, li, :8]); print("              noasm", b[it, li, :8], "diff", (a[it, li, :8] != b[it, li, :8]).astype(int))
