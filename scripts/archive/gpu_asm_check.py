"""asm fast path vs compiler path (PLENVEC_NO_ASM=1): same arithmetic in the same order -> results
should agree bit for bit on every env; a hazard in the hand-written row would show up here."""
import os, sys, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
def run(tag):
    import torch
    from plen_ml_walk_amd.vec_env import PlenVecEnv
    n, T = 512, 30
    g = torch.Generator(device="cpu").manual_seed(1)
    acts = (torch.rand(T, n, 18, generator=g) * 2 - 1).float().cuda()
    env = PlenVecEnv(n, dtype=torch.float32); env.reset()
    obs = []
    for t in range(T):
        o, r, d, info = env.step(acts[t]); obs.append(torch.cat([o, r[:, None], info["flags"].float()[:, None]], 1).cpu().numpy().copy())
    np.save(os.path.join(ROOT, "gpurun_out", "asmcheck_%s.npy" % tag), np.array(obs))
if __name__ == "__main__":
    if len(sys.argv) > 1: run(sys.argv[1]); sys.exit(0)
    env = dict(os.environ); subprocess.check_call([sys.executable, __file__, "asm"], env=env)
    env["PLENVEC_NO_ASM"] = "1"; subprocess.check_call([sys.executable, __file__, "noasm"], env=env)
    a = np.load(os.path.join(ROOT, "gpurun_out", "asmcheck_asm.npy")); b = np.load(os.path.join(ROOT, "gpurun_out", "asmcheck_noasm.npy"))
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    print("asm vs compiler path: identical fraction", same.mean(), "max abs diff", np.nanmax(np.abs(a - b)), "first mismatch step", (np.argwhere(~same)[:1]))
    for t in range(0, a.shape[0], 5): print(" step", t, "identical", same[t].all(1).mean())
