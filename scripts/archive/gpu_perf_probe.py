"""Where does the time go: vary solver iterations / env count / dtype."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from plen_ml_walk_amd.vec_env import PlenVecEnv

def run(n, dtype, steps=40, **ov):
    env = PlenVecEnv(n, dtype=dtype, cfg_overrides=ov)
    env.reset()
    acts = (torch.rand(steps + 5, n, 18, device="cuda") * 2 - 1)
    for t in range(5): env.step(acts[t])
    torch.cuda.synchronize()
    env.timing_begin()
    for t in range(steps): env.step(acts[5 + t])
    ms, nl = env.timing_end()
    env.close()
    return ms / nl

if __name__ == "__main__":
    for dtype in (torch.float32, torch.float64):
        for it in (50, 25, 1):
            t = run(4096, dtype, num_iterations=it)
            print("%s N=4096 iterations=%2d: %.3f ms/step -> %.2f M env-steps/s" % (dtype, it, t, 4096 / t / 1e3))
    for n in (1024, 2048, 4096, 8192, 16384, 65536):
        t = run(n, torch.float32)
        print("f32 N=%6d: %.3f ms/step -> %.2f M env-steps/s" % (n, t, n / t / 1e3))
