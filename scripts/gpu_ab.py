"""A/B timing of kernel builds inside ONE gpurun call (boxes differ by a few percent between calls).
usage: python scripts/gpu_ab.py libA.so libB.so ...   (paths relative to plen_ml_walk_amd/csrc/variants/)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from plen_ml_walk_amd.vec_env import PlenVecEnv
def run(n, steps=60):
    env = PlenVecEnv(n); env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    acts = torch.rand(steps + 10, n, 18, device="cuda", generator=g) * 2 - 1
    for t in range(10): env.step(acts[t])
    torch.cuda.synchronize(); env.timing_begin()
    for t in range(steps): env.step(acts[10 + t])
    ms, nl = env.timing_end(); env.close(); return ms / nl
print(" ".join("%%d:%%.4f" %% (n, run(n)) for n in (1024, 2048, 4096, 8192, 16384)))
''' % ROOT
if __name__ == "__main__":
    libs = sys.argv[1:]
    for rnd in range(2):
        for lib in libs:
            name, _, extra = lib.partition("@")          # "file.so@VAR=1,VAR2=x" sets environment variables for that run
            env = dict(os.environ, PLENVEC_LIB=os.path.join(ROOT, "plen_ml_walk_amd", "csrc", "variants", name))
            env.update(kv.split("=", 1) for kv in extra.split(",") if kv)
            out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=120)
            print("%-28s %s %s" % (lib, out.stdout.strip(), out.stderr.strip()[-200:] if out.returncode else ""), flush=True)
