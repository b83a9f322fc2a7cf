import os, sys, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NITS = (28, 29, 30, 31, 32)
def run(tag):
    import torch
    from plen_ml_walk_amd.vec_env import PlenVecEnv
    out = {}
    for nit in NITS:
        env = PlenVecEnv(64, dtype=torch.float32, cfg_overrides={"num_iterations": nit, "reset_substeps": 0})
        env.reset()
        tg = torch.full((64, 18), 0.2)
        d = env.debug_substeps(tg, nsub=1, dump=True).cpu().numpy()
        out["lam%d" % nit] = d[:, 3616:3664]
        out["st%d" % nit] = env.get_state().cpu().numpy()
        out["it%d" % nit] = env.get_aux().cpu().numpy()[:, 6]
        env.close()
    np.savez(os.path.join(ROOT, "gpurun_out", "asmdiff_%s.npz" % tag), **out)
if __name__ == "__main__":
    if len(sys.argv) > 1: run(sys.argv[1]); sys.exit(0)
    env = dict(os.environ)
    subprocess.check_call([sys.executable, __file__, "asm"], env=env); subprocess.check_call([sys.executable, __file__, "asm2"], env=env)
    env["PLENVEC_NO_ASM"] = "1"; subprocess.check_call([sys.executable, __file__, "noasm"], env=env); subprocess.check_call([sys.executable, __file__, "noasm2"], env=env)
    L = {t: np.load(os.path.join(ROOT, "gpurun_out", "asmdiff_%s.npz" % t)) for t in ("asm", "asm2", "noasm", "noasm2")}
    for nit in NITS:
        k = "st%d" % nit
        print("nit", nit, "asm run-to-run", np.abs(L["asm"][k] - L["asm2"][k]).max(), "noasm run-to-run", np.abs(L["noasm"][k] - L["noasm2"][k]).max(),
              "asm vs noasm", np.abs(L["asm"][k] - L["noasm"][k]).max(), "envs identical within asm run", (L["asm"][k] == L["asm"][k][0:1]).all(), "iters asm/noasm", L["asm"]["it%d" % nit][0], L["noasm"]["it%d" % nit][0])
        dl = np.abs(L["asm"]["lam%d" % nit][0] - L["noasm"]["lam%d" % nit][0]); print("    lam diff ports", np.nonzero(dl)[0].tolist(), dl[np.nonzero(dl)[0]][:8], "lam asm", L["asm"]["lam%d" % nit][0][np.nonzero(dl)[0]][:8])
