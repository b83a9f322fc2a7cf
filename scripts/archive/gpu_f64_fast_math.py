"""What do the f64 kernel's Newton-refined reciprocals / reciprocal square roots (v_rcp_f64 / v_rsq_f64 + two steps; friction-cone square roots skipped
inside the circle) change?  Builds a variant with the compiler's correctly rounded division / sqrt (-DPLENVEC_EXACT_MATH) and measures, for both builds
and all 4096 envs, the error against the f64 oracle after 1..4 steps (reference configuration and rolling friction off), and the two builds against
each other."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CHILD = r'''
import sys, json, numpy as np, torch
sys.path.insert(0, %r)
from oracle import oracle
from plen_ml_walk_amd.vec_env import PlenVecEnv
n, T = 4096, 4
g = torch.Generator(device="cuda").manual_seed(0)
acts = torch.rand(T, n, 18, generator=g, device="cuda") * 2 - 1
out = {}
for name, rolling in (("reference", None), ("rolling_off", 0.0)):
    env = PlenVecEnv(n, dtype=torch.float64, cfg_overrides={} if rolling is None else dict(rolling_friction=rolling)); env.reset()
    O = []
    for t in range(T):
        o, r, d, _ = env.step(acts[t]); O.append(o.cpu().numpy().astype(np.float64))
    env.close()
    oo, rr, ff = oracle.batch_rollout(acts.cpu().numpy(), None, None, -1.0 if rolling is None else rolling)
    err = np.abs(np.array(O) - oo).max(2)
    out[name] = [dict(median=float(np.median(e)), p90=float(np.quantile(e, .9)), within_1e4=float((e <= 1e-4).mean())) for e in err]
    np.save("/tmp/f64_obs_" + name + "_" + (sys.argv[1] if len(sys.argv) > 1 else "x") + ".npy", np.array(O))
print(json.dumps(out))
''' % ROOT
if __name__ == "__main__":
    from plen_ml_walk_amd.build import build_variant
    res = {}
    for tag, lib in (("v_rcp_f64 / v_rsq_f64 + Newton (product build)", None), ("exact division / sqrt", build_variant("exact_math", ["-DPLENVEC_EXACT_MATH"]))):
        env = dict(os.environ)
        if lib:
            env["PLENVEC_LIB"] = lib
        out = subprocess.run([sys.executable, "-c", CHILD, "exact" if lib else "fast"], env=env, capture_output=True, text=True, timeout=600)
        res[tag] = json.loads(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 else out.stderr[-400:]
        print(tag)
        if out.returncode == 0:
            for cfg, rows in res[tag].items():
                print("   %-12s" % cfg, " | ".join("t=%d med %.1e p90 %.1e <=1e-4 %.3f" % (t, r["median"], r["p90"], r["within_1e4"]) for t, r in enumerate(rows)))
        else:
            print(res[tag])
    import numpy as np
    res["fast build vs exact build (max abs difference of the observations over all envs, per step)"] = {}
    for name in ("reference", "rolling_off"):
        a, b = np.load("/tmp/f64_obs_%s_fast.npy" % name), np.load("/tmp/f64_obs_%s_exact.npy" % name)
        d = np.abs(a - b).max(2)
        res["fast build vs exact build (max abs difference of the observations over all envs, per step)"][name] = [dict(median=float(np.median(e)), p99=float(np.quantile(e, .99)), max=float(e.max())) for e in d]
        print("fast vs exact %-12s" % name, " | ".join("t=%d med %.1e p99 %.1e max %.1e" % (t, np.median(e), np.quantile(e, .99), e.max()) for t, e in enumerate(d)))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r02_f64_fast_math.json"), "w"), indent=1)
