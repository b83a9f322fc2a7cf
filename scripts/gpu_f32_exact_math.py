"""Is the f32 kernel's distance from the f64 oracle in the reference configuration caused by its 1-ulp hardware rcp / rsq / sin / cos
?  Builds a variant with correctly rounded division / sqrt and libm sine / cosine (-DPLENVEC_EXACT_MATH) and measures the
first-step and 4-step error quantiles of all 4096 envs for both builds, reference configuration and rolling friction off."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
    env = PlenVecEnv(n, cfg_overrides={} if rolling is None else dict(rolling_friction=rolling)); env.reset()
    O = []
    for t in range(T):
        o, r, d, _ = env.step(acts[t]); O.append(o.cpu().numpy().astype(np.float64))
    env.close()
    oo, rr, ff = oracle.batch_rollout(acts.cpu().numpy(), None, None, -1.0 if rolling is None else rolling)
    err = np.abs(np.array(O) - oo).max(2)
    out[name] = [dict(median=float(np.median(e)), p90=float(np.quantile(e, .9)), within_1e4=float((e <= 1e-4).mean())) for e in err]
print(json.dumps(out))
''' % ROOT
if __name__ == "__main__":
    from plen_ml_walk_amd.build import build_variant
    res = {}
    for tag, lib in (("hardware rcp/rsq/sin/cos (product build)", None), ("exact division/sqrt, libm sin/cos", build_variant("exact_math", ["-DPLENVEC_EXACT_MATH"]))):
        env = dict(os.environ)
        if lib:
            env["PLENVEC_LIB"] = lib
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
        res[tag] = json.loads(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 else out.stderr[-400:]
        print(tag)
        if out.returncode == 0:
            for cfg, rows in res[tag].items():
                print("   %-12s" % cfg, " | ".join("t=%d med %.1e p90 %.1e <=1e-4 %.3f" % (t, r["median"], r["p90"], r["within_1e4"]) for t, r in enumerate(rows)))
        else:
            print(res[tag])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r02_f32_exact_math.json"), "w"), indent=1)
