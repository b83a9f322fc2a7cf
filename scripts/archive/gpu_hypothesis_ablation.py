"""The hypothesis ablation on the KERNEL (f64, through the C ABI's PlenCfg), for the hypotheses that are cfg fields: per variant the pin
residuals R_0..R_4 of the kernel itself (tests/pybullet_pin.py) and the shipped actor's episode statistics at sigma = 0.01 over 4096 episodes
(the oracle's ablation, scripts/pin/hypothesis_ablation.py, holds 128: these are the tight error bars).  The structural variants (manifold
family, row order, warm starting, inertia source) exist only in the oracle.  Writes gpurun_out/r03_hypothesis_ablation_gpu.json."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import pybullet_pin as P
from plen_ml_walk_amd.vec_env import PlenVecEnv
from plen_ml_walk_amd.walk_eval import load_policy, evaluate

V = [("baseline", {})]
for k, vals in (("motor_kp", (0.05, 0.09, 0.11, 0.2)), ("motor_kd", (0.5, 0.9, 1.1, 2.0)), ("motor_max_force", (0.1, 0.14, 0.16, 0.2)),
                ("num_iterations", (10, 49, 51, 100)), ("erp2", (0.04, 0.06, 0.1, 0.2)), ("lateral_friction", (0.4, 0.5, 0.8, 1.0)),
                ("rolling_friction", (0.0, 0.008, 0.04, 0.1)), ("spinning_friction", (0.0, 0.04)), ("restitution", (0.0,)),
                ("restitution_velocity_threshold", (0.0,)), ("linear_slop", (0.0, 1e-4)), ("body_contacts", (0,))):
    for v in vals:
        V.append(("%s = %g" % (k, v), {k: v}))


def pin(cfg):
    env = PlenVecEnv(2, dtype=torch.float64, auto_reset=False, cfg_overrides=cfg)

    def step(a):
        o, _, d, _ = env.step(torch.from_numpy(np.tile(np.asarray(a, dtype=np.float32), (2, 1))).cuda())
        return o[0].cpu().numpy(), bool(d[0].item() & 1)
    R, _ = P.residuals(lambda: env.reset()[0].cpu().numpy(), step, 4)
    env.close()
    return [None if np.isnan(x) else round(float(x), 5) for x in R]


pol = load_policy(os.path.join(ROOT, "tests", "golden", "policy_3229999.npz"))
out = []
t0 = time.time()
for name, cfg in V:
    r = evaluate(pol, 4096, 1, torch.float64, action_noise=0.01, seed=0, cfg_overrides=cfg)
    ret, ln = np.array(r["returns"]), np.array(r["lengths"])
    n = len(ret)
    row = dict(name=name, cfg=cfg, pin_R=pin(cfg), episodes=n, mean_length=float(ln.mean()), length_sem=float(ln.std() / np.sqrt(n)),
               mean_return=float(ret.mean()), return_sem=float(ret.std() / np.sqrt(n)), full_length_fraction=float((ln >= 500).mean()))
    out.append(row)
    print("%-42s R0 %.4f R1 %.3f | len %6.1f +- %.1f  ret %7.1f +- %.1f  full %.3f" % (name, row["pin_R"][0], row["pin_R"][1], row["mean_length"], row["length_sem"],
          row["mean_return"], row["return_sem"], row["full_length_fraction"]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(dict(what=__doc__.split("\n\n")[0], seconds=round(time.time() - t0, 1), variants=out), open(os.path.join(ROOT, "gpurun_out", "r03_hypothesis_ablation_gpu.json"), "w"), indent=1)
