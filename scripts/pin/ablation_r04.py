#!/usr/bin/env python3
"""Round-4 hypothesis ablation against the chaos-robust, reference-held objectives (scripts/pin/closed_loop_stats.py) beside the spawn pins
R_0 / R_1 (tests/pybullet_pin.py).  Runs on the CPU ORACLE.  usage: ablation_r04.py [episodes] [name-filter] -> profiles/r04_ablation.json"""
import json, os, sys, time
import numpy as np
from multiprocessing import Pool
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pin_eval import make_env, residuals, ROOT
from closed_loop_stats import evaluate, _episodes

VARIANTS = []
def V(name, **hyp): VARIANTS.append((name, hyp))
V("baseline (round-3 defaults)")
V("motor kd 1.1", kd=1.1)
V("joint damping 0.001", joint_damping=1e-3)
V("motor max force 0.16", max_force=0.16)
V("velocity clamp 30", rhs_clamp=30.0)
V("solver iterations 100", iters=100)
V("restitution 0", rest=0.0)
V("rolling friction 0.04", roll=0.04)
V("rolling friction 0.02", roll=0.02)
V("rolling rows off", roll=0.0)
V("torsional rows on the first point only", tors_pts=1)
V("torsional rows on the first two points", tors_pts=2)
V("pyramid friction", pyramid=1)
V("lateral friction 0.5", mu=0.5)
V("persistent manifold (anchors)", manifold=1)
V("persistent manifold, no anchors", manifold=1, friction_erp=0.0)
V("persistent manifold, no anchors, torsional first point", manifold=1, friction_erp=0.0, tors_pts=1)
V("persistent manifold, no anchors, first point (10, 6) mm", manifold=1, friction_erp=0.0, man_p1=1, man_p1x=0.010, man_p1y=0.006)
P = dict(manifold=1, man_add_all=1, friction_erp=0.0, man_drift=1e6)
V("P: all in-range hull vertices per pass", **P)
V("P with the drift test", **dict(P, man_drift=1.0))
V("persistent + warm starting 0.1", manifold=1, friction_erp=0.0, warm=0.1)
V("erp2 0.1", erp2=0.1)
V("erp2 0.2 (Bullet default)", erp2=0.2)


def run(item):
    name, hyp = item
    R, _ = residuals(make_env(hyp=hyp), K=4)
    return dict(name=name, hyp=hyp, R0=float(R[0]), R1=float(R[1]))


if __name__ == "__main__":
    episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    todo = [v for v in VARIANTS if flt in v[0]]
    out = []
    with Pool(8) as p:
        pins = p.map(run, todo)
        for pin in pins:
            t0 = time.time()
            pin["sigma_0.1"] = evaluate(dict(hyp=pin["hyp"]), sigma=0.1, episodes=episodes, pool=p)
            pin["sigma_1e-4"] = evaluate(dict(hyp=pin["hyp"]), sigma=1e-4, episodes=episodes // 2, pool=p)
            a, b = pin["sigma_0.1"], pin["sigma_1e-4"]
            print("%-58s R0 %.4f R1 %.3f | s=.1: early %.2f full %.2f med %+5.0f q95 %+5.0f W1 %5.1f | s=1e-4: len %5.1f full %.2f | %.0fs" % (
                pin["name"][:58], pin["R0"], pin["R1"], a["early_falls_lt50"], a["full_length"], a["ret_q"][2], a["ret_q"][4], a["w1_to_reference_last1000"],
                b["mean_length"], b["full_length"], time.time() - t0), flush=True)
            out.append(pin)
    if not flt:
        json.dump(dict(what=__doc__, reference=dict(last1000_mean=50.4, quantiles_5_25_50_75_95=[-113, -15, 55, 119, 200], max_over_24832_episodes=328,
                                                     deterministic_episode_length=500), variants=out),
                  open(os.path.join(ROOT, "profiles", "r04_ablation.json"), "w"), indent=1)
