#!/usr/bin/env python3
"""Chaos-averaged closed-loop pins of the physics restatement (round 4; runs on the CPU ORACLE = test infrastructure).

In the reference configuration one control step amplifies a 1e-9 state perturbation by 1e6-1e7 in ~20 % of the steps (the rolling-friction
bounds are rewritten from the normal impulse inside every solver iteration with a 0.08 m coefficient, larger than the foot: DESIGN.md section 5,
scripts/pin/expanding_mode.py), so PyBullet's recorded episode is ONE sample path of a chaotic system and cannot be followed step by step
(tests/pin_track.py: even the simulator observing itself loses the episode).  What the reference holds that survives chaos is statistics:

  (a) results/plen_walk_gazebo_.npy -- the return of every training episode (24 832); the last 1000 were collected by the policies around
      the shipped checkpoints under the driver's exploration noise N(0, 0.1) (plen_td3.py:101-104): mean +50, median +55, quartiles -15 / +119,
      5 % / 95 % -113 / +200, maximum over the whole run 328.
  (b) trajectories/*_cmd.npy -- the deterministic 500-step episode of actor 3229999: it did NOT fall, and its action statistics
      (per-channel mean / std / saturation, gait period 19 steps) describe the steady gait.

This script evaluates a simulator variant against both: the shipped actor 3229999 under N(0, sigma) action noise, N episodes.
"""
import os, sys
import numpy as np
from multiprocessing import Pool
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pin_eval import make_env, ACTS, ROOT
from pin_eval import pre

def _episodes(args):
    kw, sigma, seed, n = args
    rng = np.random.default_rng(seed)
    e = make_env(**kw)
    out = []
    for ep in range(n):
        obs = e.reset(); ret = 0.0; acts = []
        for t in range(500):
            a = np.tanh(pre(obs))
            if sigma:
                a = np.clip(a + sigma * rng.standard_normal(18), -1, 1)
            a = a.astype(np.float32); acts.append(a)
            obs, r, done, _ = e.step(a.astype(np.float64)); ret += r
            if done:
                break
        out.append((t + 1, ret, np.array(acts) if t + 1 >= 500 else None))
    return out


from pybullet_pin import action_features, survivor_action_stats, w1, closed_loop_summary


def evaluate(kw, sigma=0.1, episodes=512, procs=8, pool=None):
    jobs = [(kw, sigma, 1000 * s + 7, episodes // procs) for s in range(procs)]
    res = sum((pool.map(_episodes, jobs) if pool else map(_episodes, jobs)), [])
    out = closed_loop_summary([r[0] for r in res], [r[1] for r in res], sigma)
    out["ret_q"] = out["ret_q_5_25_50_75_95"]
    surv = [r[2] for r in res if r[2] is not None]
    if surv:
        out["survivor_action_stats"] = survivor_action_stats(surv)
    return out


if __name__ == "__main__":
    kw = dict(hyp=eval(sys.argv[1])) if len(sys.argv) > 1 else {}
    with Pool(8) as p:
        for sigma in (0.1, 1e-4):
            print(evaluate(kw, sigma=sigma, pool=p))
