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

REF_LAST1000 = None


def ref_returns():
    global REF_LAST1000
    if REF_LAST1000 is None:
        REF_LAST1000 = np.load(os.path.join(ROOT, "tests", "golden", "ref_training_log_summary.npz"))["last1000_returns"]
    return REF_LAST1000


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


def action_features(A):
    """Steady-gait statistics of a 500-step action sequence (first 100 steps dropped)."""
    A = np.asarray(A, dtype=np.float64)[100:]
    X = A - A.mean(0)
    u, s, vt = np.linalg.svd(X, full_matrices=False)
    pc = u[:, 0] * s[0]
    ac = np.correlate(pc, pc, "full")[len(pc) - 1:]; ac /= ac[0]
    k0 = int(np.argmin(ac[:60])); k = k0 + int(np.argmax(ac[k0:k0 + 80]))
    return dict(mean=A.mean(0), std=A.std(0), sat=(np.abs(A) > 0.995).mean(0), period=k, ac_peak=float(ac[k]))


LOG_FEAT = action_features(ACTS)


def w1(a, b):
    """1-Wasserstein distance between two samples (quantile form)."""
    q = np.linspace(0.005, 0.995, 199)
    return float(np.abs(np.quantile(a, q) - np.quantile(b, q)).mean())


def evaluate(kw, sigma=0.1, episodes=512, procs=8, pool=None):
    jobs = [(kw, sigma, 1000 * s + 7, episodes // procs) for s in range(procs)]
    res = sum((pool.map(_episodes, jobs) if pool else map(_episodes, jobs)), [])
    L = np.array([r[0] for r in res]); R = np.array([r[1] for r in res])
    ref = ref_returns()
    out = dict(sigma=sigma, episodes=len(res), mean_length=float(L.mean()), early_falls_lt50=float((L < 50).mean()), full_length=float((L >= 500).mean()),
               ret_mean=float(R.mean()), ret_q=[float(v) for v in np.quantile(R, [0.05, 0.25, 0.5, 0.75, 0.95])], ret_max=float(R.max()),
               w1_to_reference_last1000=w1(R, ref))
    surv = [r[2] for r in res if r[2] is not None]
    if surv:
        F = [action_features(a) for a in surv]
        d = lambda k: float(np.abs(np.mean([f[k] for f in F], 0) - LOG_FEAT[k]).mean())
        out["survivor_action_stats"] = dict(n=len(surv), mean_abs_diff_of_channel_means=d("mean"), of_stds=d("std"), of_saturation=d("sat"),
                                            period_median=float(np.median([f["period"] for f in F])), period_log=LOG_FEAT["period"],
                                            ac_peak=float(np.mean([f["ac_peak"] for f in F])), ac_peak_log=LOG_FEAT["ac_peak"])
    return out


if __name__ == "__main__":
    kw = dict(hyp=eval(sys.argv[1])) if len(sys.argv) > 1 else {}
    with Pool(8) as p:
        for sigma in (0.1, 1e-4):
            print(evaluate(kw, sigma=sigma, pool=p))
