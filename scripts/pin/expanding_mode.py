#!/usr/bin/env python3
"""Mechanism of the "expanding mode" of the solver iteration in the reference configuration.

Runs on the CPU ORACLE (test infrastructure).  Along closed-loop episodes of the shipped actor (the states the benchmark's policy legs and the
reference's own training visit) every control step is run twice, from the state and from the state with the joint angles moved by 1e-9 rad:

  amplification  = max |obs difference after the step| / 1e-9           (a smooth step map gives O(1..100))

For ~100 amplifying (> 1e4) and ~100 benign (< 30) steps the solver is traced: per substep and per PGS iteration k the difference d_k of the
two delta-velocity vectors and the iteration's residual (oracle_set_trace).  An expanding iteration shows d_k growing geometrically while the
residual stalls; the growth factor is the spectral radius of the (locally linear) iteration map.  A/B on the SAME start states:
  * the bounds of the spinning / rolling rows no longer rewritten from the normal impulse after iteration k (tors_freeze = k),
  * rolling friction coefficient 0.08 m (reference: 0.1 x 0.8, plen_env.py:439-456) -> 0.04, 0.02, 0.01, 0,
  * torsional rows on the first point of a foot only,
  * spinning rows off.
Writes profiles/r04_expanding_mode.json.
"""
import json, os, sys
import numpy as np
from multiprocessing import Pool
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pin_eval import make_env, ROOT, pre
from oracle.oracle import agent_to_env, _dp

EPS = 1e-9


def collect(seed, episodes=6):
    """(state49, action18) pairs along closed-loop episodes, sigma = 0.05."""
    rng = np.random.default_rng(seed); e = make_env(); out = []
    for ep in range(episodes):
        obs = e.reset()
        for t in range(500):
            a = np.clip(np.tanh(pre(obs)) + 0.05 * rng.standard_normal(18), -1, 1).astype(np.float32)
            out.append((e.get_state(), a.copy()))
            obs, r, d, _ = e.step(a.astype(np.float64))
            if d:
                break
    return out


def amp_of(args):
    hyp, state, action, seed, trace = args
    rng = np.random.default_rng(seed)
    e1, e2 = make_env(hyp=hyp), make_env(hyp=hyp)
    e1.reset(); e2.reset()
    s2 = state.copy(); s2[13:31] += EPS * rng.standard_normal(18)
    e1.set_state(state); e2.set_state(s2)
    tg = np.array([agent_to_env(d, float(action[d])) for d in range(18)])
    e1.set_targets(tg); e2.set_targets(tg)
    tr = []
    b1, b2 = np.zeros((50, 34)), np.zeros((50, 34))
    for sub in range(4):
        if trace:
            e1.lib.oracle_set_trace(_dp(b1), 50); e1.substep()
            e1.lib.oracle_set_trace(_dp(b2), 50); e2.substep()
            e1.lib.oracle_set_trace(None, 0)
            n1, n2 = e1.contacts()["iterations"], e2.contacts()["iterations"]
            n = min(n1, n2)
            d = np.abs(b1[:n, 2:26] - b2[:n, 2:26]).max(1)
            tr.append(dict(ncp=e1.contacts()["ncp"], iterations=n, d=d.tolist(), residual=b1[:n, 0].tolist(),
                           normal_impulses=b1[n - 1, 26:34].tolist()))
        else:
            e1.substep(); e2.substep()
    x1, x2 = e1.get_state(), e2.get_state()
    amp = float(np.abs(x1[:31] - x2[:31]).max() / EPS)
    return amp, tr


def growth(d):
    """Geometric growth factor per iteration of a difference trace over its last 20 iterations (1 = neutral)."""
    d = np.maximum(np.asarray(d), 1e-300)
    if len(d) < 25:
        return 1.0
    k = np.arange(len(d) - 20, len(d))
    return float(np.exp(np.polyfit(k, np.log(d[k]), 1)[0]))


if __name__ == "__main__":
    with Pool(8) as p:
        pairs = sum(p.map(collect, range(8)), [])
        print("control steps collected:", len(pairs), flush=True)
        base = p.map(amp_of, [({}, s, a, i, False) for i, (s, a) in enumerate(pairs)])
        amps = np.array([b[0] for b in base])
        hi = [i for i in np.argsort(-amps) if amps[i] > 1e4][:100]
        lo = [i for i in np.where(amps < 30)[0]][:100]
        out = dict(what=__doc__, control_steps=len(pairs), eps=EPS,
                   amplification=dict(median=float(np.median(amps)), p75=float(np.quantile(amps, 0.75)), p90=float(np.quantile(amps, 0.9)), p99=float(np.quantile(amps, 0.99)),
                                      frac_gt_1e3=float((amps > 1e3).mean()), frac_gt_1e6=float((amps > 1e6).mean())))
        print(out["amplification"], flush=True)
        traces = {}
        for name, idx in (("amplifying", hi), ("benign", lo)):
            res = p.map(amp_of, [({}, pairs[i][0], pairs[i][1], i, True) for i in idx])
            g, ncp, stall, lam = [], [], [], []
            keep = []
            for (amp, tr), i in zip(res, idx):
                gs = [growth(t["d"]) for t in tr]
                k = int(np.argmax(gs))
                g.append(max(gs)); ncp.append(tr[k]["ncp"]); lam.append(sum(tr[k]["normal_impulses"]))
                r = np.array(tr[k]["residual"]); stall.append(float(r[-1] / max(r[len(r) // 2], 1e-300)))
                if len(keep) < 6:
                    keep.append(dict(amplification=amp, substep=k, ncp=tr[k]["ncp"], d_every_5th_iteration=[float(x) for x in tr[k]["d"][::5]],
                                     residual_every_5th_iteration=[float(x) for x in tr[k]["residual"][::5]], normal_impulses_last=tr[k]["normal_impulses"]))
            traces[name] = dict(n=len(idx), growth_factor_per_iteration=dict(median=float(np.median(g)), p10=float(np.quantile(g, 0.1)), p90=float(np.quantile(g, 0.9)), max=float(np.max(g))),
                                contact_points_in_the_worst_substep=dict(zip(*[x.tolist() for x in np.unique(ncp, return_counts=True)])) if ncp else {},
                                residual_last_over_mid=float(np.median(stall)),
                                sum_normal_impulse_over_weight_impulse=float(np.median(lam) / (0.495834 * 9.81 / 240)), examples=keep)
            print(name, traces[name]["growth_factor_per_iteration"], traces[name]["contact_points_in_the_worst_substep"], flush=True)
        out["traces"] = traces
        ab = {}
        sel = hi + lo
        for name, hyp in [("reference configuration", {}), ("torsional bounds frozen after iteration 25", dict(tors_freeze=25)), ("frozen after iteration 10", dict(tors_freeze=10)),
                          ("frozen after iteration 1", dict(tors_freeze=1)), ("rolling friction 0.04 m", dict(roll=0.04)), ("rolling friction 0.02 m", dict(roll=0.02)),
                          ("rolling friction 0.01 m", dict(roll=0.01)), ("rolling rows off", dict(roll=0.0)), ("spinning rows off", dict(spin=0.0)),
                          ("torsional rows on the first point only", dict(tors_pts=1)), ("100 iterations", dict(iters=100)), ("pyramid friction", dict(pyramid=1))]:
            r = p.map(amp_of, [(hyp, pairs[i][0], pairs[i][1], i, False) for i in sel])
            a = np.array([x[0] for x in r])
            ab[name] = dict(on_the_100_amplifying=dict(median=float(np.median(a[:len(hi)])), frac_gt_1e3=float((a[:len(hi)] > 1e3).mean())),
                            on_the_100_benign=dict(median=float(np.median(a[len(hi):])), frac_gt_1e3=float((a[len(hi):] > 1e3).mean())))
            print("%-46s amplifying: median %.3g, > 1e3: %.2f | benign: median %.3g, > 1e3: %.2f" % (name, ab[name]["on_the_100_amplifying"]["median"], ab[name]["on_the_100_amplifying"]["frac_gt_1e3"],
                  ab[name]["on_the_100_benign"]["median"], ab[name]["on_the_100_benign"]["frac_gt_1e3"]), flush=True)
        out["ab_on_the_same_states"] = ab
    json.dump(out, open(os.path.join(ROOT, "profiles", "r04_expanding_mode.json"), "w"), indent=1)
