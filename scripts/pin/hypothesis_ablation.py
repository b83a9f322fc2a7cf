#!/usr/bin/env python3
"""Hypothesis ablation of the physics restatement against the reference-held pins.

Runs on the CPU ORACLE (test infrastructure; every switch below exists only there) -- one hypothesis of DESIGN.md section 2 at a time:

  pin R_t      teacher-forced residual of the PyBullet-held pin (scripts/pin/pin_eval.py): the reference's recorded command log is
               a_t = actor_3229999(obs_t^PyBullet); the oracle is driven open loop by a_0..a_{t-1} and R_t = rms over the unsaturated
               channels of atanh(a_t) - preactivation(actor(obs_t^oracle)).  R_0 pins the reset stance (8 settle substeps), R_1 one
               control step from it, R_2.. the accumulated trajectory (chaotic: only R_0 and R_1 are smooth in the parameters).
  open loop    control steps survived replaying the 500 recorded commands.
  shipped actor, sigma = 0.01 (walk_eval.py:83-85 + the test's action noise), 128 episodes: mean length / return, full-length fraction.
  stance       zero-action stance for 2 s: torso height against init_height = 0.160178937611 (plen_env.py:70) and its drift over the last second.

Writes profiles/r03_hypothesis_ablation.json.  ~6 min on 8 cores.
"""
import json, os, sys, time
import numpy as np
from multiprocessing import Pool
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pin_eval import make_env, residuals, survive, ACTS, SD, ROOT
from pin_eval import pre as _pre
pre = lambda sd, x: _pre(x)

VARIANTS = [("baseline (DESIGN section 2 defaults)", {})]
def V(name, **kw): VARIANTS.append((name, kw))
V("inertia from URDF <inertia> (URDF_USE_INERTIA_FROM_FILE)", urdf_inertia=True)
for v in (0.05, 0.09, 0.11, 0.2): V("motor kp %g" % v, hyp=dict(kp=v))
for v in (0.5, 0.9, 1.1, 2.0): V("motor kd %g" % v, hyp=dict(kd=v))
for v in (0.1, 0.14, 0.16, 0.2, 1.0): V("motor max force %g" % v, hyp=dict(max_force=v))
for v in (8.0, 30.0): V("motor velocity clamp %g rad/s (URDF velocity=8)" % v, hyp=dict(rhs_clamp=v))
for v in (1e-4, 1e-3): V("joint damping %g" % v, hyp=dict(joint_damping=v))
for v in (10, 49, 51, 100, 200): V("solver iterations %d" % v, hyp=dict(iters=v))
V("motor rows in DoF order (no quickSort scramble)", hyp=dict(nc_order=1))
V("non-contact rows not reversed on even iterations", hyp=dict(no_flip=1))
for v in (0.04, 0.06, 0.1, 0.2): V("contact erp2 %g" % v, hyp=dict(erp2=v))
for v in (0.0, 1e-4): V("linear slop %g" % v, hyp=dict(slop=v))
for v in (0.4, 0.5, 0.8, 1.0): V("lateral friction %g" % v, hyp=dict(mu=v))
V("rolling friction rows off", hyp=dict(roll=0.0))
V("rolling friction 0.008 (joint_act value)", hyp=dict(roll=0.008))
V("rolling friction 0.04", hyp=dict(roll=0.04))
V("spinning friction rows off", hyp=dict(spin=0.0))
V("rolling + spinning rows off", hyp=dict(roll=0.0, spin=0.0))
V("pyramid friction (two independent lateral rows)", hyp=dict(pyramid=1))
V("restitution 0", hyp=dict(rest=0.0))
V("restitution threshold off (always bounce)", hyp=dict(rest_thr=0.0))
V("multibody damping 0.04 / 0.04 (btMultiBody defaults kept)", hyp=dict(lin_damp=0.04, ang_damp=0.04))
V("base gyroscopic term off", hyp=dict(gyro_off=1))
V("box colliders of non-foot links off", hyp=dict(body_contacts=0))
V("persistent manifold (btPersistentManifold restated, friction anchors)", hyp=dict(manifold=1))
V("persistent manifold, frictionERP 0", hyp=dict(manifold=1, friction_erp=0.0))
V("persistent manifold + warm starting 0.1", hyp=dict(manifold=1, warm=0.1))
V("persistent manifold + warm starting 1.0", hyp=dict(manifold=1, warm=1.0))
V("persistent manifold, torsional rows on the first point only", hyp=dict(manifold=1, tors_pts=1))
# the manifold family around "every in-range hull vertex goes through addContactPoint each pass" (all 209 vertices: the sole's bevel rings count)
P = dict(manifold=1, man_add_all=1, friction_erp=0.0, man_drift=1e6)
V("persistent, all in-range hull vertices per pass, no anchors (P)", hyp=dict(P))
V("P with the drift test", hyp=dict(P, man_drift=1.0))
V("P rebuilt from nothing every pass (memoryless)", hyp=dict(P, man_fresh=1))
V("P, candidates = the 32 sole vertices", hyp=dict(P, man_cand=1))
V("P, candidates = the 8 corner representatives", hyp=dict(P, man_cand=2))
V("P, lowest candidate first", hyp=dict(P, man_order=1))
V("P, merge radius x 2", hyp=dict(P, man_cache=2.0))
V("P, in-range threshold x 1.2", hyp=dict(P, man_range=1.2))
V("P, motor kp 0.09", hyp=dict(P, kp=0.09))
V("P, motor kp 0.11", hyp=dict(P, kp=0.11))
V("P, rolling friction 0.06", hyp=dict(P, roll=0.06))
V("P, rolling friction 0.1", hyp=dict(P, roll=0.1))
V("P, frictionERP 0.2", hyp=dict(P, friction_erp=0.2))
# the literal one-point-per-pass manifold whose FIRST point (flat-on-flat start: an artefact of EPA) is a scanned interior point of the sole
V("persistent manifold, first point near the sole centre (10, 6) mm", hyp=dict(manifold=1, man_p1=1, man_p1x=0.010, man_p1y=0.006))
V("persistent manifold, first point (4, -2) mm", hyp=dict(manifold=1, man_p1=1, man_p1x=0.004, man_p1y=-0.002))
V("persistent manifold, first point (16, 10) mm", hyp=dict(manifold=1, man_p1=1, man_p1x=0.016, man_p1y=0.010))


def actor_stats(kw, episodes=128, sigma=0.01, seed=0):
    rng = np.random.default_rng(seed)
    e = make_env(**kw)
    lens, rets = [], []
    for ep in range(episodes):
        obs = e.reset(); ret = 0.0
        for t in range(500):
            a = np.clip(np.tanh(pre(SD, obs)) + sigma * rng.standard_normal(18), -1, 1).astype(np.float32)
            obs, r, done, _ = e.step(a.astype(np.float64)); ret += r
            if done:
                break
        lens.append(t + 1); rets.append(ret)
    lens = np.array(lens)
    return dict(episodes=episodes, sigma=sigma, mean_length=float(lens.mean()), mean_return=float(np.mean(rets)), full_length_fraction=float((lens >= 500).mean()))


def stance(kw):
    e = make_env(**kw); e.reset()
    z = []
    for t in range(120):
        obs, _, _, _ = e.step(np.zeros(18))   # agent-space 0 is not joint zero: use raw substeps instead
        z.append(obs[18])
    return z


def stance_raw(kw):
    e = make_env(**kw)
    s = np.zeros(49); s[2] = 0.158; s[6] = 1.0
    e.reset(); e.set_state(s); e.set_targets(np.zeros(18))
    z = []
    for k in range(480):
        e.substep(); z.append(e.get_state()[2])
    z = np.array(z)
    return dict(z_after_8_substeps=float(z[7]), z_max=float(z.max()), z_at_1s=float(z[239]), z_at_2s=float(z[479]),
                drift_last_second_mm=float((z[479] - z[239]) * 1e3), init_height_error_mm_at_2s=float((z[479] - 0.160178937611) * 1e3),
                init_height_error_mm_best=float(np.abs(z - 0.160178937611).min() * 1e3))


def run(item):
    name, kw = item
    t0 = time.time()
    R, _ = residuals(make_env(**kw), K=8)
    out = dict(name=name, switches={k: (v if not isinstance(v, dict) else v) for k, v in kw.items()},
               pin=dict(R=[None if np.isnan(x) else round(float(x), 5) for x in R], R0=float(R[0]), R1=float(R[1]), sum_R1_R4=float(np.nansum(R[1:5]))),
               open_loop_steps_survived=survive(make_env(**kw)), shipped_actor=actor_stats(kw), stance=stance_raw(kw))
    out["seconds"] = round(time.time() - t0, 1)
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    with Pool(n) as p:
        res = p.map(run, VARIANTS, chunksize=1)
    base = res[0]
    for r in res:
        r["pin"]["R0_vs_baseline"] = round(r["pin"]["R0"] / base["pin"]["R0"], 2)
        r["pin"]["R1_vs_baseline"] = round(r["pin"]["R1"] / base["pin"]["R1"], 2)
    out = dict(what=__doc__.split("\n\n")[0], pin_source="tests/golden/policy_cmd_sequence.npz + tests/golden/policy_3229999.npz",
               variants=res)
    path = os.path.join(ROOT, "profiles", "r03_hypothesis_ablation.json")
    json.dump(out, open(path, "w"), indent=1)
    print("%-70s %7s %6s %6s | %4s | %6s %7s %5s | %8s %7s" % ("variant", "R0", "R1", "R1-4", "open", "len", "ret", "full", "z2s-h0mm", "drift"))
    for r in res:
        print("%-70s %7.4f %6.3f %6.2f | %4d | %6.1f %7.1f %5.2f | %8.3f %7.3f" % (r["name"][:70], r["pin"]["R0"], r["pin"]["R1"], r["pin"]["sum_R1_R4"],
              r["open_loop_steps_survived"], r["shipped_actor"]["mean_length"], r["shipped_actor"]["mean_return"], r["shipped_actor"]["full_length_fraction"],
              r["stance"]["init_height_error_mm_at_2s"], r["stance"]["drift_last_second_mm"]))
