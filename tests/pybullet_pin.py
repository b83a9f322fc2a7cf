"""The PyBullet-held pin of the physics (test infrastructure, shared by tests/, bench.py's parity block and scripts/pin/).

The reference's recorded command log (plen_bullet/trajectories/*_cmd.npy -> tests/golden/policy_cmd_sequence.npz, written by
plen_env.py:604-608 at the reset that ended a 500-step episode) is the DETERMINISTIC output of the shipped actor along that episode:
a_t = actor_3229999(obs_t^PyBullet) (walk_eval.py:83-85).  Evidence: of the seven shipped actors only 3229999 reproduces a_0 from this
repository's reset observation (max |da| 0.024 against 0.54-0.95 for the other six, tools/make_golden_cmd.py prints the table), and only
with the contact flags (right 0, left 1).  So the actor turns the log into 500 x 18 equations on PyBullet's own observation sequence:

  R_t = rms over the unsaturated channels of  atanh(a_t) - preactivation(actor(obs_t))

with obs_t produced by whatever is under test, driven open loop by a_0..a_{t-1} (what PyBullet was driven with).  R_0 pins the reset stance
(8 settle substeps from the spawn pose), R_1 one control step from it under a full-range command, R_2.. the accumulated trajectory
(chaotic in the reference configuration, DESIGN.md section 5: only R_0 and R_1 vary smoothly with the parameters).
The actor's Jacobian has column norms 7-180 per unit of observation, so R = 0.01 corresponds to observation errors of 1e-4..1e-3.
"""
import json
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

# oracle_set_hyp keys (oracle/plen_oracle.c)
HYP = dict(erp=0, erp2=1, friction_erp=2, cfm=3, slop=4, resid=5, rest_thr=6, maxvel=7, mu=8, box_mu=9, spin=10, roll=11, rest=12,
           lin_damp=13, ang_damp=14, kp=15, kd=16, max_force=17, iters=18, body_contacts=19, dt=20, manifold=21, warm=22, pyramid=23,
           gyro_off=24, tors_pts=25, rhs_clamp=26, joint_damping=27, nc_order=28, no_flip=29, man_cand=30, man_drift=31, man_add_all=32, man_fresh=33, man_order=34, man_cache=35, man_range=36, sole_grow=37, sole_dz=38, man_p1=39, man_p1x=40, man_p1y=41)


def load():
    acts = np.load(os.path.join(GOLD, "policy_cmd_sequence.npz"))["actions"]
    z = np.load(os.path.join(GOLD, "policy_3229999.npz"))
    sd = {k[6:]: z[k].astype(np.float64) for k in z if k.startswith("actor.")}
    return acts, sd


ACTS, SD = load()


def pre(x, sd=SD):
    """Pre-tanh output of the shipped actor (td3.py:19-57) in float64."""
    h = np.maximum(sd["fc1.weight"] @ x + sd["fc1.bias"], 0)
    h = np.maximum(sd["fc2.weight"] @ h + sd["fc2.bias"], 0)
    return sd["fc3.weight"] @ h + sd["fc3.bias"]


def jac(x, eps=1e-6):
    J = np.zeros((18, 26))
    for i in range(26):
        d = np.zeros(26); d[i] = eps
        J[:, i] = (pre(x + d) - pre(x - d)) / (2 * eps)
    return J


def target(a, sat=0.995):
    """atanh of a recorded action and the mask of channels that are not saturated (float32 tanh loses the argument beyond ~0.995)."""
    a = np.asarray(a, dtype=np.float64)
    return np.arctanh(np.clip(a, -0.9999999, 0.9999999)), np.abs(a) < sat


def residual(obs, t):
    tgt, un = target(ACTS[t])
    r = (pre(np.asarray(obs, dtype=np.float64)) - tgt)[un]
    return float(np.sqrt((r ** 2).mean()))


def min_norm_obs_correction(obs, t):
    """Smallest (scaled) change of the 24 continuous observation entries that reproduces a_t exactly, linearised: a LOWER bound on the
    distance to PyBullet's observation.  Scales: joints / angles 1e-2 rad, z and y 1e-3 m, vx 2e-2 m/s."""
    x = np.asarray(obs, dtype=np.float64)
    tgt, un = target(ACTS[t])
    w = np.array([1e-2] * 18 + [1e-3, 2e-2, 1e-2, 1e-2, 1e-2, 1e-3])
    J = jac(x)[un][:, :24]
    return w * np.linalg.lstsq(J * w, (tgt - pre(x))[un], rcond=1e-3)[0]


def residuals(reset, step, K=8):
    """R_0..R_K for an environment given as reset() -> obs[26] and step(action float32[18]) -> (obs[26], done)."""
    obs = reset()
    out, seq = [], [np.array(obs, dtype=np.float64)]
    for t in range(K + 1):
        out.append(residual(obs, t))
        obs, done = step(ACTS[t])
        seq.append(np.array(obs, dtype=np.float64))
        if done:
            out += [float("nan")] * (K - t)
            break
    return np.array(out), np.array(seq)


# ---- the oracle under the pin (tests and scripts/pin only) ----
def make_oracle(hyp=None, urdf_inertia=False):
    from oracle.oracle import OracleEnv
    e = OracleEnv()
    for k, v in (hyp or {}).items():
        assert e.lib.oracle_set_hyp(e.h, HYP[k], float(v)) == 0, k
    if urdf_inertia:
        m = json.load(open(os.path.join(ROOT, "plen_ml_walk_amd", "model", "plen_model.json")))
        e.lib.oracle_set_link_inertia(e.h, 0, *m["base"]["inertia_urdf"])
        for l in m["links"]:
            e.lib.oracle_set_link_inertia(e.h, l["index"] + 1, *l["inertia_urdf"])
    return e


def oracle_residuals(K=8, **kw):
    e = make_oracle(**kw)

    def step(a):
        o, _, d, _ = e.step(np.asarray(a, dtype=np.float64))
        return o, d
    return residuals(e.reset, step, K)
