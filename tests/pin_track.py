"""The simulator under test as an extended Kalman OBSERVER of the reference's recorded PyBullet episode (test infrastructure).

tests/pybullet_pin.py turns the reference's 500-step command log into 500 x 18 equations a_t = actor_3229999(obs_t^PyBullet) on PyBullet's
observations (plen_env.py:597-608, walk_eval.py:83-85).  Open loop only the first steps are usable (the trajectory decorrelates), and a
per-step min-norm correction of the observation (`pybullet_pin.track`) loses the episode after ~20 steps: ~12 unsaturated equations per
step do not determine 24 observation entries, and nothing corrects the velocities.  An EKF does both: the state error covariance is carried
through the finite-difference Jacobian of the simulator's own control step, so equations of earlier steps keep constraining later ones and
position innovations correct the velocities through the cross-covariances.

  error state (47): base y, z | base rotation (world-frame rotation vector) | base omega | base v | q18 | qd18      (x is unobservable)
  measurement     : pre-activations of the actor on its unsaturated channels, flags decoded (a flipped flag moves them by 30-180)
  Rhat_t          : rms of the innovation = the one-step-ahead residual, BEFORE the update, along PyBullet's own trajectory

Whatever is under test only has to provide reset / step / get_state / set_state / clone (oracle: OracleEnv.copy_from).
"""
import numpy as np
from pybullet_pin import ACTS, FLAGS, pre, target

NX = 47


def qmul(a, b):   # (x, y, z, w)
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def rotvec_quat(th):
    a = np.linalg.norm(th)
    k = 0.5 - a * a / 48.0 if a < 1e-4 else np.sin(0.5 * a) / a
    return np.array([th[0] * k, th[1] * k, th[2] * k, np.cos(0.5 * a)])


def quat_rotvec(q):
    if q[3] < 0:
        q = -q
    n = np.linalg.norm(q[:3])
    if n < 1e-12:
        return 2.0 * q[:3]
    return q[:3] / n * (2.0 * np.arctan2(n, q[3]))


def inject(s, d):
    s = np.array(s, dtype=np.float64)
    s[1] += d[0]; s[2] += d[1]
    q = qmul(rotvec_quat(d[2:5]), s[3:7]); s[3:7] = q / np.linalg.norm(q)
    s[7:10] += d[5:8]; s[10:13] += d[8:11]; s[13:31] += d[11:29]; s[31:49] += d[29:47]
    return s


def diff(s1, s0):
    d = np.zeros(NX)
    d[0] = s1[1] - s0[1]; d[1] = s1[2] - s0[2]
    q0 = s0[3:7]; qc = np.array([-q0[0], -q0[1], -q0[2], q0[3]])
    d[2:5] = quat_rotvec(qmul(s1[3:7], qc))
    d[5:8] = s1[7:10] - s0[7:10]; d[8:11] = s1[10:13] - s0[10:13]; d[11:29] = s1[13:31] - s0[13:31]; d[29:47] = s1[31:49] - s0[31:49]
    return d


def euler(q):     # pybullet getEulerFromQuaternion, regular branch
    x, y, z, w = q
    sarg = -2 * (x * z - w * y)
    sarg = min(1.0, max(-1.0, sarg))
    return np.array([np.arctan2(2 * (y * z + w * x), w * w - x * x - y * y + z * z), np.arcsin(sarg), np.arctan2(2 * (x * y + w * z), w * w + x * x - y * y - z * z)])


def obs_of_state(s, flags):
    o = np.zeros(26)
    o[:18] = s[13:31]; o[18] = s[2]; o[19] = s[10]; o[20:23] = euler(s[3:7]); o[23] = s[1]; o[24], o[25] = flags
    return o


# process noise per control step (std): what one step of an imperfect contact model may add
Q_STD = np.concatenate([[3e-4, 3e-4], [2e-3] * 3, [0.3] * 3, [0.02] * 3, [3e-3] * 18, [0.5] * 18])
P0_STD = np.concatenate([[1e-4, 1e-4], [1e-3] * 3, [0.05] * 3, [0.005] * 3, [1e-3] * 18, [0.1] * 18])


PMAX_STD = 10.0 * Q_STD


class Stepper(object):
    """Adapter: clone-able simulator.  make() -> fresh env (same hypotheses); envs need reset/step/get_state/set_state/copy_from."""

    def __init__(self, make):
        self.make = make
        self.env = make()
        self.scratch = make()

    def step_from(self, state, action):
        """One control step of a COPY of self.env started from `state`: returns (next state, flags (right, left))."""
        self.scratch.copy_from(self.env)
        self.scratch.set_state(state)
        o = self.scratch.step(action)[0]
        return self.scratch.get_state(), (o[24], o[25])


def ekf_track(make, T=500, meas_std=0.05, q_scale=1.0, fd_eps=1e-6, gate=25.0, one_sided=True, verbose=False, acts=None, fd_sigma=0.0, fclamp=0.0):
    """Returns dict(Rhat[T], nis[T], flags_sim[T,2], flags_dec[T,2], steps)."""
    ACTS_ = ACTS if acts is None else acts
    T = min(T, len(ACTS_))
    st = Stepper(make)
    env = st.env
    obs = np.array(env.reset(), dtype=np.float64)
    x = env.get_state()
    P = np.diag(P0_STD ** 2)
    Q = np.diag((q_scale * Q_STD) ** 2)
    flags_sim = (obs[24], obs[25])
    Rhat, NIS, FS, FD, NU = [], [], [], [], []
    for t in range(T):
        tgt, un = target(ACTS_[t])
        # ---- decode the flags, innovation ----
        r4 = []
        for f in FLAGS:
            r4.append(np.sqrt((((pre(obs_of_state(x, f)) - tgt)[un]) ** 2).mean()))
        k = int(np.argmin(r4)); fl = FLAGS[k]
        nu = (tgt - pre(obs_of_state(x, fl)))[un]
        Rhat.append(float(np.sqrt((nu ** 2).mean()))); FS.append(flags_sim); FD.append(fl); NU.append(int(un.sum()))
        # ---- measurement update ----
        H = np.zeros((18, NX))
        h0 = pre(obs_of_state(x, fl))
        for i in range(NX):
            d = np.zeros(NX); d[i] = 1e-6
            H[:, i] = (pre(obs_of_state(inject(x, d), fl)) - h0) / 1e-6
        H = H[un]
        S = H @ P @ H.T + (meas_std ** 2) * np.eye(H.shape[0])
        Sinv = np.linalg.inv(S)
        nis = float(nu @ Sinv @ nu) / max(1, len(nu)); NIS.append(nis)
        if nis > gate:                                   # outlier (a discrete contact event went the other way): soften the update
            S = H @ P @ H.T + (meas_std ** 2) * (nis / gate) * np.eye(H.shape[0]) + (nis / gate - 1) * np.diag(np.diag(H @ P @ H.T))
            Sinv = np.linalg.inv(S)
        K = P @ H.T @ Sinv
        dx = K @ nu
        x = inject(x, dx)
        I_KH = np.eye(NX) - K @ H
        P = I_KH @ P @ I_KH.T + (meas_std ** 2) * (K @ K.T)
        # ---- time update through the simulator's own step ----
        a = ACTS_[t].astype(np.float64)
        env.set_state(x)
        x1, fl1 = st.step_from(x, a)
        F = np.zeros((NX, NX))
        for i in range(NX):
            d = np.zeros(NX); d[i] = fd_sigma * np.sqrt(P[i, i]) if fd_sigma > 0 else fd_eps * max(1.0, Q_STD[i] / 3e-3)
            xp, _ = st.step_from(inject(x, d), a)
            if one_sided:
                F[:, i] = diff(xp, x1) / d[i]
            else:
                xm, _ = st.step_from(inject(x, -d), a)
                F[:, i] = diff(xp, xm) / (2 * d[i])
        # the real step (advances the env's own contact cache / bookkeeping)
        o = env.step(a)[0]
        x = env.get_state(); flags_sim = (o[24], o[25])
        if not np.all(np.isfinite(x)) or not np.all(np.isfinite(F)):
            break
        # bound the amplification of a single step: the step map has discontinuities
        Fn = np.linalg.norm(F, 2)
        if fclamp > 0 and Fn > fclamp:
            F *= fclamp / Fn
        P = F @ P @ F.T + Q
        P = 0.5 * (P + P.T)
        lim = np.minimum(1.0, (PMAX_STD * q_scale) / np.sqrt(np.diag(P)))      # the step map has discontinuities: bound the uncertainty
        P = P * lim[:, None] * lim[None, :]
        if verbose and t % 25 == 0:
            print(t, "Rhat %.3f nis %.2f |F| %.1f trP %.3g" % (Rhat[-1], nis, Fn, np.trace(P)), flush=True)
    return dict(Rhat=np.array(Rhat), nis=np.array(NIS), flags_sim=np.array(FS), flags_dec=np.array(FD), nun=np.array(NU), steps=len(Rhat))


def summary(tr, skip=5):
    R = tr["Rhat"]
    same = (tr["flags_sim"] == tr["flags_dec"]).all(1)
    st = R[skip:]
    return dict(steps=int(tr["steps"]), median=float(np.median(st)), mean=float(st.mean()), p75=float(np.quantile(st, 0.75)), p90=float(np.quantile(st, 0.9)),
                first=[round(float(v), 4) for v in R[:skip]], flags_agree=float(same.mean()),
                median_when_flags_agree=float(np.median(R[same])) if same.any() else None, frac_below_0p3=float((st < 0.3).mean()))
