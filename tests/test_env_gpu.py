"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the
CPU oracle on identical inputs.

Tolerances (written here, explained in DESIGN.md "Parity and conditioning"):
  * everything that is a smooth function of the state (kinematics, mass matrix, bias, Cholesky factor,
    port Delassus matrix, row right-hand sides, contact detection) must agree to rounding:
    f64 1e-10, f32 2e-4 relative;
  * the solver logic is exact: with 1..3 PGS iterations f64 agrees with the oracle to 1e-9;
  * with the reference's 50 iterations AND its rolling friction (0.1 -> 0.08 m combined), Bullet's
    iteration has an expanding mode in a fraction of contact states: the ORACLE ITSELF moves by >1e-2
    when its input is perturbed by 6e-8 (f32 epsilon) in ~10% of reachable states.  There, no
    implementation can match another to 1e-4; so the reference configuration is asserted on the
    median, on the fraction within north_star's 1e-4, and (f32) against the oracle's own sensitivity;
  * with rolling friction off the iteration is contractive and multi-step rollouts are asserted tightly.
"""
import json
import os
import subprocess
import sys
import numpy as np
import pytest
import torch

import np_model as nm
from oracle.oracle import OracleEnv, agent_to_env

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(n, dtype, **kw):
    from plen_ml_walk_amd.vec_env import PlenVecEnv
    return PlenVecEnv(n, dtype=dtype, **kw)


def collect_states(n, seed=3, amp=1.0, with_targets=False):
    """Reachable states: sampled along oracle rollouts with random actions."""
    rng = np.random.default_rng(seed)
    e = OracleEnv(); e.reset()
    S, T = [], []
    t = 0
    while len(S) < n:
        a = rng.uniform(-1, 1, 18) * amp
        _, _, d, _ = e.step(a); t += 1
        if d or t % 40 == 0:
            e.reset(); continue
        if t % 2 == 0:
            S.append(e.get_state()); T.append([agent_to_env(j, a[j]) for j in range(18)])
    return (np.array(S), np.array(T)) if with_targets else np.array(S)


def oracle_step_from(state, action, perturb=0.0, rng=None, **cfg):
    o = OracleEnv()
    if cfg.get("rolling") is not None:
        o.set_friction(rolling=cfg["rolling"])
    s = state if not perturb else state * (1 + perturb * rng.standard_normal(state.shape))
    o.set_state(s); o.script_reset()
    return o.step(action)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 1e-5)])
def test_reset_observation(dtype, tol):
    env = _env(5, dtype)
    obs = env.reset().cpu().numpy().astype(np.float64)
    ref = OracleEnv().reset()
    assert np.abs(obs - ref[None]).max() <= tol
    aux = env.get_aux().cpu().numpy()
    assert np.all(aux[:, :4] == 0) and np.all(aux[:, 4] == ref[24]) and np.all(aux[:, 5] == ref[25])
    env.close()


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 2e-4)])
def test_presolver_quantities(dtype, tol):
    """M, bias, Cholesky factor, port Delassus matrix, port velocities, contact distances of one substep
    (debug dump of the kernel) against the NumPy statement of the same formulation."""
    n = 48
    S, T = collect_states(n, seed=4, with_targets=True)
    env = _env(n, dtype)
    env.set_state(torch.tensor(S))
    d = env.debug_substeps(torch.tensor(T), nsub=1, dump=True).cpu().numpy().astype(np.float64)
    aux = env.get_aux().cpu().numpy()
    for i in range(n):
        info = {}
        nm.substep(S[i], T[i], info=info)
        rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
        assert rel(d[i, :576].reshape(24, 24), info["M"]) <= tol
        assert rel(d[i, 576:600], info["tau"]) <= tol * 10
        assert rel(np.tril(d[i, 640:1216].reshape(24, 24)), info["L"]) <= tol * 10
        assert rel(d[i, 1216:3520].reshape(48, 48), info["A"]) <= tol * 100
        assert np.abs(d[i, 3520:3568] - info["b"]).max() <= tol * 100 * max(1.0, np.abs(info["b"]).max())
        dist = d[i, 3568 + 18:3568 + 48].reshape(2, 15)[:, 3::3]
        assert np.abs(dist - info["dist"]).max() <= max(tol, 1e-7)
        assert aux[i, 4] == int(info["right"]) and aux[i, 5] == int(info["left"])
    env.close()


@pytest.mark.parametrize("nit", [1, 3])
def test_solver_logic_exact_f64(nit):
    n = 48
    S, T = collect_states(n, seed=5, with_targets=True)
    env = _env(n, torch.float64, cfg_overrides={"num_iterations": nit})
    env.set_state(torch.tensor(S))
    env.debug_substeps(torch.tensor(T), nsub=1)
    out = env.get_state().cpu().numpy()
    for i in range(n):
        o = OracleEnv(); o.set_world(num_iterations=nit); o.set_state(S[i]); o.set_targets(T[i]); o.substep()
        assert np.abs(out[i] - o.get_state()).max() <= 1e-9
    env.close()


def _rollout_vs_oracle(dtype, n, T, acts, rolling=None, joint_act=False, reward_head=0):
    ov = {} if rolling is None else {"rolling_friction": rolling}
    if reward_head:
        ov["reward_head"] = reward_head
    env = _env(n, dtype, joint_act=joint_act, cfg_overrides=ov)
    ors = []
    for _ in range(n):
        o = OracleEnv(joint_act=joint_act, reward_head=reward_head)
        if rolling is not None:
            o.set_friction(rolling=rolling)
        o.reset(); ors.append(o)
    env.reset()
    alive = np.ones(n, bool)
    eo, er, mism, compared = [], [], 0, 0
    for t in range(T):
        nobs, rew, done, _ = env.step(acts[t].cuda())
        nobs = nobs.cpu().numpy().astype(np.float64); rew = rew.cpu().numpy().astype(np.float64); fl = done.cpu().numpy()
        for i in range(n):
            if not alive[i]:
                continue
            ob, r, d, _ = ors[i].step(acts[t, i].numpy().astype(np.float64))
            eo.append(np.abs(ob - nobs[i]).max()); er.append(abs(r - rew[i]) / max(1.0, abs(r))); compared += 1
            mism += int(bool(fl[i] & 1) != d)
            if d or fl[i]:
                alive[i] = False
    env.close()
    return np.array(eo), np.array(er), mism, compared


def test_rollout_well_conditioned_f64():
    """Random actions, rolling friction off (contractive solver): whole first episodes agree."""
    g = torch.Generator().manual_seed(0)
    acts = (torch.rand(12, 24, 18, generator=g) * 2 - 1).float()
    eo, er, mism, n = _rollout_vs_oracle(torch.float64, 24, 12, acts, rolling=0.0)
    assert n > 200 and mism == 0
    assert eo.max() <= 1e-5 and np.median(eo) <= 1e-11 and er.max() <= 1e-5


def test_gazebo_reward_head_f64():
    """PlenWalkEnv-v0 contract (force-threshold contact flags, its done and reward) on the same physics.
    The oracle's head is pinned to the reference by tests/golden/gazebo_reward_done.npz."""
    g = torch.Generator().manual_seed(3)
    acts = (torch.rand(12, 24, 18, generator=g) * 2 - 1).float()
    eo, er, mism, n = _rollout_vs_oracle(torch.float64, 24, 12, acts, rolling=0.0, reward_head=1)
    assert n > 200 and mism == 0
    assert eo.max() <= 1e-4 and np.median(eo) <= 1e-11 and er.max() <= 1e-4     # obs includes the two force-threshold flags


def test_recorded_policy_action_sequence_f64(golden_dir):
    """The reference's own recorded policy commands (plen_bullet/trajectories/*_cmd.npy -> tests/golden/policy_cmd_sequence.npz),
    near-saturated bang-bang actions, replayed open loop (the robot tumbles after ~30 steps).  Kernel = oracle to rounding until
    contact events amplify the 1e-15 differences: 15 steps with rolling friction off, 4 in the reference configuration, whose
    solver iteration is expanding; SURVEY 8c item 5."""
    a = np.load(os.path.join(golden_dir, "policy_cmd_sequence.npz"))["actions"]        # [500, 18] float32
    for rolling, exact_steps, loose_steps in ((0.0, 15, 22), (None, 4, 4)):
        ov = {} if rolling is None else {"rolling_friction": rolling}
        env = _env(1, torch.float64, cfg_overrides=ov); env.reset()
        o = OracleEnv()
        if rolling is not None:
            o.set_friction(rolling=rolling)
        o.reset()
        errs = []
        for t in range(loose_steps):
            nobs, rew, done, _ = env.step(torch.tensor(a[t:t + 1]).cuda())
            ob, r, d, _ = o.step(a[t].astype(np.float64))
            errs.append(max(np.abs(ob - nobs[0].cpu().numpy()).max(), abs(r - float(rew[0])) / max(1.0, abs(r))))
            assert bool(int(done[0]) & 1) == d
        env.close()
        assert max(errs[:exact_steps]) <= 1e-9 and max(errs) <= 1e-5, (rolling, errs)


def test_recorded_policy_action_sequence_f32(golden_dir):
    """The same recorded commands through the f32 kernel: f32 rounding (6e-8) stays at the 1e-5 level for the 14 steps
    before the robot starts to tumble with rolling friction off, and for the first 2 steps in the reference configuration, whose solver
    amplifies it by ~50x per control step from the start (measured: 8e-7, 9e-6, 5e-4, ...; the f64 kernel: 2e-15, 1e-15, 5e-13, 5e-13, 4e-4)."""
    a = np.load(os.path.join(golden_dir, "policy_cmd_sequence.npz"))["actions"]
    for rolling, steps, tol in ((0.0, 14, 5e-5), (None, 2, 5e-5)):
        ov = {} if rolling is None else {"rolling_friction": rolling}
        env = _env(1, torch.float32, cfg_overrides=ov); env.reset()
        o = OracleEnv()
        if rolling is not None:
            o.set_friction(rolling=rolling)
        o.reset()
        errs = []
        for t in range(steps):
            nobs, rew, done, _ = env.step(torch.tensor(a[t:t + 1]).cuda())
            ob, r, d, _ = o.step(a[t].astype(np.float64))
            errs.append(max(np.abs(ob - nobs[0].cpu().numpy().astype(np.float64)).max(), abs(r - float(rew[0])) / max(1.0, abs(r)) * 1e-2))
            assert bool(int(done[0]) & 1) == d
        env.close()
        assert max(errs) <= tol, (rolling, errs)


def test_rollout_well_conditioned_f32():
    g = torch.Generator().manual_seed(0)
    acts = ((torch.rand(10, 16, 18, generator=g) * 2 - 1) * 0.3).float()
    eo, er, mism, n = _rollout_vs_oracle(torch.float32, 16, 10, acts, rolling=0.0)
    assert n >= 150 and mism == 0
    assert eo.max() <= 3e-3 and np.median(eo) <= 1e-4 and er.max() <= 3e-3


def test_reference_config_one_step_distribution():
    """The reference's own configuration (rolling friction 0.1): one control step from 128 reachable
    injected states, all 26 observation components, contact flags compared exactly."""
    n = 128
    S = collect_states(n, seed=3)
    rng = np.random.default_rng(9)
    A = rng.uniform(-1, 1, (n, 18)).astype(np.float32)
    ref = np.array([oracle_step_from(S[i], A[i].astype(np.float64))[0] for i in range(n)])
    rref = np.array([oracle_step_from(S[i], A[i].astype(np.float64))[1] for i in range(n)])
    sens = np.array([np.abs(oracle_step_from(S[i], A[i].astype(np.float64), 6e-8, rng)[0] - ref[i])[:24].max() for i in range(n)])
    for dtype in (torch.float64, torch.float32):
        env = _env(n, dtype)
        env.set_state(torch.tensor(S))
        nobs, rew, done, _ = env.step(torch.tensor(A).cuda())
        got = nobs.cpu().numpy().astype(np.float64)
        err = np.abs(got - ref)[:, :24].max(1)
        flags_bad = int((np.abs(got - ref)[:, 24:] > 0).any(1).sum())
        if dtype == torch.float64:
            assert np.median(err) <= 1e-12
            assert np.mean(err <= 1e-4) >= 0.75            # north_star's 1e-4 wherever the solver is well conditioned (measured 0.84)
            assert flags_bad <= 3
        else:
            assert np.median(err) <= 2e-5
            assert np.mean(err <= 1e-4) >= 0.45
            ratio = err / np.maximum(sens, 1e-6)
            # f32 rounding moves the result no more than f32-sized input noise moves the oracle itself
            assert np.median(ratio) <= 5 and np.quantile(ratio, 0.9) <= 60
            assert flags_bad <= 10        # of 128 (contact flags: 8 candidate corners per foot switch more often than 4)
        r = rew.cpu().numpy().astype(np.float64)
        # reward = smooth function of the observation (slope up to ~20 per unit) + discrete terms
        ok = err <= (1e-9 if dtype == torch.float64 else 1e-5)
        assert np.abs(r - rref)[ok].max() <= (1e-6 if dtype == torch.float64 else 2e-3)
        env.close()


def test_terminal_and_time_limit_semantics():
    """compute_done -> -100 and PLENVEC_DONE_TERMINAL; gym TimeLimit -> PLENVEC_DONE_TIMELIMIT; auto-reset
    hands back the reset observation and clears the counters (plen_env.py:1072-1093, :15-19; plen_td3.py:109-133)."""
    from plen_ml_walk_amd import _lib as L
    n = 4
    env = _env(n, torch.float64, cfg_overrides={"max_episode_steps": 3})
    reset_obs = env.reset().clone()
    s = env.get_state()
    s[1, 1] = 1.5                       # env 1 starts beyond the y > 1 threshold -> dead on the first step
    env.set_state(s)
    z = torch.zeros(n, 18)
    flags, rewards, cur = [], [], []
    for t in range(3):
        nobs, rew, done, info = env.step(z.cuda())
        flags.append(done.cpu().numpy().copy()); rewards.append(rew.cpu().numpy().copy()); cur.append(info["obs"].clone())
        if t == 0:
            assert bool(info["terminal"][1]) and not bool(info["time_limit"][1])
    assert flags[0][1] == L.DONE_TERMINAL and rewards[0][1] < -90
    assert torch.equal(cur[0][1], reset_obs[1])                       # env 1 was auto-reset
    assert flags[0][0] == 0 and flags[1][0] == 0 and (flags[2][0] & L.DONE_TIMELIMIT)
    assert flags[2][1] == 0                                           # env 1 restarted its clock at step 1
    aux = env.get_aux().cpu().numpy()
    assert aux[0, 2] == 0 and aux[0, 0] == 0 and aux[1, 2] == 2       # episode steps: env 0 reset at the limit, env 1 two steps in
    env.close()


def test_gait_counter_branches_long_episode_f64():
    """Episodes long enough to reach the gait-period branches (counter >= 80 / >= 120, double support >= 16):
    joint targets held at the standing pose in joint_act mode (rolling friction 0.008: well conditioned, and
    Bullet's per-link linear damping 0.1 is exercised); observations, rewards and flags must agree."""
    n, T = 2, 140
    acts = torch.zeros(T, n, 18)
    eo, er, mism, cmpd = _rollout_vs_oracle(torch.float64, n, T, acts, joint_act=True)
    assert cmpd >= 2 * 100 and mism == 0
    assert eo.max() <= 1e-6 and er.max() <= 1e-6 and np.median(eo) <= 1e-10


def test_joint_act_generated_gait_f64():
    """the open-loop gait from the trajectory generator played through joint_act mode."""
    from plen_ml_walk_amd.trajectory_generator import TrajectoryGenerator
    a = TrajectoryGenerator(num_DoubleSupport=20, num_SingleSupport=20, height=20.0, stride=20.0).walk_cycle_actions(cycles=1)[:60]
    acts = torch.tensor(np.repeat(a[:, None, :], 2, axis=1), dtype=torch.float32)
    eo, er, mism, cmpd = _rollout_vs_oracle(torch.float64, 2, acts.shape[0], acts, joint_act=True)
    assert cmpd >= 60 and mism == 0
    assert eo.max() <= 1e-7 and er.max() <= 1e-6


def test_reference_walking_trajectory_joint_act_f64():
    """The reference's own open-loop walking trajectory (trajectory_eval.py: 20 bend steps, then the assembled 18-joint gait, whose
    arrays are pinned to the reference's committed trajectories/*.npy by the CPU suite) through joint_act=True: kernel = oracle."""
    from plen_ml_walk_amd.trajectory_eval import assemble_joint_trajectories
    walk, bend = assemble_joint_trajectories()
    a = np.concatenate([np.tile(bend, (20, 1)), walk[:50]], 0)
    acts = torch.tensor(np.repeat(a[:, None, :], 2, axis=1), dtype=torch.float32)
    eo, er, mism, cmpd = _rollout_vs_oracle(torch.float64, 2, acts.shape[0], acts, joint_act=True)
    assert cmpd >= 40 and mism == 0
    assert np.median(eo) <= 1e-9 and eo.max() <= 1e-5 and er.max() <= 1e-5


def test_reference_walking_trajectory_joint_act_f32():
    """The reference's open-loop walking trajectory through the f32 kernel against the f64 oracle: 70 control steps (20 bend + 50 of the
    gait) without a single flag mismatch; the error grows smoothly from 5e-7 to ~2e-3 (f64: 1e-15 to 1e-12)."""
    from plen_ml_walk_amd.trajectory_eval import assemble_joint_trajectories
    walk, bend = assemble_joint_trajectories()
    a = np.concatenate([np.tile(bend, (20, 1)), walk[:50]], 0)
    acts = torch.tensor(np.repeat(a[:, None, :], 2, axis=1), dtype=torch.float32)
    eo, er, mism, cmpd = _rollout_vs_oracle(torch.float32, 2, acts.shape[0], acts, joint_act=True)
    assert cmpd >= 2 * 70 and mism == 0
    # (an error of exactly 1 is a foot-contact flag that switches one control step earlier or later in f32)
    assert eo[:24].max() <= 5e-5 and np.median(eo) <= 1e-3 and (eo > 0.5).sum() <= 4 and eo[eo <= 0.5].max() <= 2e-2


@pytest.mark.parametrize("dt", ["float32", "float64"])
def test_asm_path_bitwise_equals_compiler_path(tmp_path, dt):
    """The hand-scheduled row updates (f32: round 1; f64 motor pass, generic row and cone tail: round 3) vs the compiler-generated ones: same
    operations in the same order, so 20 steps x 256 envs must agree bit for bit (any pipeline hazard -- a wait state too few -- would show)."""
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
            "from plen_ml_walk_amd.vec_env import PlenVecEnv\n"
            "g = torch.Generator().manual_seed(1); acts = (torch.rand(20, 256, 18, generator=g) * 2 - 1).float().cuda()\n"
            "env = PlenVecEnv(256, dtype=torch.%s); env.reset(); out = []\n"
            "for t in range(20):\n"
            "    o, r, d, _ = env.step(acts[t]); out.append(torch.cat([o, r[:, None], d.to(o.dtype)[:, None]], 1).cpu().numpy().copy())\n"
            "np.save(sys.argv[1], np.array(out))\n" % (ROOT, dt))
    outs = []
    for tag, extra in (("asm", {}), ("noasm", {"PLENVEC_NO_ASM": "1"})):
        p = str(tmp_path / (tag + ".npy"))
        subprocess.run([sys.executable, "-c", code, p], check=True, timeout=300, env=dict(os.environ, **extra))
        outs.append(np.load(p))
    assert np.array_equal(outs[0], outs[1], equal_nan=True)


@pytest.mark.parametrize("dt,cfg", [("float32", "reference"), ("float64", "reference"), ("float64", "no_torsional_friction"), ("float32", "fallen_robots")])
def test_count_specialised_solver_loops_bitwise_equal_the_run_time_tested_loop(tmp_path, dt, cfg):
    """Round 4: the solver's iteration loop is compiled once per pair of foot point counts and chosen once per substep.  That changes which code runs, not one
    operation or its order: against a build with rounds 1-3's run-time point tests (-DPLENVEC_COUNT_SPECIALISED=0, csrc/variants/nospec.so, built by
    __graft_entry__.build()) 40 steps x 512 envs -- random actions with amplitude 1.7 in joint_act mode for half of them, so that joint limits are violated
    and every contact configuration from airborne to both feet planted occurs -- agree bit for bit, including states and auto-resets.
    The specialised copies run the spinning / rolling rows unconditionally and rely on (+-0, +-0) bounds being exact no-ops when a coefficient is zero, and on
    lent box slots carrying zero coefficients: `no_torsional_friction` sets both coefficients to 0, `fallen_robots` switches the auto-reset off so
    that robots fall and stay down -- links other than the feet on the ground, contact slots lent to their box corners.
    Round 5: the same twin also builds the Delassus matrix A = Y^T Y as vector multiply-adds from broadcast LDS reads (-DPLENVEC_MFMA_DELASSUS=0) where the shipped
    kernel uses `v_mfma_f{32,64}_16x16x4` tiles (dense when a slot is lent, the structurally zero pieces skipped otherwise), and commits / zeroes its delta vectors
    after every pass where the shipped loops do neither: bit for bit the same trajectories says the matrix instruction adds its k terms in order, and that the hoisted
    commits touch no value a row reads.  (The mass matrix's matrix-core build has no such twin: switching it off changes the compiler's contractions elsewhere in
    its phase; it is held to the oracle like everything else.)"""
    from plen_ml_walk_amd.build import build_variant, REFERENCE_FORM_FLAGS
    lib0 = build_variant("nospec", REFERENCE_FORM_FLAGS)
    kw = {"reference": "", "no_torsional_friction": ", cfg_overrides={'spinning_friction': 0.0, 'rolling_friction': 0.0}", "fallen_robots": ", auto_reset=False"}[cfg]
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
            "from plen_ml_walk_amd.vec_env import PlenVecEnv\n"
            "out, st = [], []\n"
            "for ja, amp in ((False, 1.0), (True, 1.7)):\n"
            "    g = torch.Generator().manual_seed(5); acts = ((torch.rand(40, 512, 18, generator=g) * 2 - 1) * amp).float().cuda()\n"
            "    env = PlenVecEnv(512, dtype=torch.%s, joint_act=ja%s); env.reset()\n"
            "    for t in range(40):\n"
            "        o, r, d, _ = env.step(acts[t]); out.append(torch.cat([o, r[:, None], d.to(o.dtype)[:, None]], 1).cpu().numpy().copy())\n"
            "    st.append(env.get_state().cpu().numpy()); env.close()\n"
            "np.save(sys.argv[1], np.array(out)); np.save(sys.argv[1] + '.state.npy', np.array(st))\n" % (ROOT, dt, kw))
    outs = []
    for tag, extra in (("spec", {}), ("nospec", {"PLENVEC_LIB": lib0})):
        p = str(tmp_path / (tag + ".npy"))
        subprocess.run([sys.executable, "-c", code, p], check=True, timeout=600, env=dict(os.environ, **extra))
        outs.append((np.load(p), np.load(p + ".state.npy")))
    assert outs[0][0].shape == outs[1][0].shape and np.array_equal(outs[0][0], outs[1][0], equal_nan=True) and np.array_equal(outs[0][1], outs[1][1], equal_nan=True)
    assert (outs[0][0][:40, :, 27] != 0).sum() > 100                 # episodes ended (and, with the auto-reset on, restarted) inside the window
    if cfg == "fallen_robots":                                        # ... and without it the robots are down: torso below the termination height at the end
        assert (outs[0][1][0][:, 2] < 0.08).mean() > 0.5


@pytest.mark.parametrize("nenv", [2048, 1500, 3000])       # full slots, ragged odd last slot, ragged even last slot
def test_simd_load_balancing_does_not_change_results(tmp_path, nenv):
    """plen_balance_kernel only decides WHICH block (hence SIMD) runs an env; 2048 envs x 25 steps must be bit-identical with
    the identity placement (PLENVEC_NO_BALANCE=1), including across auto-resets, and the placement must be a permutation."""
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
            "from plen_ml_walk_amd.vec_env import PlenVecEnv\n"
            "g = torch.Generator().manual_seed(2); acts = (torch.rand(25, %d, 18, generator=g) * 2 - 1).float().cuda()\n"
            "env = PlenVecEnv(%d); env.reset(); out = []\n"
            "for t in range(25):\n"
            "    o, r, d, _ = env.step(acts[t]); out.append(torch.cat([o, r[:, None], d.float()[:, None]], 1).cpu().numpy().copy())\n"
            "np.save(sys.argv[1], np.array(out)); np.save(sys.argv[1] + '.state.npy', env.get_state().cpu().numpy())\n" % (ROOT, nenv, nenv))
    outs = []
    for tag, extra in (("bal", {}), ("ident", {"PLENVEC_NO_BALANCE": "1"})):
        p = str(tmp_path / (tag + ".npy"))
        subprocess.run([sys.executable, "-c", code, p], check=True, timeout=300, env=dict(os.environ, **extra))
        outs.append((np.load(p), np.load(p + ".state.npy")))
    assert np.array_equal(outs[0][0], outs[1][0], equal_nan=True) and np.array_equal(outs[0][1], outs[1][1], equal_nan=True)
    assert outs[0][0][..., 27].sum() > 0        # episodes ended (and restarted) inside the window


def test_determinism_and_independence():
    env = _env(64, torch.float32)
    g = torch.Generator().manual_seed(2)
    a = (torch.rand(6, 1, 18, generator=g) * 2 - 1).repeat(1, 64, 1).contiguous()
    env.reset()
    for t in range(6):
        o, r, d, _ = env.step(a[t].cuda())
    assert torch.all(o == o[0:1]) and torch.all(r == r[0]) and torch.all(d == d[0])     # same inputs -> same bits in every env
    env.close()


def test_domain_randomisation_matches_oracle():
    n = 4
    ms = torch.tensor([0.8, 1.0, 1.2, 1.1], dtype=torch.float64)
    mu = torch.tensor([0.4, 0.64, 1.0, 0.5], dtype=torch.float64)
    env = _env(n, torch.float64, cfg_overrides={"rolling_friction": 0.0})
    env.set_params(ms, mu)
    obs = env.reset().cpu().numpy()
    a = torch.full((n, 18), 0.1)
    nobs, rew, _, _ = env.step(a.cuda())
    for i in range(n):
        o = OracleEnv(); o.set_friction(rolling=0.0); o.set_params(float(ms[i]), float(mu[i]))
        r0 = o.reset()
        assert np.abs(r0 - obs[i]).max() <= 1e-9
        ob, r, d, _ = o.step(np.full(18, 0.1, dtype=np.float32).astype(np.float64))
        assert np.abs(ob - nobs[i].cpu().numpy()).max() <= 1e-7 and abs(r - float(rew[i])) <= 1e-7
    env.close()


def test_full_size_batch_properties():
    """BASELINE.json's 4096 envs: size-independent invariants over 60 random-action steps."""
    n = 4096
    env = _env(n, torch.float32)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    ended = torch.zeros(n, dtype=torch.bool, device="cuda")
    for t in range(60):
        a = torch.rand(n, 18, generator=g, device="cuda") * 2 - 1
        o, r, d, info = env.step(a)
        assert torch.isfinite(o).all() and torch.isfinite(r).all()
        assert ((o[:, 24] == 0) | (o[:, 24] == 1)).all() and ((o[:, 25] == 0) | (o[:, 25] == 1)).all()
        assert (d <= 3).all()
        ended |= d != 0
        term = (d & 1) != 0
        assert (r[term] < -50).all()                      # dead penalty present exactly on terminal steps
        assert torch.equal(info["obs"][~(d != 0)], o[~(d != 0)])
    s = env.get_state()
    assert torch.allclose(s[:, 3:7].norm(dim=1), torch.ones(n, device="cuda"), atol=1e-5)
    assert (s[:, 13:31].abs() < 1.75).all()               # joint limits +-1.7 hold (small violation allowed while a limit row acts)
    assert ended.float().mean() > 0.9                      # random flailing ends episodes within 60 steps
    env.close()


def test_facade_and_reference_shaped_driver(tmp_path):
    """PlenWalkEnv mirrors the reference's return types; the reference-shaped TD3 loop runs end to end."""
    from plen_ml_walk_amd import plen_env as pe, plen_td3
    from plen_ml_walk_amd import gym_compat as gym
    env = gym.make("PlenWalkEnv-v1", render=False)
    assert env._max_episode_steps == 500
    obs = env.reset()
    assert isinstance(obs, np.ndarray) and obs.dtype == np.float64 and obs.shape == (26,)
    assert np.abs(obs - OracleEnv().reset()).max() <= 1e-12
    o2, r, d, info = env.step(env.action_space.sample())
    assert o2.shape == (26,) and isinstance(r, np.float64) and isinstance(d, bool) and info == {}
    env.close()
    ev = plen_td3.main(max_timesteps=60, start_timesteps=30, eval_freq=50, out_dir=str(tmp_path / "run"), quiet=True)
    assert os.path.exists(str(tmp_path / "results" / "plen_walk_gazebo_.npy"))
    assert os.path.exists(str(tmp_path / "models" / "plen_walk_gazebo_49_actor"))


def test_v0_facade_and_train_cli_resume(tmp_path):
    """PlenWalkEnv-v0 (the Gazebo environment's contract on this simulator) through the gym-style facade, and the train_vec command line
    stopping and continuing a run (checkpoint + counters + replay)."""
    from plen_ml_walk_amd import plen_env as pe   # noqa: F401  (registers both ids)
    from plen_ml_walk_amd import gym_compat as gym
    from plen_ml_walk_amd import train_vec
    env = gym.make("PlenWalkEnv-v0")
    o = OracleEnv(reward_head=1)
    assert np.abs(env.reset() - o.reset()).max() <= 1e-12
    rng = np.random.default_rng(0)
    a = (0.3 * rng.uniform(-1, 1, 18)).astype(np.float32)          # one step: in the reference configuration the solver amplifies the
    ob, r, d, _ = env.step(a)                                      # 1e-15 differences ~100x per step
    ob2, r2, d2, _ = o.step(a.astype(np.float64))
    assert np.abs(ob - ob2).max() <= 1e-8 and abs(r - r2) <= 1e-8 and d == d2
    assert isinstance(r, np.float64) and ob.shape == (26,) and abs(r - 0.2) < 5.0       # alive bonus 100/500 plus small shaping terms
    env.close()
    prefix, bp = str(tmp_path / "ck"), str(tmp_path / "replay")
    train_vec.main(["--envs", "64", "--steps", "6", "--warmup", "2", "--batch", "64", "--start-timesteps", "128", "--replay", "4096", "--graphs", "1",
                    "--save", prefix, "--save-replay", "1", "--buffer-path", bp])
    assert all(os.path.exists(prefix + sfx) for sfx in ("_actor", "_critic", "_actor_optimizer", "_critic_optimizer", "_actor_target", "_critic_target", "_trainer.json"))
    c1 = json.load(open(prefix + "_trainer.json"))
    train_vec.main(["--envs", "64", "--steps", "4", "--warmup", "0", "--batch", "64", "--start-timesteps", "128", "--replay", "4096", "--graphs", "1",
                    "--resume", prefix, "--load-replay", "1", "--buffer-path", bp, "--save", prefix + "2"])
    c2 = json.load(open(prefix + "2_trainer.json"))
    assert c2["env_steps"] > c1["env_steps"] and c2["grad_steps"] > c1["grad_steps"]


def test_vector_td3_training_step(golden_dir):
    from plen_ml_walk_amd.td3 import ReplayBuffer, TD3Agent
    from plen_ml_walk_amd.train_vec import VecTD3Trainer
    env = _env(256, torch.float32)
    agent = TD3Agent(26, 18, 1.0)
    assert agent.device.type == "cuda"
    g = np.load(os.path.join(golden_dir, "td3_forward.npz"))
    agent.load_arrays(np.load(os.path.join(golden_dir, "policy_3229999.npz")))
    act = agent.select_action_batch(torch.as_tensor(g["obs"], dtype=torch.float32).cuda()).cpu().numpy()
    assert np.abs(act - g["action"]).max() <= 1e-4
    replay = ReplayBuffer(20000)
    tr = VecTD3Trainer(env, agent, replay, start_timesteps=512, batch_size=256, updates_per_step=2, seed=0)
    for _ in range(6):
        tr.step()
    assert replay.size == 6 * 256 and tr.grad_steps == 2 * 5 and torch.isfinite(agent.last_critic_loss)
    s, a, s2, r, nd = replay.sample(64)
    assert s.is_cuda and s.shape == (64, 26) and nd.min() >= 0 and nd.max() <= 1
    env.close()


def test_shipped_policy_statistics(golden_dir):
    """the reference's trained actor (checkpoint 3229999, trained in PyBullet) rolled through
    this environment.  No per-step parity is possible (PyBullet absent, chaotic contact dynamics), but the
    policy must transfer statistically: with a little action noise its episodes last several times longer
    and score far higher than an untrained actor's, and its mean return is in the range the reference's own
    training log ends at (last 1000 episodes: +50, results/plen_walk_gazebo_.npy)."""
    from plen_ml_walk_amd.walk_eval import load_policy, evaluate
    from plen_ml_walk_amd.td3 import TD3Agent
    pol = load_policy(os.path.join(golden_dir, "policy_3229999.npz"))
    res = evaluate(pol, 256, 1, torch.float32, action_noise=0.01, seed=0)
    r, l = np.array(res["returns"]), np.array(res["lengths"])
    assert len(r) == 256 and np.isfinite(r).all()
    torch.manual_seed(0)
    rnd = evaluate(TD3Agent(26, 18, 1.0, data_parallel=False), 256, 1, torch.float32, action_noise=0.01, seed=0)
    r0, l0 = np.array(rnd["returns"]), np.array(rnd["lengths"])
    assert l.mean() >= 100 and l.mean() >= 2.5 * l0.mean()
    assert r.mean() >= -60 and r.mean() >= r0.mean() + 150
    assert (l >= 500).mean() >= 0.03          # some episodes walk the whole 500 steps


def test_graphed_trainer_matches_ring_semantics():
    """hipGraph-captured collect/update loop: the replay ring fills exactly like the eager ReplayBuffer
    (write position, size, done_bool masking) and the TD3 update produces finite losses."""
    from plen_ml_walk_amd.td3 import ReplayBuffer, TD3Agent
    from plen_ml_walk_amd.train_vec import GraphedVecTD3Trainer
    n = 128
    env = _env(n, torch.float32)
    agent = TD3Agent(26, 18, 1.0)
    replay = ReplayBuffer(1000)                   # small ring: wraps during the test
    tr = GraphedVecTD3Trainer(env, agent, replay, start_timesteps=256, batch_size=128, updates_per_step=1, seed=0)
    for _ in range(12):
        tr.step()
    torch.cuda.synchronize()
    # every graph is warmed up by two eager executions before capture; those are real steps and are counted
    assert tr.env_steps >= 12 * n and tr.env_steps % n == 0 and int(tr.total_t) == tr.env_steps == tr.host_total
    assert replay.size == 1000 and replay.ptr == tr.host_total % 1000
    assert tr.grad_steps >= 9 and torch.isfinite(agent.last_critic_loss)
    assert torch.isfinite(replay.state).all() and torch.isfinite(replay.reward).all()
    nd = replay.not_done[:replay.size]
    assert ((nd == 0) | (nd == 1)).all()
    # terminal transitions carry the dead penalty, non-terminal ones do not
    assert (replay.reward[:replay.size][nd == 0] < -50).all()
    env.close()


def test_training_stop_and_resume(tmp_path):
    """checkpoint (4-file layout + counters) and replay buffer written by one run are picked up by the next
    (plen_td3.py:57-69): parameters, optimiser state, ring contents/position and update cadence continue."""
    from plen_ml_walk_amd.td3 import ReplayBuffer, TD3Agent
    from plen_ml_walk_amd.train_vec import GraphedVecTD3Trainer, VecTD3Trainer, resume_training, save_training
    n = 64
    env = _env(n, torch.float32)
    agent = TD3Agent(26, 18, 1.0)
    replay = ReplayBuffer(2000); replay.buffer_path = str(tmp_path / "replay")
    tr = VecTD3Trainer(env, agent, replay, start_timesteps=128, batch_size=64, updates_per_step=1, seed=0)
    for _ in range(8):
        tr.step()
    prefix = str(tmp_path / "ckpt")
    save_training(agent, tr, prefix); replay.save(7)
    env.close()
    # second run, graph trainer this time
    env2 = _env(n, torch.float32)
    agent2 = TD3Agent(26, 18, 1.0)
    replay2 = ReplayBuffer(2000); replay2.buffer_path = replay.buffer_path
    c = resume_training(agent2, replay2, prefix, 7)
    assert c["env_steps"] == tr.env_steps and c["grad_steps"] == tr.grad_steps and agent2.total_it == agent.total_it
    for a, b in zip(agent.critic.parameters(), agent2.critic.parameters()):
        assert torch.equal(a, b)
    for a, b in zip(agent.actor_target.parameters(), agent2.actor_target.parameters()):
        assert torch.equal(a, b)
    assert replay2.size == replay.size == 8 * n and torch.equal(replay2.state[:replay.size], replay.state[:replay.size])
    assert torch.equal(replay2.not_done[:replay.size], replay.not_done[:replay.size])
    tr2 = GraphedVecTD3Trainer(env2, agent2, replay2, start_timesteps=128, batch_size=64, updates_per_step=1, seed=1)
    tr2.restore_counters(c)
    assert tr2.host_total == replay.size and int(tr2.total_t) == replay.size
    for _ in range(3):
        tr2.step()
    torch.cuda.synchronize()
    assert tr2.env_steps > tr.env_steps and tr2.grad_steps > tr.grad_steps and replay2.size > replay.size
    # the earlier transitions are still there: the ring continued behind them
    assert torch.equal(replay2.state[:replay.size], replay.state[:replay.size])
    assert torch.isfinite(agent2.last_critic_loss)
    env2.close()


def test_pipelined_sub_batches_equal_single_batch():
    """PlenVecEnvPipelined (independent sub-batches on their own streams, bench.py's mode) is the same environments: 512 envs in
    two and four groups reproduce the single-launch run bit for bit, through auto-resets, with step() and with step_async()/sync()."""
    from plen_ml_walk_amd.vec_env import PlenVecEnvPipelined
    n, T = 512, 30
    g = torch.Generator().manual_seed(4)
    acts = (torch.rand(T, n, 18, generator=g) * 2 - 1).float().cuda()
    ref = _env(n, torch.float32); ref.reset()
    want = []
    for t in range(T):
        o, r, d, info = ref.step(acts[t]); want.append((o.clone(), r.clone(), d.clone(), info["obs"].clone()))
    ref.close()
    for groups, use_async in ((2, False), (4, True)):
        env = PlenVecEnvPipelined(n, groups=groups); env.reset()
        for t in range(T):
            if use_async:
                env.step_async(acts[t]); env.sync(); o, r, d, info = env.outputs()
            else:
                o, r, d, info = env.step(acts[t])
            assert torch.equal(o, want[t][0]) and torch.equal(r, want[t][1]) and torch.equal(d, want[t][2]) and torch.equal(info["obs"], want[t][3]), (groups, t)
        env.close()
    assert sum(int(w[2].sum()) for w in want) > 0


def test_finetuned_walking_policy_still_walks(golden_dir):
    """plen_ml_walk_amd/model/walk_finetuned_actor.npz = the reference's shipped actor after 10 k TD3 updates in THIS simulator
    (scripts/gpu_train_demo.py, 17 s on one MI355X; +311 mean return when saved).  A regression signal for the physics as a whole: a
    kernel change that alters the dynamics noticeably shows up here as a policy that no longer walks."""
    from plen_ml_walk_amd.walk_eval import load_policy, evaluate
    pol = load_policy(os.path.join(ROOT, "plen_ml_walk_amd", "model", "walk_finetuned_actor.npz"))
    base = load_policy(os.path.join(golden_dir, "policy_3229999.npz"))
    res = evaluate(pol, 256, 1, torch.float32, action_noise=0.01, seed=3)
    ref = evaluate(base, 256, 1, torch.float32, action_noise=0.01, seed=3)
    r, l = np.array(res["returns"]), np.array(res["lengths"])
    r0 = np.array(ref["returns"])
    assert r.mean() >= 150 and r.mean() >= r0.mean() + 100 and l.mean() >= 250 and (l >= 500).mean() >= 0.2


@pytest.mark.parametrize("nit", [3, 50])
def test_joint_limit_rows_vs_oracle_f64(nit):
    """States with joints beyond the +-1.7 rad limits (plen.urdf:1310), moving further out, in every contact situation the collected states offer: the limit
    rows (btMultiBodyJointLimitConstraint, only while violated) against the oracle after one substep.  Round 4: the solver loop exists once per pair of
    foot point counts and a substep with a violated limit runs the (4, 4) copy whatever its contact set (rows of empty slots are exact no-ops) -- this is
    the test of that routing."""
    n = 48
    S, T = collect_states(n, seed=9, with_targets=True)
    rng = np.random.default_rng(2)
    for i in range(n):
        for j in rng.choice(18, size=1 + i % 3, replace=False):
            sgn = rng.choice([-1.0, 1.0])
            S[i, 13 + j] = sgn * (1.7 + rng.uniform(0.002, 0.03))          # beyond the limit (inside the -0.04 split-impulse threshold)
            S[i, 31 + j] = sgn * rng.uniform(0.0, 2.0)                       # and still moving outwards
    env = _env(n, torch.float64, cfg_overrides={"num_iterations": nit, "rolling_friction": 0.0})
    env.set_state(torch.tensor(S))
    env.debug_substeps(torch.tensor(T), nsub=1)
    out = env.get_state().cpu().numpy()
    worst = 0.0
    for i in range(n):
        o = OracleEnv(); o.set_world(num_iterations=nit); o.set_friction(rolling=0.0); o.set_state(S[i]); o.set_targets(T[i]); o.substep()
        worst = max(worst, np.abs(out[i] - o.get_state()).max())
    assert worst <= (1e-9 if nit == 3 else 1e-7), worst
    env.close()


def _quat(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p / 2), np.sin(p / 2), np.cos(y / 2), np.sin(y / 2)
    return np.array([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy])


def constructed_foot_states(per_pair=2, seed=0, pushes=(0.0, 0.0003, 0.001, 0.002, 0.004)):
    """States in which the right / left foot hold every pair of contact-point counts (0..4)^2, built, not found: a slightly tilted torso, each leg's joints perturbed
    on a random scale (a flat foot: 4 points, a foot on an edge: 2, on a corner: 1, a slightly tilted flat foot: 3, a lifted one: 0), the torso lowered by bisection
    until the first point touches and then a little further.  Classified by the oracle's own collision pass; states in which any other link touches are skipped."""
    rng = np.random.default_rng(seed)
    o = OracleEnv()
    found = {}

    def counts(s):
        o.set_state(s)
        box, _ = o.contact_slots(run_collide=True)
        return int((box[:4] != -2).sum()), int((box[4:] != -2).sum()), box
    tries = 0
    while (len(found) < 24 or min(len(v) for v in found.values()) < per_pair) and tries < 400000:
        tries += 1
        s = np.zeros(49)
        s[3:7] = _quat(rng.normal(0, 0.03), rng.normal(0, 0.03), 0.0)
        for leg in range(2):
            s[13 + 6 * leg:19 + 6 * leg] = rng.normal(0, rng.choice([0.0, 0.01, 0.03, 0.1, 0.4]), 6)
        lo, hi = 0.10, 0.20
        for _ in range(18):
            s[2] = 0.5 * (lo + hi)
            nr, nl, _ = counts(s)
            lo, hi = (s[2], hi) if nr + nl > 0 else (lo, s[2])
        s[2] = lo - rng.choice(list(pushes))
        nr, nl, box = counts(s)
        if (box >= 0).any() or nr + nl == 0 or len(found.get((nr, nl), [])) >= per_pair:
            continue
        found.setdefault((nr, nl), []).append(s.copy())
    air = np.zeros(49); air[2] = 0.25; air[6] = 1.0
    found[(0, 0)] = [air.copy() for _ in range(per_pair)]
    return found


@pytest.mark.parametrize("nit,rolling,tol", [(3, None, 1e-9), (50, 0.0, 1e-7)])
def test_every_copy_of_the_solver_loop_vs_oracle_f64(nit, rolling, tol):
    """The solver's iteration loop exists 50 times (one copy per pair of foot point counts (NR, NL) in {0..4}^2, each with and without the joint-limit rows) and a
    substep runs exactly one of them.  Here every copy runs: constructed states for all 25 pairs (constructed_foot_states), each with moving joints and a moving
    torso, once as it is and once with one joint beyond its +-1.7 rad limit and moving outwards (plen.urdf:1310; the limit rows exist only while violated) -- one
    substep of the f64 kernel against the oracle: 3 iterations in the reference configuration to 1e-9, 50 iterations with rolling friction off to 1e-7
    (with it on the iteration expands rounding differences).  The copy that served each state is read back from the kernel's debug dump:
    the set of copies visited must be ALL 50."""
    found = constructed_foot_states(per_pair=2, seed=3)
    assert len(found) == 25, sorted(found)
    rng = np.random.default_rng(7)
    S, T, want = [], [], []
    for (nr, nl), states in sorted(found.items()):
        for k, s0 in enumerate(states):
            for lim in (0, 1):
                s = s0.copy()
                s[7:10] = rng.normal(0, 0.3, 3); s[10:13] = rng.normal(0, 0.05, 3); s[12] -= 0.05          # turning, drifting, coming down
                s[31:49] = rng.normal(0, 1.0, 18)
                if lim:
                    j = 12 + rng.integers(0, 6)                    # an arm joint: the feet stay where they were put
                    sgn = rng.choice([-1.0, 1.0])
                    s[13 + j] = sgn * (1.7 + rng.uniform(0.002, 0.03)); s[31 + j] = sgn * rng.uniform(0.2, 2.0)
                S.append(s); T.append(rng.uniform(-0.5, 0.5, 18)); want.append(5 * nr + nl + 100 * lim)
    S, T = np.array(S), np.array(T)
    n = len(S)
    over = {"num_iterations": nit}
    if rolling is not None:
        over["rolling_friction"] = rolling
    env = _env(n, torch.float64, cfg_overrides=over)
    env.set_state(torch.tensor(S))
    dump = env.debug_substeps(torch.tensor(T), nsub=1, dump=True).cpu().numpy()
    out = env.get_state().cpu().numpy()
    env.close()
    served = dump[:, 3701].astype(int)
    # the oracle's own classification of each state as it stands (arm joints moved: a hand may have come near the ground -- then the state is not counted)
    worst, visited = 0.0, set()
    for i in range(n):
        o = OracleEnv(); o.set_world(num_iterations=nit)
        if rolling is not None:
            o.set_friction(rolling=rolling)
        o.set_state(S[i]); o.set_targets(T[i])
        box, _ = o.contact_slots(run_collide=True)
        o.substep()
        err = np.abs(out[i] - o.get_state()).max()
        assert err <= tol, (i, want[i], served[i], err)
        worst = max(worst, err)
        if not (box >= 0).any():
            assert served[i] == want[i], (i, served[i], want[i])
            visited.add(int(served[i]))
    assert visited == {5 * a + b + 100 * l for a in range(5) for b in range(5) for l in (0, 1)}, sorted({5 * a + b + 100 * l for a in range(5) for b in range(5) for l in (0, 1)} - visited)


def _every_copy_states(seed_states, seed_motion, pushes=(0.0, 0.0003, 0.001, 0.002, 0.004)):
    """The constructed states of test_every_copy_of_the_solver_loop_vs_oracle_f64 (all 25 pairs of foot point counts, each with and without a violated joint limit)
    with motion and motor targets: (states, targets, wanted copy)."""
    found = constructed_foot_states(per_pair=2, seed=seed_states, pushes=pushes)
    assert len(found) == 25, sorted(found)
    rng = np.random.default_rng(seed_motion)
    S, T, want = [], [], []
    for (nr, nl), states in sorted(found.items()):
        for s0 in states:
            for lim in (0, 1):
                s = s0.copy()
                s[7:10] = rng.normal(0, 0.3, 3); s[10:13] = rng.normal(0, 0.05, 3); s[12] -= 0.05
                s[31:49] = rng.normal(0, 1.0, 18)
                if lim:
                    j = 12 + rng.integers(0, 6)
                    sgn = rng.choice([-1.0, 1.0])
                    s[13 + j] = sgn * (1.7 + rng.uniform(0.002, 0.03)); s[31 + j] = sgn * rng.uniform(0.2, 2.0)
                S.append(s); T.append(rng.uniform(-0.5, 0.5, 18)); want.append(5 * nr + nl + 100 * lim)
    return np.array(S), np.array(T), np.array(want)


def test_every_copy_of_the_solver_loop_vs_oracle_f32():
    """VERDICT r05 weak point 7: the f32 instantiation of the 50 loop copies (its own code: DPP operands in the cone pair, pipelined v_writelane motor rows) was held
    only through whatever a random rollout visits.  The constructed states of the f64 test -- rounded to f32, pushed at least 0.3 mm past first contact so that the
    contact sets do not hang on the last bit -- through ONE substep of the f32 kernel, 3 solver iterations, against the f64 oracle started from the same rounded state:
    <= 1e-3 relative to max(1, |reference|) in every state entry, median over the states <= 1e-4.  States whose contact classification changes under a 2e-6 m shift of the torso are left out (f32 height
    arithmetic against f64 thresholds); the copies visited by the rest must still be (nearly) all 50, and the copy the kernel ran must be the one the oracle's collision
    pass names."""
    S, T, _ = _every_copy_states(11, 13, pushes=(0.0003, 0.001, 0.002, 0.004))
    S = S.astype(np.float32).astype(np.float64); T = T.astype(np.float32).astype(np.float64)
    n = len(S)
    env = _env(n, torch.float32, cfg_overrides={"num_iterations": 3})
    env.set_state(torch.tensor(S, dtype=torch.float32))
    dump = env.debug_substeps(torch.tensor(T, dtype=torch.float32), nsub=1, dump=True).cpu().numpy()
    out = env.get_state().cpu().numpy().astype(np.float64)
    env.close()
    served = dump[:, 3701].astype(int)
    errs, visited, used = [], set(), 0
    o = OracleEnv(); o.set_world(num_iterations=3)

    def classify(s):
        o.set_state(s)
        box, _ = o.contact_slots(run_collide=True)
        return int((box[:4] != -2).sum()), int((box[4:] != -2).sum()), bool((box >= 0).any())
    for i in range(n):
        cls = classify(S[i])
        robust = not cls[2]
        for dz in (-2e-6, 2e-6):
            s2 = S[i].copy(); s2[2] += dz
            robust = robust and classify(s2) == cls
        if not robust:
            continue
        lim = int((np.abs(S[i, 13:31]) >= 1.7).any())
        o.set_state(S[i]); o.set_targets(T[i]); o.substep()
        ref = o.get_state()
        err = (np.abs(out[i] - ref) / np.maximum(1.0, np.abs(ref))).max()
        assert served[i] == 5 * cls[0] + cls[1] + 100 * lim, (i, served[i], cls, lim)
        assert err <= 1e-3, (i, served[i], err)
        errs.append(err); visited.add(int(served[i])); used += 1
    print("f32 loop copies vs the f64 oracle, 3 iterations: %d states, %d copies, relative error median %.2e, max %.2e" % (used, len(visited), np.median(errs), max(errs)))
    assert np.median(errs) <= 1e-4          # (measured r06: max 2.1e-4 -- f32 rounding of a 3-iteration solve from a contact-rich state; the f64 copies are held to 1e-9)
    assert used >= 80 and len(visited) >= 46, (used, sorted(visited))
    assert {0, 1, 5, 6, 2, 10, 12, 24, 100, 101, 105, 124} <= visited          # the common copies (airborne, one and two points per foot, flat feet) are all in


@pytest.mark.parametrize("dt", ["float32", "float64"])
def test_every_copy_bitwise_equals_the_run_time_tested_loop(tmp_path, dt):
    """... and the same constructed states (every pair of point counts, with and without a violated limit: all 50 copies, not just what a rollout visits) through
    the shipped library and its A/B twin (run-time point tests, csrc/variants/nospec.so): one full substep at 50 iterations, states bit for bit equal."""
    from plen_ml_walk_amd.build import build_variant, REFERENCE_FORM_FLAGS
    lib0 = build_variant("nospec", REFERENCE_FORM_FLAGS)
    S, T, want = _every_copy_states(3, 7)
    np.save(str(tmp_path / "S.npy"), S); np.save(str(tmp_path / "T.npy"), T)
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
            "from plen_ml_walk_amd.vec_env import PlenVecEnv\n"
            "dt = torch.%s\n"
            "S = torch.tensor(np.load(sys.argv[1] + '/S.npy'), dtype=dt); T = torch.tensor(np.load(sys.argv[1] + '/T.npy'), dtype=dt)\n"
            "env = PlenVecEnv(S.shape[0], dtype=dt); env.set_state(S.cuda())\n"
            "dump = env.debug_substeps(T.cuda(), nsub=1, dump=True).cpu().numpy()\n"
            "np.save(sys.argv[2], env.get_state().cpu().numpy()); np.save(sys.argv[2] + '.served.npy', dump[:, 3701])\n" % (ROOT, dt))
    outs = []
    for tag, extra in (("spec", {}), ("nospec", {"PLENVEC_LIB": lib0})):
        q = str(tmp_path / (tag + ".npy"))
        subprocess.run([sys.executable, "-c", code, str(tmp_path), q], check=True, timeout=600, env=dict(os.environ, **extra))
        outs.append((np.load(q), np.load(q + ".served.npy")))
    assert np.array_equal(outs[0][0], outs[1][0], equal_nan=True)
    assert len(set(outs[0][1].astype(int))) >= (50 if dt == "float64" else 44)          # (f32 rounds the constructed heights: a few point counts may merge)
