"""GPU tests of the ground contact of the non-foot links (plen.urdf:504-1274 box colliders, plane loaded at
plen_env.py:306-309): the HIP kernel's lendable contact slots against the oracle on fall / kneel / prop states."""
import numpy as np
import pytest
import torch

import np_model as nm
from oracle.oracle import OracleEnv, agent_to_env

pytestmark = pytest.mark.gpu


def _env(n, dtype, **kw):
    from plen_ml_walk_amd.vec_env import PlenVecEnv
    return PlenVecEnv(n, dtype=dtype, **kw)


def fall_states(n, seed=21, rolling=0.0, stride=3):
    """States along oracle rollouts that are NOT cut off at the termination height: the robot falls, thrashes and lies on the floor."""
    rng = np.random.default_rng(seed)
    S, T, owners = [], [], []
    while len(S) < n:
        e = OracleEnv()
        if rolling is not None:
            e.set_friction(rolling=rolling)
        e.reset()
        tgt = np.zeros(18)
        for t in range(500):
            if t % 4 == 0:
                a = rng.uniform(-1, 1, 18)
                tgt = np.array([agent_to_env(j, a[j]) for j in range(18)])
            e.set_targets(tgt); e.substep()
            if t > 40 and t % stride == 0 and len(S) < n:
                s = e.get_state()
                own, _ = e.contact_slots(run_collide=True)
                if (own >= 0).any():                       # keep the states in which a box corner holds a slot
                    S.append(s); T.append(tgt.copy()); owners.append(own)
    return np.array(S), np.array(T), np.array(owners)


def _lowest_box_corner(state):
    R, O, _, _ = nm.fk(state[0:3], state[3:7], state[13:31])
    z = []
    for x in nm.BOXES:
        b = x["body"]; h = np.array(x["half"])
        for cn in range(8):
            sg = np.array([h[0] if cn & 1 else -h[0], h[1] if cn & 2 else -h[1], h[2] if cn & 4 else -h[2]])
            z.append((O[b] + R[b] @ (np.array(x["t_body"]) + np.array(x["R_body"]) @ sg))[2])
    return min(z)


@pytest.mark.parametrize("nit,tol", [(3, 1e-9), (50, None)])
def test_box_contact_substeps_match_oracle_f64(nit, tol):
    """One substep from 96 fall states with box corners in contact: slot assignment (which slots are occupied, which are lent) is identical,
    the state after the substep agrees to 1e-9 with 3 solver iterations (logic) and in the median with 50 (thrashing robots are ill-conditioned)."""
    n = 96
    S, T, owners = fall_states(n)
    env = _env(n, torch.float64, cfg_overrides=dict(rolling_friction=0.0, num_iterations=nit))
    env.set_state(torch.as_tensor(S))
    env.debug_substeps(torch.as_tensor(T), 1)
    got = env.get_state().cpu().numpy()
    aux = env.get_aux().cpu().numpy()
    errs = []
    for i in range(n):
        o = OracleEnv(); o.set_friction(rolling=0.0); o.set_world(num_iterations=nit)
        o.set_state(S[i]); o.set_targets(T[i]); o.substep()
        own, _ = o.contact_slots()
        assert (own == owners[i]).all()
        lent = sum(1 << c for c in range(8) if own[c] >= 0)
        occ = sum(1 << c for c in range(8) if own[c] != -2)
        assert aux[i, 7] == (lent | (occ << 8)), (i, own, hex(aux[i, 7]))
        c = o.contacts()
        assert aux[i, 4] == c["right"] and aux[i, 5] == c["left"] and aux[i, 6] == c["iterations"]
        errs.append(np.abs(got[i] - o.get_state()).max())
    errs = np.array(errs)
    assert (owners >= 0).sum() >= n and ((owners >= 0).any(1) & (owners == -1).any(1)).sum() >= 10     # lent slots, also next to foot points
    if tol is not None:
        assert errs.max() <= tol
    else:
        assert np.median(errs) <= 1e-11 and np.mean(errs <= 1e-6) >= 0.9
    env.close()


def test_box_contact_substeps_f32():
    """The f32 kernel on the same states: same slot assignment except where a corner sits within rounding of its threshold, state within f32
    accuracy of the oracle in the median (3 iterations: no solver amplification)."""
    n = 96
    S, T, owners = fall_states(n)
    env = _env(n, torch.float32, cfg_overrides=dict(rolling_friction=0.0, num_iterations=3))
    env.set_state(torch.as_tensor(S))
    env.debug_substeps(torch.as_tensor(T), 1)
    got = env.get_state().cpu().numpy().astype(np.float64)
    aux = env.get_aux().cpu().numpy()
    errs, same = [], 0
    for i in range(n):
        o = OracleEnv(); o.set_friction(rolling=0.0); o.set_world(num_iterations=3)
        o.set_state(S[i]); o.set_targets(T[i]); o.substep()
        own, _ = o.contact_slots()
        lent = sum(1 << c for c in range(8) if own[c] >= 0); occ = sum(1 << c for c in range(8) if own[c] != -2)
        same += int(aux[i, 7] == (lent | (occ << 8)))
        errs.append(np.abs(got[i] - o.get_state()).max() / max(1.0, np.abs(o.get_state()).max()))
    assert same >= n - 3 and np.median(errs) <= 2e-5 and np.mean(np.array(errs) <= 1e-3) >= 0.9
    env.close()


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_fallen_robot_rests_on_its_boxes(dtype):
    """64 envs thrash for 150 steps without being reset (auto_reset off), then relax for 2 s: every robot ends up lying or kneeling on
    the floor, held up by the box colliders of its links (no box corner more than a few mm under the ground), at rest, with finite state.
    With body_contacts = 0 (the round-1 model: feet only) the same run leaves links sunk deep into the floor."""
    n = 64
    res = {}
    for bc in (1, 0):
        env = _env(n, dtype, auto_reset=False, cfg_overrides=dict(body_contacts=bc))
        env.reset()
        g = torch.Generator(device="cuda").manual_seed(5)
        for t in range(150):
            env.step(torch.rand(n, 18, generator=g, device="cuda") * 2 - 1)
        for t in range(120):
            env.step(torch.zeros(n, 18, device="cuda"))
        s = env.get_state().cpu().numpy().astype(np.float64)
        assert np.isfinite(s).all() and env.nonfinite_count() == 0
        res[bc] = np.array([_lowest_box_corner(s[i]) for i in range(n)]), s
        env.close()
    low, s = res[1]
    assert (s[:, 2] < 0.13).mean() >= 0.9                      # they did fall
    assert low.min() >= -0.006 and np.median(low) >= -0.002
    assert np.median(np.abs(s[:, 7:13]).max(1)) < 0.3          # and came to rest
    assert np.median(res[0][0]) < -0.01                        # feet-only contact model: boxes inside the floor


def test_body_contacts_off_reproduces_the_feet_only_model():
    """cfg.body_contacts = 0 is the round-1 contact model; the oracle has the same switch: rollouts agree as before."""
    n, T = 16, 12
    rng = np.random.default_rng(3)
    acts = (0.3 * rng.uniform(-1, 1, (T, n, 18))).astype(np.float32)
    env = _env(n, torch.float64, cfg_overrides=dict(body_contacts=0, rolling_friction=0.0))
    env.reset()
    want = []
    for i in range(n):
        o = OracleEnv(); o.set_body_contacts(False); o.set_friction(rolling=0.0); o.reset()
        want.append(o.rollout(acts[:, i])[0])
    for t in range(T):
        ob, r, d, _ = env.step(torch.as_tensor(acts[t]).cuda())
        ref = np.array([want[i][t] for i in range(n)])
        assert np.abs(ob.cpu().numpy() - ref).max() <= 1e-6
    env.close()


# both feet flat on the ground (8 foot points) AND the right hand's box on the floor: found by a least-squares search over the joint angles,
# base pitch / roll and height (sole vertices and the hand's lowest corner 0.3 mm inside the ground); joints within +-1.6 rad
HAND_DOWN_STATE = [0.0, 0.0, 0.089671, 0.25096, 0.458035, -0.136556, 0.841769, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, -0.262156, 0.382294, -0.594681, 1.194079,
                   -0.552259, 0.236333, 0.403619, -0.007623, 0.459387, -1.1328, 0.286389, 0.643254, 0.863805, 0.32101, -0.210111] + [0.0] * 21
R_HAND_BOX = 24


def _hand_low(state):
    R, O, _, _ = nm.fk(state[0:3], state[3:7], state[13:31])
    x = nm.BOXES[R_HAND_BOX]; b = x["body"]; h = np.array(x["half"])
    return min((O[b] + R[b] @ (np.array(x["t_body"]) + np.array(x["R_body"]) @ np.array([h[0] if cn & 1 else -h[0], h[1] if cn & 2 else -h[1], h[2] if cn & 4 else -h[2]])))[2]
               for cn in range(8))


def test_both_feet_flat_and_a_hand_pressed_to_the_ground():
    """VERDICT r02 item 6 (the 8-slot cap): with both feet fully planted (8 foot points) a hand on the ground used to get NO contact.  Now each
    foot gives up the slot of its fourth point while another link is near the ground: the hand's corner takes slot 3, the left foot keeps its
    four points (nobody took slot 7), kernel = oracle on the slot masks and on the state (1e-9 with 3 iterations), and the hand CARRIES LOAD:
    pressed down by its shoulder motor for 0.2 s it stays on the floor, while with body_contacts = 0 it sinks centimetres into it."""
    s0 = np.array(HAND_DOWN_STATE)
    tgt = s0[13:31].copy(); tgt[12] += 0.5                                       # the shoulder pushes the arm down
    o = OracleEnv(); o.set_state(s0)
    own, _ = o.contact_slots(run_collide=True)
    assert own.tolist() == [-1, -1, -1, R_HAND_BOX, -1, -1, -1, -1]
    # one substep, 3 and 50 iterations, kernel vs oracle: slot masks and state
    for nit, tol in ((3, 1e-9), (50, 1e-6)):
        env = _env(2, torch.float64, auto_reset=False, cfg_overrides=dict(rolling_friction=0.0, num_iterations=nit))
        env.set_state(torch.as_tensor(np.tile(s0, (2, 1))))
        env.debug_substeps(torch.as_tensor(np.tile(tgt, (2, 1))), 1)
        got = env.get_state().cpu().numpy(); aux = env.get_aux().cpu().numpy()
        o = OracleEnv(); o.set_friction(rolling=0.0); o.set_world(num_iterations=nit)
        o.set_state(s0); o.set_targets(tgt); o.substep()
        own, _ = o.contact_slots()
        assert own.tolist() == [-1, -1, -1, R_HAND_BOX, -1, -1, -1, -1]
        assert aux[0, 7] == ((1 << 3) | (0xff << 8)), hex(aux[0, 7])
        assert aux[0, 4] == 1 and aux[0, 5] == 1                                 # contact flags are the feet's, unchanged
        assert np.abs(got[0] - o.get_state()).max() <= tol, (nit, np.abs(got[0] - o.get_state()).max())
        env.close()
    # 48 substeps of pressing: held up with the box contacts, sunk without
    low = {}
    for bc in (1, 0):
        env = _env(2, torch.float64, auto_reset=False, cfg_overrides=dict(rolling_friction=0.0, body_contacts=bc))
        env.set_state(torch.as_tensor(np.tile(s0, (2, 1))))
        env.debug_substeps(torch.as_tensor(np.tile(tgt, (2, 1))), 48)
        st = env.get_state().cpu().numpy()[0]
        assert np.isfinite(st).all()
        low[bc] = _hand_low(st)
        env.close()
    assert low[1] > -0.001 and low[0] < -0.02, low
    o = OracleEnv(); o.set_friction(rolling=0.0); o.set_state(s0); o.set_targets(tgt)
    for _ in range(48):
        o.substep()
    assert abs(_hand_low(o.get_state()) - low[1]) < 1e-3
