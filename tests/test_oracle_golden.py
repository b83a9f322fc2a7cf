"""Pin the CPU oracle's Python-level arithmetic against golden vectors captured from the REFERENCE
(tools/make_golden.py stub-imports plen_env.py).  CPU only."""
import os
import numpy as np
import pytest
from oracle.oracle import OracleEnv, agent_to_env


def test_a1_agent_to_env(golden_dir):
    g = np.load(os.path.join(golden_dir, "a1_agent_to_env.npz"))
    act, ref = g["action"], g["env_action"]
    got = np.array([[agent_to_env(j, act[i, j]) for j in range(18)] for i in range(act.shape[0])])
    assert np.max(np.abs(got - ref)) <= 1e-15          # f64: identical arithmetic
    # the clamps were exercised on both sides
    assert np.any(ref == g["env_ranges"][:, 1] - 0.001) and np.any(ref == g["env_ranges"][:, 0] + 0.001)


def test_a78_reward_done_single_states(golden_dir):
    g = np.load(os.path.join(golden_dir, "a78_reward_done.npz"))
    n = len(g["reward"])
    import ctypes as C
    from oracle.oracle import _lib
    lib = _lib()
    lib.oracle_script_inject.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
    lib.oracle_script_reward.restype = C.c_double
    lib.oracle_script_reward.argtypes = [C.c_void_p] + [C.c_double] * 6 + [C.c_int] * 2 + [C.c_double] * 4 + [C.POINTER(C.c_int)]
    e = OracleEnv()
    worst = 0.0
    branches = set()
    for i in range(n):
        nh = int(g["nh"][i])
        hist = np.ascontiguousarray(g["hist"][i][:, :nh])
        diffs = np.ascontiguousarray(g["diffs"][i])
        lib.oracle_script_inject(e.h, int(g["cnt"][i]), int(g["ds"][i]), nh,
                                 hist.ctypes.data_as(C.POINTER(C.c_double)), diffs.ctypes.data_as(C.POINTER(C.c_double)),
                                 int(g["first"][i]))
        done = C.c_int(0)
        r = lib.oracle_script_reward(e.h, g["z"][i], g["vx"][i], g["roll"][i], g["pitch"][i], g["yaw"][i], g["y"][i],
                                     int(g["rc"][i]), int(g["lc"][i]), g["lrp"][i][0], g["lrp"][i][1], g["rrp"][i][0], g["rrp"][i][1],
                                     C.byref(done))
        assert bool(done.value) == bool(g["done"][i])
        aux = e.get_aux()
        assert aux["gait_period_counter"] == g["cnt_after"][i]
        assert aux["double_support_counter"] == g["ds_after"][i]
        assert aux["nhist"] == g["nh_after"][i]
        ref = g["reward"][i]
        if np.isnan(ref):
            assert np.isnan(r)
        else:
            worst = max(worst, abs(r - ref) / max(1.0, abs(ref)))
        branches.add((bool(g["done"][i]), g["cnt"][i] >= 120, g["cnt"][i] >= 80 and g["rc"][i] == 1, g["vx"][i] < 0))
    assert worst <= 1e-12
    assert len(branches) >= 8          # both sides of the main conditionals were hit


@pytest.mark.parametrize("ep", [0, 1, 2, 3])
def test_a69_scripted_episode(golden_dir, ep):
    """600 scripted steps through the env-level state machine: obs assembly, gait histories,
    cosine-similarity / diff penalties, counters, termination.  Mirrors plen_env.py:638-692 order."""
    g = np.load(os.path.join(golden_dir, "a69_script.npz"))
    k = lambda name: g["%s_%d" % (name, ep)]
    q, pos, rpy, linvel, rc, lc, frp = k("q"), k("pos"), k("rpy"), k("linvel"), k("rc"), k("lc"), k("frp")
    e = OracleEnv()
    e.script_reset()
    worst = 0.0
    for t in range(q.shape[0]):
        r, done = e.script_step(q[t], pos[t, 2], linvel[t, 0], rpy[t, 0], rpy[t, 1], rpy[t, 2], pos[t, 1],
                                rc[t], lc[t], frp[t, 0], frp[t, 1], frp[t, 2], frp[t, 3])
        assert done == bool(k("done")[t]), t
        aux = e.get_aux()
        ref_c = k("counters")[t]
        assert [aux["gait_period_counter"], aux["double_support_counter"], aux["episode_timestep"], aux["nhist"], aux["first_pass"]] == list(ref_c), t
        ref = k("reward")[t]
        worst = max(worst, abs(r - ref) / max(1.0, abs(ref)))
        # the reference's obs is exactly what was scripted in (assembly order check)
        o = k("obs")[t]
        assert np.allclose(o[:18], q[t], atol=0) and o[18] == pos[t, 2] and o[19] == linvel[t, 0] and o[23] == pos[t, 1]
        assert abs(o[20] - rpy[t, 0]) < 1e-12 and abs(o[21] - rpy[t, 1]) < 1e-12 and abs(o[22] - rpy[t, 2]) < 1e-12
        assert o[24] == rc[t] and o[25] == lc[t]
    assert worst <= 1e-12


def test_a1_targets_in_script(golden_dir):
    """move_joints received agent_to_env(action) for every step (plen_env.py:650-663)."""
    g = np.load(os.path.join(golden_dir, "a69_script.npz"))
    acts, tg = g["actions_0"], g["targets_0"]
    got = np.array([[agent_to_env(j, acts[t, j]) for j in range(18)] for t in range(acts.shape[0])])
    assert np.max(np.abs(got - tg)) <= 1e-15


def test_gazebo_head_reward_done_contact(golden_dir):
    """the PlenWalkEnv-v0 contract (plen_walk.py:346-396, 597-650), oracle vs vectors captured
    from the reference itself (tools/make_golden_gazebo.py)."""
    g = np.load(os.path.join(golden_dir, "gazebo_reward_done.npz"))
    w = g["weights"]
    assert w.tolist() == [100.0, 0.2, 3.0, 0.158, 20.0, 1.0, 1.0, 0.5, 1.0, 500.0]     # constants restated in the oracle and the kernel
    env = OracleEnv(reward_head=1)
    st, ref = g["states"], g["done_dead_reward"]
    for i in range(st.shape[0]):
        r, done, dead = env.gazebo_script(*st[i, :7], int(st[i, 7]))
        assert done == bool(ref[i, 0]) and dead == bool(ref[i, 1]), i
        assert abs(r - ref[i, 2]) <= 1e-12 * max(1.0, abs(ref[i, 2])), (i, r, ref[i, 2])
    for f, fl in zip(g["forces"], g["flags"]):
        assert env.gazebo_contact(f) == int(fl[0]) == int(fl[1])
