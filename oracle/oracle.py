"""ctypes binding of the CPU oracle (oracle/plen_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (plen_ml_walk_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PLEN_ORACLE_LIB_DIR: load libplen_oracle_<dtype>.so from there (an oracle built on other model tables: scripts/pin/margin_pooled.py)
_LIB_DIR = os.environ.get("PLEN_ORACLE_LIB_DIR") or _HERE
_LIBS = {}


def build(force=False):
    """Compile the oracle shared objects with gcc (a few seconds)."""
    targets = ["libplen_oracle_f64.so", "libplen_oracle_f32.so"]
    if force or not all(os.path.exists(os.path.join(_HERE, t)) for t in targets):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + targets)


def _lib(dtype="f64"):
    if dtype not in _LIBS:
        path = os.path.join(_LIB_DIR, "libplen_oracle_%s.so" % dtype)
        if not os.path.exists(path):
            assert _LIB_DIR == _HERE, path
            build()
        lib = C.CDLL(path)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        lib.oracle_create.restype = C.c_void_p
        lib.oracle_create.argtypes = [C.c_int]
        lib.oracle_destroy.argtypes = [C.c_void_p]
        lib.oracle_copy.argtypes = [C.c_void_p, C.c_void_p]
        lib.oracle_set_params.argtypes = [C.c_void_p, C.c_double, C.c_double]
        lib.oracle_set_world.argtypes = [C.c_void_p, C.c_int, C.c_double]
        lib.oracle_set_friction.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        lib.oracle_set_hyp.argtypes = [C.c_void_p, C.c_int, C.c_double]
        lib.oracle_set_link_inertia.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
        lib.oracle_last_residual.restype = C.c_double
        lib.oracle_last_residual.argtypes = [C.c_void_p]
        lib.oracle_get_state.argtypes = [C.c_void_p, dp]
        lib.oracle_set_state.argtypes = [C.c_void_p, dp]
        lib.oracle_get_aux.argtypes = [C.c_void_p, ip]
        lib.oracle_set_targets.argtypes = [C.c_void_p, dp]
        lib.oracle_substep.argtypes = [C.c_void_p]
        lib.oracle_set_trace.argtypes = [dp, C.c_int]
        lib.oracle_contacts.argtypes = [C.c_void_p, ip]
        lib.oracle_contact_slots.argtypes = [C.c_void_p, C.c_int, ip, dp]
        lib.oracle_set_body_contacts.argtypes = [C.c_void_p, C.c_int]
        lib.oracle_forward_dynamics.argtypes = [C.c_void_p, dp]
        lib.oracle_minv_times.argtypes = [C.c_void_p, dp, dp]
        lib.oracle_link_frames.argtypes = [C.c_void_p, dp, dp, dp]
        lib.oracle_agent_to_env.restype = C.c_double
        lib.oracle_agent_to_env.argtypes = [C.c_int, C.c_double]
        lib.oracle_script_reset.argtypes = [C.c_void_p]
        lib.oracle_script_step.restype = C.c_double
        lib.oracle_script_step.argtypes = [C.c_void_p, dp] + [C.c_double] * 6 + [C.c_int] * 2 + [C.c_double] * 4 + [ip]
        lib.oracle_reset.argtypes = [C.c_void_p, dp]
        lib.oracle_step.restype = C.c_double
        lib.oracle_step.argtypes = [C.c_void_p, dp, dp, ip]
        lib.oracle_rollout.restype = C.c_int
        lib.oracle_rollout.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), dp, dp, C.POINTER(C.c_uint8)]
        lib.oracle_batch_rollout.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_float), dp, dp, C.c_double, C.c_int, C.c_int, dp, dp, C.POINTER(C.c_uint8)]
        lib.oracle_set_reward_head.argtypes = [C.c_void_p, C.c_int]
        lib.oracle_ensemble.argtypes = [C.c_int] + [dp] * 6 + [C.c_double, C.c_uint, C.c_int, C.c_int, ip, dp, ip, dp]
        lib.oracle_foot_forces.argtypes = [C.c_void_p, dp]
        lib.oracle_gazebo_contact.argtypes = [dp]
        lib.oracle_gazebo_contact.restype = C.c_int
        lib.oracle_gazebo_script.argtypes = [C.c_void_p] + [C.c_double] * 7 + [C.c_int, ip, ip]
        lib.oracle_gazebo_script.restype = C.c_double
        _LIBS[dtype] = lib
    return _LIBS[dtype]


def batch_rollout(actions, mass_scale=None, lateral_friction=None, rolling=-1.0, body_contacts=True, threads=None, dtype="f64"):
    """actions float32 [T, N, 18] -> (obs [T, N, 26], rew [T, N], flags uint8 [T, N]): N independent oracle environments, each reset and
    rolled out with auto-reset, partitioned over host threads (full-size parity tests)."""
    lib = _lib(dtype)
    a = np.ascontiguousarray(actions, dtype=np.float32)
    T, n = a.shape[0], a.shape[1]
    obs = np.zeros((T, n, 26)); rew = np.zeros((T, n)); flags = np.zeros((T, n), dtype=np.uint8)
    ms = None if mass_scale is None else np.ascontiguousarray(mass_scale, dtype=np.float64)
    mu = None if lateral_friction is None else np.ascontiguousarray(lateral_friction, dtype=np.float64)
    if threads is None:
        try:
            threads = min(32, len(os.sched_getaffinity(0)))
        except AttributeError:
            threads = min(32, os.cpu_count() or 1)
    lib.oracle_batch_rollout(n, T, a.ctypes.data_as(C.POINTER(C.c_float)), None if ms is None else _dp(ms), None if mu is None else _dp(mu),
                             float(rolling), int(bool(body_contacts)), int(threads), _dp(obs), _dp(rew), flags.ctypes.data_as(C.POINTER(C.c_uint8)))
    return obs, rew, flags


def ensemble(n_episodes, actor=None, sigma=0.0, seed=0, threads=None, hyp=None, dtype="f64"):
    """n_episodes closed-loop episodes from reset on the oracle (C, threaded): actor = dict of float64 arrays fc1.weight .. fc3.bias (td3.py:19-57) or None
    for uniform random actions; N(0, sigma) action noise.  hyp: {oracle_set_hyp key number: value}.  Returns (lengths int[n], returns float[n])."""
    lib = _lib(dtype)
    if threads is None:
        try:
            threads = min(32, len(os.sched_getaffinity(0)))
        except AttributeError:
            threads = min(32, os.cpu_count() or 1)
    L = np.zeros(n_episodes, dtype=np.int32); R = np.zeros(n_episodes)
    if actor is not None:
        w = [np.ascontiguousarray(actor[k], dtype=np.float64) for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias")]
        assert w[0].shape == (256, 26) and w[2].shape == (256, 256) and w[4].shape == (18, 256)
        ptrs = [_dp(x) for x in w]
    else:
        ptrs = [None] * 6
    hk = np.array(list((hyp or {}).keys()), dtype=np.int32); hv = np.array(list((hyp or {}).values()), dtype=np.float64)
    lib.oracle_ensemble(int(n_episodes), *ptrs, float(sigma), int(seed), int(threads), len(hk), hk.ctypes.data_as(C.POINTER(C.c_int)), _dp(hv),
                        L.ctypes.data_as(C.POINTER(C.c_int)), _dp(R))
    return L, R


def native_lib():
    """The CPU-baseline build (gcc -O3 -march=native -fopenmp), always recompiled on the machine that calls this: -march=native code
    must not travel between hosts."""
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libplen_oracle_native.so"])
    lib = C.CDLL(os.path.join(_HERE, "libplen_oracle_native.so"))
    lib.oracle_throughput.restype = C.c_longlong
    lib.oracle_throughput.argtypes = [C.c_int, C.c_int, C.c_double, C.c_uint, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    return lib


def native_throughput(lib, n_envs, threads, budget_s, seed=0):
    """n_envs oracle environments, random actions, auto-reset, whole vector steps until budget_s has passed."""
    sec, vs = C.c_double(0), C.c_int(0)
    n = lib.oracle_throughput(int(n_envs), int(threads), float(budget_s), int(seed), C.byref(sec), C.byref(vs))
    return {"env_steps": int(n), "seconds": sec.value, "vector_steps": vs.value, "env_steps_per_s": n / sec.value}


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class OracleEnv(object):
    """One PLEN environment on the CPU oracle; mirrors PlenWalkEnv.reset/step (plen_env.py:558,638)."""

    def __init__(self, joint_act=False, dtype="f64", reward_head=0):
        self.lib = _lib(dtype)
        self.h = self.lib.oracle_create(int(joint_act))
        if reward_head:
            self.lib.oracle_set_reward_head(self.h, int(reward_head))   # 1: PlenWalkEnv-v0 contract (plen_walk.py)

    def foot_forces(self):
        """Contact force on the right and left foot over the last substep, [2,3] N."""
        f = np.zeros(6)
        self.lib.oracle_foot_forces(self.h, _dp(f))
        return f.reshape(2, 3)

    def gazebo_contact(self, force3):
        return int(self.lib.oracle_gazebo_contact(_dp(np.ascontiguousarray(force3, dtype=np.float64))))

    def gazebo_script(self, vx, z, y, roll, pitch, yaw, x, episode_timestep):
        """plen_walk.py:597-650 on hand-set attributes -> (reward, done, dead)."""
        done, dead = C.c_int(0), C.c_int(0)
        r = self.lib.oracle_gazebo_script(self.h, float(vx), float(z), float(y), float(roll), float(pitch), float(yaw), float(x),
                                          int(episode_timestep), C.byref(done), C.byref(dead))
        return r, bool(done.value), bool(dead.value)

    def __del__(self):
        try:
            self.lib.oracle_destroy(self.h)
        except Exception:
            pass

    def copy_from(self, other):
        """Make this environment an exact copy of `other` (state, contact cache, parameters)."""
        self.lib.oracle_copy(self.h, other.h)

    def set_params(self, mass_scale=1.0, lateral_friction=-1.0):
        self.lib.oracle_set_params(self.h, float(mass_scale), float(lateral_friction))

    def set_world(self, num_iterations=0, residual_threshold=-1.0):
        self.lib.oracle_set_world(self.h, int(num_iterations), float(residual_threshold))

    def set_friction(self, lateral=-1.0, spinning=-1.0, rolling=-1.0):
        self.lib.oracle_set_friction(self.h, float(lateral), float(spinning), float(rolling))

    def reset(self):
        obs = np.zeros(26)
        self.lib.oracle_reset(self.h, _dp(obs))
        return obs

    def step(self, action):
        a = np.ascontiguousarray(action, dtype=np.float64)
        obs = np.zeros(26)
        done = C.c_int(0)
        r = self.lib.oracle_step(self.h, _dp(a), _dp(obs), C.byref(done))
        return obs, r, bool(done.value), {}

    def rollout(self, actions):
        """actions float32 [T,18] -> obs [T,26], rew [T], flags uint8 [T] (bit0 terminal, bit1 time limit)."""
        a = np.ascontiguousarray(actions, dtype=np.float32)
        T = a.shape[0]
        obs = np.zeros((T, 26))
        rew = np.zeros(T)
        flags = np.zeros(T, dtype=np.uint8)
        self.lib.oracle_rollout(self.h, T, a.ctypes.data_as(C.POINTER(C.c_float)), _dp(obs), _dp(rew),
                                flags.ctypes.data_as(C.POINTER(C.c_uint8)))
        return obs, rew, flags

    # ---- state access -------------------------------------------------------------------
    def get_state(self):
        s = np.zeros(49)
        self.lib.oracle_get_state(self.h, _dp(s))
        return s

    def set_state(self, s):
        s = np.ascontiguousarray(s, dtype=np.float64)
        assert s.shape == (49,)
        self.lib.oracle_set_state(self.h, _dp(s))

    def get_aux(self):
        a = (C.c_int * 6)()
        self.lib.oracle_get_aux(self.h, a)
        return dict(zip(["gait_period_counter", "double_support_counter", "episode_timestep", "dead", "first_pass", "nhist"], list(a)))

    def set_targets(self, t):
        t = np.ascontiguousarray(t, dtype=np.float64)
        self.lib.oracle_set_targets(self.h, _dp(t))

    def substep(self):
        self.lib.oracle_substep(self.h)

    def contacts(self):
        a = (C.c_int * 4)()
        self.lib.oracle_contacts(self.h, a)
        return dict(right=a[0], left=a[1], ncp=a[2], iterations=a[3])

    def contact_slots(self, run_collide=False):
        """The 8 contact-point slots of the last collision pass: owner (-2 empty, -1 foot point, >= 0 box index) and point positions."""
        box = (C.c_int * 8)(); pos = np.zeros(24)
        self.lib.oracle_contact_slots(self.h, int(run_collide), box, _dp(pos))
        return np.array(list(box)), pos.reshape(8, 3)

    def set_body_contacts(self, on=True):
        """False: only the feet collide with the ground (the round-1 contact model)."""
        self.lib.oracle_set_body_contacts(self.h, int(bool(on)))

    def forward_dynamics(self):
        qdd = np.zeros(24)
        self.lib.oracle_forward_dynamics(self.h, _dp(qdd))
        return qdd

    def minv_times(self, f):
        f = np.ascontiguousarray(f, dtype=np.float64)
        out = np.zeros(24)
        self.lib.oracle_minv_times(self.h, _dp(f), _dp(out))
        return out

    def link_frames(self):
        R = np.zeros((33, 9)); O = np.zeros((33, 3)); Cc = np.zeros((33, 3))
        self.lib.oracle_link_frames(self.h, _dp(R), _dp(O), _dp(Cc))
        return R.reshape(33, 3, 3), O, Cc

    # ---- python-level arithmetic pins -----------------------------------------------------
    def script_reset(self):
        self.lib.oracle_script_reset(self.h)

    def script_step(self, q18, z, vx, roll, pitch, yaw, y, rc, lc, lroll, lpitch, rroll, rpitch):
        q = np.ascontiguousarray(q18, dtype=np.float64)
        done = C.c_int(0)
        r = self.lib.oracle_script_step(self.h, _dp(q), z, vx, roll, pitch, yaw, y, int(rc), int(lc),
                                        lroll, lpitch, rroll, rpitch, C.byref(done))
        return r, bool(done.value)


def agent_to_env(joint, a, dtype="f64"):
    return _lib(dtype).oracle_agent_to_env(int(joint), float(a))


def cpu_baseline_main(budget_s=12.0, n_envs=4096):
    """`python -m oracle.oracle baseline [budget_s]`: the CPU-baseline measurement of bench.py, run in its own process (own OpenMP runtime,
    passive waiting).  (a) one thread, one env; (b) n_envs envs partitioned over T threads for T in a short sweep up to the cores this process
    may use -- GPU boxes expose many more hardware threads than their CPU quota sustains, so the fastest T is kept and reported."""
    import json
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    lib = native_lib()
    one = native_throughput(lib, 1, 1, min(3.0, budget_s / 4))
    sweep, t = [], 4
    while t < cores:
        sweep.append(t); t *= 2
    sweep.append(cores)
    probe = {}
    for t in sweep:
        probe[t] = native_throughput(lib, n_envs, t, 1.5)["env_steps_per_s"]
    best = max(probe, key=probe.get)
    many = native_throughput(lib, n_envs, best, budget_s)
    print(json.dumps({"one": one, "many": many, "threads": best, "cores_visible": cores, "sweep": {str(k): v for k, v in probe.items()}}))


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "baseline":
        cpu_baseline_main(float(sys.argv[2]) if len(sys.argv) > 2 else 12.0)
