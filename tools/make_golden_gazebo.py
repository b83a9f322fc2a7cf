#!/usr/bin/env python3
"""Golden vectors for the Gazebo-style reward head: the reference's PlenWalkEnv-v0
`_is_done` / `_compute_reward` (plen_ros/src/plen_ros_helpers/plen_walk.py:597-650) and its contact rule
(force magnitude > 4.8559/3 N, :346-396), evaluated by IMPORTING the reference with the ROS stack stubbed.
Writes tests/golden/gazebo_reward_done.npz.  Run in the build container only (needs /root/reference)."""
import contextlib, io, os, sys, types
import numpy as np
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PLEN_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")


class _Dummy(object):
    def __init__(self, *a, **k): pass
    def __call__(self, *a, **k): return _Dummy()
    def __getattr__(self, n):
        if n.startswith("__"): raise AttributeError(n)
        return _Dummy()


class _StubModule(types.ModuleType):
    def __getattr__(self, n):
        if n.startswith("__"): raise AttributeError(n)
        return _Dummy


class _Vec3(object):
    def __init__(self, x=0.0, y=0.0, z=0.0): self.x, self.y, self.z = x, y, z


def main():
    for n in ("rospy", "nav_msgs", "nav_msgs.msg", "geometry_msgs", "geometry_msgs.msg", "gazebo_msgs", "gazebo_msgs.msg", "sensor_msgs",
              "sensor_msgs.msg", "tf", "tf.transformations", "plen_ros", "plen_ros.srv", "std_msgs", "std_msgs.msg", "std_srvs", "std_srvs.srv",
              "gym", "gym.spaces", "gym.envs", "gym.envs.registration", "gym.utils", "gym.utils.seeding", "controller_manager_msgs",
              "controller_manager_msgs.srv", "gazebo_msgs.srv", "trajectory_msgs", "trajectory_msgs.msg"):
        sys.modules[n] = _StubModule(n)
    sys.modules["rospy"].logdebug = lambda *a, **k: None
    sys.modules["gym"].spaces = sys.modules["gym.spaces"]
    sys.modules["geometry_msgs.msg"].Vector3 = _Vec3
    sys.modules["gym.envs.registration"].register = lambda **k: None
    base = types.ModuleType("plen_ros_helpers.plen_env")
    class PlenEnv(object):
        def __init__(self, *a, **k): pass
    base.PlenEnv = PlenEnv
    sys.path.insert(0, os.path.join(REF, "plen_ros/src"))
    import plen_ros_helpers                      # the real package (namespace of the reference helpers)
    sys.modules["plen_ros_helpers.plen_env"] = base
    with contextlib.redirect_stdout(io.StringIO()):
        from plen_ros_helpers import plen_walk as ref
        env = ref.PlenWalkEnv()
    weights = np.array([env.dead_penalty, env.alive_reward, env.vel_weight, env.init_height, env.height_weight, env.straight_weight,
                        env.roll_weight, env.pitch_weight, env.yaw_weight, env.max_episode_steps])
    rng = np.random.default_rng(11)
    n = 2000
    st = np.zeros((n, 8))       # vx z y roll pitch yaw x episode_timestep
    st[:, 0] = rng.normal(0, 0.3, n); st[:, 1] = rng.uniform(0.05, 0.2, n); st[:, 2] = rng.normal(0, 0.6, n)
    st[:, 3:6] = rng.normal(0, 0.6, (n, 3)); st[:, 6] = rng.uniform(-0.5, 2.0, n); st[:, 7] = rng.integers(0, 520, n)
    st[:16, 0] = 0.0                                     # np.sign(0) branch
    out = np.zeros((n, 3))      # done dead reward
    for i in range(n):
        env.torso_vx, env.torso_z, env.torso_y, env.torso_roll, env.torso_pitch, env.torso_yaw, env.torso_x = st[i, :7]
        env.episode_timestep = int(st[i, 7]); env.dead = False
        done = env._is_done(None)
        dead = env.dead
        out[i] = (float(done), float(dead), float(env._compute_reward(None, done)))
    # contact rule: last state's total_wrench.force magnitude against weight/3
    forces = rng.normal(0, 1.5, (256, 3)); flags = np.zeros((256, 2))
    for i in range(256):
        msg = types.SimpleNamespace(states=[types.SimpleNamespace(total_wrench=types.SimpleNamespace(force=_Vec3(*forces[i])))])
        env.right_contact_subscriber_callback(msg); env.left_contact_subscriber_callback(msg)
        flags[i] = (env.right_contact, env.left_contact)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "gazebo_reward_done.npz"), weights=weights, states=st, done_dead_reward=out, forces=forces, flags=flags)
    print("gazebo_reward_done.npz: %d states, done %.0f dead %.0f, reward range [%.3f, %.3f], contacts %d/256" %
          (n, out[:, 0].sum(), out[:, 1].sum(), out[:, 2].min(), out[:, 2].max(), int(flags[:, 0].sum())))


if __name__ == "__main__":
    main()
