#!/usr/bin/env python3
"""Generate golden input/output vectors from the REFERENCE ITSELF (run in the build container only).

The reference's Python-level arithmetic is importable with stub modules for the packages that are
absent here (gym, pybullet, pybullet_data): see SURVEY.md section 8c.  This script
  * imports /root/reference/plen_bullet/src/plen_bullet/plen_env.py and
    /root/reference/plen_ros/src/plen_ros_helpers/td3.py with `sys.dont_write_bytecode = True`
    (never writes into /root/reference),
  * drives them with seeded inputs through a scripted fake physics client,
  * stores inputs + the reference's outputs as small .npz fixtures under tests/golden/.
Only data is written; no reference source text is copied.

Fixtures:
  a1_agent_to_env.npz    action map (plen_env.py:694-714) incl. the clamps
  a78_reward_done.npz    compute_done / compute_reward on 1200 random single states
  a69_script.npz         600-step scripted episodes through env.step()/reset(): obs, reward, done,
                         motor targets, gait counters  (A6, A8, A9 state machine)
  td3_forward.npz        shipped checkpoint 3229999: actor(obs), critic(obs, act) on 64 inputs
  td3_train.npz          two TD3Agent.train() iterations from a fixed buffer with the sampled
                         indices and noise captured, initial and final parameters
  traj_gen.npz           trajectory_generator.py IK / foot paths known answers
"""
import io
import os
import sys
import types
import contextlib
import numpy as np

sys.dont_write_bytecode = True
REF = os.environ.get("PLEN_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


# --------------------------------------------------------------------------- stub modules
class Box(object):
    def __init__(self, low, high, dtype=np.float32):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        self.shape = self.low.shape
        self.dtype = dtype


def euler_from_quat(q):
    """pybullet.getEulerFromQuaternion (public formula)."""
    x, y, z, w = q
    sarg = -2 * (x * z - w * y)
    if sarg <= -0.99999:
        return (0.0, -0.5 * np.pi, 2 * np.arctan2(x, -y))
    if sarg >= 0.99999:
        return (0.0, 0.5 * np.pi, 2 * np.arctan2(-x, y))
    return (np.arctan2(2 * (y * z + w * x), w * w - x * x - y * y + z * z), np.arcsin(sarg),
            np.arctan2(2 * (x * y + w * z), w * w + x * x - y * y - z * z))


class FakeBullet(types.ModuleType):
    """A scripted physics client: the env reads whatever `self.frame` currently holds."""
    GUI, DIRECT, POSITION_CONTROL = 1, 2, 3

    def __init__(self):
        super().__init__("pybullet")
        self.frame = dict(pos=[0, 0, 0.158], quat=[0, 0, 0, 1], q=np.zeros(18), linvel=[0, 0, 0], angvel=[0, 0, 0],
                          rc=0, lc=0, lquat=[0, 0, 0, 1], rquat=[0, 0, 0, 1])
        self.targets = []
        self.n_step = 0
        self.n_filter_pairs = 0

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return lambda *a, **k: None

    def connect(self, *a, **k): return 0
    def getNumJoints(self, *a, **k): return 32
    def getJointInfo(self, body, i): return (i, b"joint%d" % i, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, b"link%d" % i)
    def getBodyInfo(self, *a, **k): return (b"torso", b"plen")
    def loadURDF(self, name, *a, **k): return 0 if "plane" in name else 1
    def getQuaternionFromEuler(self, e): return (0.0, 0.0, 0.0, 1.0)
    def getEulerFromQuaternion(self, q): return euler_from_quat(q)
    def setCollisionFilterPair(self, *a, **k): self.n_filter_pairs += 1
    def stepSimulation(self): self.n_step += 1
    def setJointMotorControlArray(self, **k): self.targets.append(np.array(k["targetPositions"], dtype=np.float64))
    def getBasePositionAndOrientation(self, b):
        # the reference wraps this in np.array(...) (plen_env.py:771): ragged, so hand NumPy 2 an
        # object array, which is what NumPy 1.x built implicitly
        out = np.empty(2, dtype=object)
        out[0] = tuple(float(v) for v in self.frame["pos"]); out[1] = tuple(float(v) for v in self.frame["quat"])
        return out
    def getJointStates(self, b, idx): return [(float(self.frame["q"][i]), 0.0, (0,) * 6, 0.0) for i in range(18)]
    def getBaseVelocity(self, b): return (tuple(self.frame["linvel"]), tuple(self.frame["angvel"]))
    def getContactPoints(self, a, b, link): return [1] * int(self.frame["lc"] if link == 19 else self.frame["rc"])
    def getLinkState(self, b, link): return ((0, 0, 0), tuple(self.frame["lquat"] if link == 19 else self.frame["rquat"]))


def install_stubs():
    gym = types.ModuleType("gym")
    gym.Env = object
    spaces = types.ModuleType("gym.spaces"); spaces.Box = Box
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    seeding.np_random = lambda seed=None: (np.random.RandomState(seed), seed)
    utils.seeding = seeding
    envs = types.ModuleType("gym.envs")
    registration = types.ModuleType("gym.envs.registration")
    registered = {}
    registration.register = lambda **k: registered.update({k["id"]: k})
    envs.registration = registration
    gym.spaces = spaces; gym.utils = utils; gym.envs = envs
    gym.make = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("gym.make is not stubbed"))
    pb = FakeBullet()
    pbd = types.ModuleType("pybullet_data"); pbd.getDataPath = lambda: "/nonexistent"
    for n, m in (("gym", gym), ("gym.spaces", spaces), ("gym.utils", utils), ("gym.utils.seeding", seeding),
                 ("gym.envs", envs), ("gym.envs.registration", registration), ("pybullet", pb), ("pybullet_data", pbd)):
        sys.modules[n] = m
    return pb, registered


def quat_from_rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p / 2), np.sin(p / 2), np.cos(y / 2), np.sin(y / 2)
    return [sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy]


def main():
    os.makedirs(OUT, exist_ok=True)
    pb, registered = install_stubs()
    sys.path.insert(0, os.path.join(REF, "plen_bullet/src"))
    sys.path.insert(0, os.path.join(REF, "plen_ros/src"))
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        from plen_bullet import plen_env as ref_env
        env = ref_env.PlenWalkEnv()
    assert registered["PlenWalkEnv-v1"]["max_episode_steps"] == 500
    assert env.sim_stepsize == 4 and pb.n_filter_pairs == 457
    rng = np.random.default_rng(20261002)

    # ---------------- A1: action map -------------------------------------------------------
    grid = np.array([-1.0, -0.999999, -0.9995, -0.5, 0.0, 0.5, 0.9995, 0.999999, 1.0, 1.5, -1.5])
    acts = np.concatenate([np.tile(grid[:, None], (1, 18)), rng.uniform(-1, 1, (200, 18)).astype(np.float32).astype(np.float64)])
    out = np.zeros_like(acts)
    for i in range(acts.shape[0]):
        for j in range(18):
            out[i, j] = env.agent_to_env(env.env_ranges[j], float(acts[i, j]))
    np.savez_compressed(os.path.join(OUT, "a1_agent_to_env.npz"), action=acts, env_action=out,
                        env_ranges=np.array(env.env_ranges), real_ranges=np.array(env.real_ranges),
                        obs_low=env.observation_space.low.astype(np.float64), obs_high=env.observation_space.high.astype(np.float64),
                        moving_joints=np.array(env.movingJoints))

    # ---------------- A7/A8: single-state reward / done ------------------------------------
    N = 1200
    rec = dict(vx=[], z=[], y=[], roll=[], pitch=[], yaw=[], rc=[], lc=[], cnt=[], ds=[], first=[], nh=[],
               hist=[], diffs=[], lrp=[], rrp=[], reward=[], done=[], cnt_after=[], ds_after=[], nh_after=[])
    for i in range(N):
        vx = rng.normal() * 0.3 if rng.uniform() < 0.9 else 0.0
        z = 0.16 + rng.normal() * 0.03
        y = rng.normal() * (0.6 if rng.uniform() < 0.2 else 0.05)
        roll, pitch, yaw = rng.normal(size=3) * np.array([0.5, 0.5, 0.8])
        rc, lc = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        cnt = int(rng.choice([0, 1, 5, 39, 40, 41, 79, 80, 81, 119, 120, 121, 200, int(rng.integers(0, 130))]))
        ds = int(rng.choice([0, 14, 15, 16, 30]))
        nh = int(rng.integers(1, 24)) if cnt > 0 else int(rng.integers(1, 3))
        hist = rng.uniform(-1, 1, (6, nh))
        diffs = rng.normal(size=6) * 0.05
        first = bool(rng.uniform() < 0.2)
        lrp = rng.normal(size=2) * 0.12
        rrp = rng.normal(size=2) * 0.12
        env.torso_vx, env.torso_z, env.torso_y = vx, z, y
        env.torso_roll, env.torso_pitch, env.torso_yaw = roll, pitch, yaw
        env.right_contact, env.left_contact = rc, lc
        env.gait_period_counter, env.double_support_preriod_counter = cnt, ds
        (env.lhip_joint_angles, env.rhip_joint_angles, env.lknee_joint_angles, env.rknee_joint_angles,
         env.lankle_joint_angles, env.rankle_joint_angles) = [h.copy() for h in hist]
        (env.lhip_joint_angle_diff, env.rhip_joint_angle_diff, env.lknee_joint_angle_diff, env.rknee_joint_angle_diff,
         env.lankle_joint_angle_diff, env.rankle_joint_angle_diff) = diffs
        env.first_pass = first
        pb.frame["lquat"] = quat_from_rpy(lrp[0], lrp[1], rng.normal())
        pb.frame["rquat"] = quat_from_rpy(rrp[0], rrp[1], rng.normal())
        done = env.compute_done()
        reward = env.compute_reward()
        hp = np.zeros((6, 24)); hp[:, :nh] = hist
        for k, v in dict(vx=vx, z=z, y=y, roll=roll, pitch=pitch, yaw=yaw, rc=rc, lc=lc, cnt=cnt, ds=ds, first=first, nh=nh,
                         hist=hp, diffs=diffs, lrp=np.array(euler_from_quat(pb.frame["lquat"])[:2]),
                         rrp=np.array(euler_from_quat(pb.frame["rquat"])[:2]), reward=reward, done=done,
                         cnt_after=env.gait_period_counter, ds_after=env.double_support_preriod_counter,
                         nh_after=len(env.lhip_joint_angles)).items():
            rec[k].append(v)
    np.savez_compressed(os.path.join(OUT, "a78_reward_done.npz"), **{k: np.array(v) for k, v in rec.items()})

    # ---------------- A6/A8/A9: scripted episodes through step()/reset() -------------------
    episodes = []
    for ep in range(4):
        T = 600
        t = np.arange(T)
        ph = 2 * np.pi * t / (60 + 15 * ep)
        q = 0.4 * np.sin(ph[:, None] + rng.uniform(0, 6.28, 18)[None, :]) * rng.uniform(0.2, 1.0, 18)[None, :]
        if ep == 1:
            q[:, [2, 3, 4]] = -q[:, [8, 9, 10]]            # mirrored legs -> cosine similarity -1
        if ep == 2:
            q[40:60] = q[40]                                # standing still -> diff penalties saturate
        pos = np.stack([0.002 * t, 0.05 * np.sin(ph / 3), 0.16 + 0.01 * np.sin(ph)], 1)
        rpy = np.stack([0.1 * np.sin(ph), 0.15 * np.cos(ph / 2), 0.3 * np.sin(ph / 5)], 1)
        linvel = np.stack([0.2 * np.sin(ph / 2) - 0.02, 0 * ph, 0 * ph], 1)
        rc = (np.sin(ph) > -0.2).astype(int); lc = (np.sin(ph) < 0.2).astype(int)
        if ep == 3:
            pos[300:, 2] -= np.linspace(0, 0.12, T - 300)   # sinks below 0.08 -> done + dead penalty
            rpy[200:220, 0] = 1.2                           # roll beyond pi/3 -> done
            pos[250:260, 1] = 1.3                           # y > 1 -> done
        frp = rng.normal(size=(T, 4)) * 0.08
        acts = rng.uniform(-1.1, 1.1, (T, 18))
        pb.targets = []
        pb.frame.update(pos=[0, 0, 0.158], quat=[0, 0, 0, 1], q=q[0], linvel=[0, 0, 0], rc=1, lc=1)
        with contextlib.redirect_stdout(sink):
            obs0 = env.reset()
        n_reset_substeps = pb.n_step
        o_l, r_l, d_l, c_l = [], [], [], []
        for k in range(T):
            pb.frame.update(pos=pos[k], quat=quat_from_rpy(*rpy[k]), q=q[k], linvel=linvel[k], rc=rc[k], lc=lc[k],
                            lquat=quat_from_rpy(frp[k, 0], frp[k, 1], 0.3), rquat=quat_from_rpy(frp[k, 2], frp[k, 3], -0.2))
            o, r, d, _ = env.step(acts[k])
            o_l.append(o); r_l.append(r); d_l.append(d)
            c_l.append([env.gait_period_counter, env.double_support_preriod_counter, env.episode_timestep,
                        len(env.lhip_joint_angles), int(env.first_pass)])
        episodes.append(dict(q=q, pos=pos, rpy=rpy, linvel=linvel, rc=rc, lc=lc, frp=frp, actions=acts, obs0=obs0,
                             obs=np.array(o_l), reward=np.array(r_l), done=np.array(d_l), counters=np.array(c_l),
                             targets=np.array(pb.targets[1:])))   # targets[0] is reset's zeros
    np.savez_compressed(os.path.join(OUT, "a69_script.npz"),
                        **{"%s_%d" % (k, i): v for i, e in enumerate(episodes) for k, v in e.items()})

    # ---------------- trajectory generator --------------------------------
    try:
        mpl = types.ModuleType("matplotlib"); plt = types.ModuleType("matplotlib.pyplot"); mpl.pyplot = plt
        sys.modules.setdefault("matplotlib", mpl); sys.modules.setdefault("matplotlib.pyplot", plt)
        gym_mod = sys.modules["gym"]
        gym_mod.make = lambda *a, **k: env
        with contextlib.redirect_stdout(sink):
            from plen_bullet import trajectory_generator as tg
            gen = tg.TrajectoryGenerator()
        tj = {}
        for name, kw in (("default", {}), ("eval", dict(num_DoubleSupport=20, num_SingleSupport=20, height=20.0, stride=20.0))):
            with contextlib.redirect_stdout(sink):
                gen = tg.TrajectoryGenerator(**kw)
                gen.main()
            tj.update({name + ".foot_walk_rfwd_r": gen.foot_walk_rfwd_r, name + ".foot_walk_lfwd_r": gen.foot_walk_lfwd_r,
                       name + ".foot_walk_rfwd": gen.foot_walk_rfwd, name + ".foot_walk_lfwd": gen.foot_walk_lfwd,
                       name + ".bend": gen.bend,
                       name + ".params": np.array([gen.num_DoubleSupport, gen.num_SingleSupport, gen.foot_lift_height,
                                                   gen.stride_length, gen._bend_distance, gen._body_sway, gen.fwd_bias])})
        pts = (rng.uniform(-1, 1, (3, 64)) * np.array([[25.0], [12.0], [10.0]]) + np.array([[0.0], [0.0], [12.0]]))
        with contextlib.redirect_stdout(sink):
            tj.update(ik_points=pts, ik_right=gen.IK(pts, True), ik_left=gen.IK(pts, False))
        np.savez_compressed(os.path.join(OUT, "traj_gen.npz"), **tj)
    except Exception as ex:            # optional fixture; reported, not fatal
        print("trajectory generator fixture skipped:", repr(ex))

    # ---------------- TD3 ------------------------------------------------------------------
    import torch
    from plen_ros_helpers import td3 as ref_td3
    torch.manual_seed(0); np.random.seed(0)
    ckpt = os.path.join(REF, "plen_bullet/models/plen_walk_gazebo_3229999")
    actor_sd = torch.load(ckpt + "_actor", map_location="cpu", weights_only=True)
    critic_sd = torch.load(ckpt + "_critic", map_location="cpu", weights_only=True)
    agent = ref_td3.TD3Agent(26, 18, 1.0)
    agent.actor.load_state_dict(actor_sd); agent.critic.load_state_dict(critic_sd)
    obs = rng.normal(size=(64, 26)) * 0.3
    obs[:, 18] += 0.16
    act = np.stack([agent.select_action(o) for o in obs])
    with torch.no_grad():
        q1, q2 = agent.critic(torch.FloatTensor(obs), torch.FloatTensor(act))
    np.savez_compressed(os.path.join(OUT, "td3_forward.npz"), obs=obs, action=act, q1=q1.numpy(), q2=q2.numpy(),
                        actor_param_count=sum(p.numel() for p in agent.actor.parameters()),
                        critic_param_count=sum(p.numel() for p in agent.critic.parameters()))
    # the shipped policy itself, as plain arrays (data artefact of the reference)
    np.savez_compressed(os.path.join(OUT, "policy_3229999.npz"),
                        **{"actor." + k: v.numpy() for k, v in actor_sd.items()},
                        **{"critic." + k: v.numpy() for k, v in critic_sd.items()})

    # two train() iterations with all randomness captured
    torch.manual_seed(123)
    agent = ref_td3.TD3Agent(26, 18, 1.0)
    init = {"actor." + k: v.detach().clone().numpy() for k, v in agent.actor.state_dict().items()}
    init.update({"critic." + k: v.detach().clone().numpy() for k, v in agent.critic.state_dict().items()})
    buf = ref_td3.ReplayBuffer()
    NB = 300
    S = rng.normal(size=(NB, 26)); A = rng.uniform(-1, 1, (NB, 18)); S2 = S + rng.normal(size=(NB, 26)) * 0.1
    Rw = rng.normal(size=NB); Dn = (rng.uniform(size=NB) < 0.1).astype(np.float64)
    for i in range(NB):
        buf.add((S[i], A[i], S2[i], np.array(Rw[i]), np.array(Dn[i])))
    idx_log, noise_log = [], []
    real_randint, real_randn_like = np.random.randint, torch.randn_like

    def rec_randint(*a, **k):
        v = real_randint(*a, **k); idx_log.append(np.array(v)); return v

    def rec_randn_like(x, *a, **k):
        v = real_randn_like(x, *a, **k); noise_log.append(v.clone().numpy()); return v

    np.random.randint = rec_randint; torch.randn_like = rec_randn_like
    B = 100
    snaps = []
    for it in range(2):
        agent.train(buf, B)
        snap = {"actor." + k: v.detach().clone().numpy() for k, v in agent.actor.state_dict().items()}
        snap.update({"critic." + k: v.detach().clone().numpy() for k, v in agent.critic.state_dict().items()})
        snap.update({"actor_target." + k: v.detach().clone().numpy() for k, v in agent.actor_target.state_dict().items()})
        snap.update({"critic_target." + k: v.detach().clone().numpy() for k, v in agent.critic_target.state_dict().items()})
        snaps.append(snap)
    np.random.randint = real_randint; torch.randn_like = real_randn_like
    save = dict(S=S, A=A, S2=S2, R=Rw, D=Dn, idx=np.array(idx_log), noise=np.array(noise_log), batch=B)
    save.update({"init." + k: v for k, v in init.items()})
    # per iteration: full small tensors + (sum, abs-sum) of every tensor (keeps the fixture small)
    for it, snap in enumerate(snaps):
        for k, v in snap.items():
            if v.size <= 18 * 256:
                save["it%d.%s" % (it + 1, k)] = v
            save["it%d.sum.%s" % (it + 1, k)] = np.array([v.astype(np.float64).sum(), np.abs(v.astype(np.float64)).sum()])
    np.savez_compressed(os.path.join(OUT, "td3_train.npz"), **save)
    for f in sorted(os.listdir(OUT)):
        print("%-28s %8d bytes" % (f, os.path.getsize(os.path.join(OUT, f))))


if __name__ == "__main__":
    main()
