"""PyBullet comparison (north_star: observations within 1e-4 of PyBullet).  PyBullet is not installed on any machine this project runs
on (profiles/r02_pybullet_probe_*.json), so the comparison tests skip themselves; what CAN run everywhere is the check that the robot
description the harness would hand to PyBullet is this repository's model."""
import os
import xml.etree.ElementTree as ET
import numpy as np
import pytest

import pybullet_harness as H

try:
    import pybullet  # noqa: F401
    HAVE_PYBULLET = True
except Exception:
    HAVE_PYBULLET = False


def test_emitted_urdf_is_the_model(tmp_path):
    path = H.emit_urdf(str(tmp_path))
    root = ET.parse(path).getroot()
    links, joints = root.findall("link"), root.findall("joint")
    assert len(links) == 33 and len(joints) == 32
    assert abs(sum(float(l.find("inertial/mass").get("value")) for l in links) - 0.495834) < 1e-9        # total mass (cf. plen_walk.py:350 "0.495Kg")
    # depth-first numbering over children in file order (what Bullet's URDF importer does) reproduces movingJoints (plen_env.py:318-320)
    children = {}
    for j in joints:
        children.setdefault(j.find("parent").get("link"), []).append(j)
    order = []

    def dfs(name):
        for j in children.get(name, []):
            order.append(j); dfs(j.find("child").get("link"))
    dfs("torso")
    assert [i for i, j in enumerate(order) if j.get("type") == "revolute"] == [5, 6, 7, 9, 10, 11, 13, 14, 15, 17, 18, 19, 20, 21, 24, 26, 27, 30]
    assert [j.find("child").get("link") for j in order] == [l["name"] for l in H.MODEL["links"]]
    boxes = [l for l in links if l.find("collision/geometry/box") is not None]
    meshes = [l for l in links if l.find("collision/geometry/mesh") is not None]
    assert len(boxes) == 31 and sorted(m.get("name") for m in meshes) == ["l_foot", "r_foot"]
    for m in meshes:
        fn = m.find("collision/geometry/mesh").get("filename")
        v = np.array([[float(x) for x in ln.split()[1:]] for ln in open(fn) if ln.startswith("v ")])
        assert v.shape == (209, 3) or v.shape[1] == 3 and len(v) > 100
    # joint frames survive the rpy round trip
    for l in H.MODEL["links"]:
        j = [q for q in joints if q.get("name") == l["joint"]][0]
        r, p, y = (float(x) for x in j.find("origin").get("rpy").split())
        cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
        R = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]]) @ np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]]) @ np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
        assert np.abs(R - np.array(l["R"])).max() < 1e-10          # (the URDF writes pi/2 with 10 digits: gimbal-locked frames come back to 5e-12)


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE_PYBULLET, reason="pybullet is not installed on this machine (see profiles/r02_pybullet_probe_*.json)")
def test_observations_against_pybullet(tmp_path):
    """Runs only where PyBullet exists.  Reports rather than presumes: the hypotheses of DESIGN.md section 2 (warm starting, row order,
    inertia from shape, manifold) are confirmed to the extent these numbers are small."""
    res = H.compare_with_oracle(str(tmp_path))
    print(res)
    assert res["reset_err"] <= 1e-3
    assert res["first_step_contact_flags_equal"] >= 0.9


class _StubBullet(object):
    """A stand-in for the `pybullet` MODULE with exactly the public calls tests/pybullet_harness.py makes, its physics supplied by this repository's oracle:
    the harness cannot be run against the real engine on this project's machines, and a one-command check that is never executed rots.
    Driving it through the stub executes every line of the harness -- robot emission, world set-up, the reference's call order in reset / step / observe --
    and, because the engine behind the stub IS the oracle, the comparison it reports must come out at exactly zero."""
    DIRECT, POSITION_CONTROL, JOINT_REVOLUTE, JOINT_FIXED = 2, 2, 0, 4
    __version__ = "stub (oracle engine)"

    def __init__(self):
        from oracle.oracle import OracleEnv
        self.o = OracleEnv()
        self.state = np.zeros(49)
        self.calls = []

    def __getattr__(self, name):              # any call the harness starts to make that the stub does not know fails loudly
        raise AttributeError("pybullet stub has no %r: extend tests/test_pybullet_parity.py::_StubBullet" % name)

    def connect(self, mode):
        self.calls.append("connect"); return 0

    def disconnect(self, cid):
        self.calls.append("disconnect")

    def setRealTimeSimulation(self, v): pass
    def resetSimulation(self): pass
    def setAdditionalSearchPath(self, path): pass
    def getAPIVersion(self): return 0

    def setGravity(self, x, y, z):
        assert (x, y, z) == (0, 0, -9.81)

    def loadURDF(self, path, pos=None, quat=None):
        if path == "plane.urdf":
            return 0
        root = ET.parse(path).getroot()       # the emitted robot: what a real importer would be handed
        assert len(root.findall("joint")) == 32 and pos == [0, 0, 0.158]
        return 1

    def changeDynamics(self, body, link, **kw):
        self.calls.append(("changeDynamics", body, link, tuple(sorted(kw.items()))))

    def getNumJoints(self, body): return 32

    def getJointInfo(self, body, j):
        return (j, b"joint", self.JOINT_REVOLUTE if j in H.MOVING else self.JOINT_FIXED)

    def getQuaternionFromEuler(self, e):
        assert list(e) == [0, 0, 0]
        return (0.0, 0.0, 0.0, 1.0)

    def getEulerFromQuaternion(self, q):       # pybullet.c getEulerFromQuaternion, as the oracle's observation computes it
        x, y, z, w = q
        sarg = -2.0 * (x * z - w * y)
        if sarg <= -0.99999:
            return (0.0, -0.5 * np.pi, 2 * np.arctan2(x, -y))
        if sarg >= 0.99999:
            return (0.0, 0.5 * np.pi, 2 * np.arctan2(-x, y))
        return (np.arctan2(2 * (y * z + w * x), w * w - x * x - y * y + z * z), np.arcsin(sarg), np.arctan2(2 * (x * y + w * z), w * w + x * x - y * y - z * z))

    def resetBasePositionAndOrientation(self, body, pos, quat):
        self.state[:] = 0.0
        self.state[0:3] = pos; self.state[3:7] = quat
        self.o.reset()                        # (clears the oracle's episode bookkeeping; the state is injected next)
        self.o.set_state(self.state)

    def resetJointState(self, body, j, q):
        k = H.MOVING.index(j)
        self.state[13 + k] = q; self.state[31 + k] = 0.0
        self.o.set_state(self.state)

    def setJointMotorControlArray(self, bodyUniqueId, jointIndices, controlMode, targetPositions, forces):
        assert list(jointIndices) == H.MOVING and controlMode == self.POSITION_CONTROL and list(forces) == [0.15] * 18
        self.o.set_targets(np.asarray(targetPositions, dtype=np.float64))

    def stepSimulation(self):
        self.o.substep()
        self.state = self.o.get_state()

    def getBasePositionAndOrientation(self, body):
        s = self.o.get_state(); return tuple(s[0:3]), tuple(s[3:7])

    def getBaseVelocity(self, body):
        s = self.o.get_state(); return tuple(s[10:13]), tuple(s[7:10])

    def getJointStates(self, body, joints):
        s = self.o.get_state(); return [(s[13 + k], s[31 + k], None, 0.0) for k in range(18)]

    def getContactPoints(self, a, b, linkIndexA):
        c = self.o.contacts()
        return [()] * int(c["right"] if linkIndexA == 11 else c["left"])


def test_harness_runs_end_to_end_against_a_stub_pybullet(tmp_path, monkeypatch):
    """tests/pybullet_harness.py -- the one command to run the day PyBullet is at hand -- executed here against a stub `pybullet` whose engine is the oracle:
    the emitted robot loads, the reference's call order (plen_env.py:275-315, 439-481, 558-570, 638-667, 768-822) goes through, the foot / link dynamics are
    set as the reference sets them, and the harness's comparison of "PyBullet" with the oracle reports zero error on the reset stance, on 8 first steps and on
    20 steps of the recorded policy commands; its step timer returns a rate."""
    import sys
    import types
    stub = _StubBullet()
    mod = types.ModuleType("pybullet")
    for name in dir(stub):
        if not name.startswith("_") or name == "__version__":
            setattr(mod, name, getattr(stub, name))
    data = types.ModuleType("pybullet_data"); data.getDataPath = lambda: str(tmp_path)
    monkeypatch.setitem(sys.modules, "pybullet", mod); monkeypatch.setitem(sys.modules, "pybullet_data", data)
    res = H.compare_with_oracle(str(tmp_path), n_first_steps=8, replay_steps=20)
    assert res["reset_err"] <= 1e-12 and res["first_step_max_abs_err"] <= 1e-12 and res["first_step_contact_flags_equal"] == 1.0
    assert res["policy_replay_steps_within_1e4"] == 20 and max(res["policy_replay_err_per_step"]) <= 1e-12
    feet = [c for c in stub.calls if c[0] == "changeDynamics" and c[1] == 1 and c[2] in (11, 19) and dict(c[3]).get("rollingFriction") == 0.1]
    assert len(feet) == 2 and all(dict(c[3]) == {"lateralFriction": 0.8, "spinningFriction": 0.1, "rollingFriction": 0.1} for c in feet)      # plen_env.py:439-456
    assert sum(1 for c in stub.calls if c[0] == "changeDynamics" and dict(c[3]).get("restitution") == 0.5 and c[1] == 1) == 32                 # plen_env.py:472-481
    t = H.time_steps(str(tmp_path), budget_s=0.3)
    assert t["value"] > 0 and t["cores"] == 1 and t["env_steps"] >= 1
    assert "disconnect" in stub.calls
