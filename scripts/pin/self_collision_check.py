#!/usr/bin/env python3
"""Would the reference's 457 setCollisionFilterPair(..., enableCollision=1) calls (plen_env.py:355-434) matter if they DID switch self-collision on?
The pair list is rebuilt from the loops of that code (movingJoints and the link count are data of the model); for every listed pair of links the
oriented collision boxes (feet: the hull's bounding box) are tested for overlap (separating-axis test) at the reset pose and along the first control steps
of the recorded command log.  If listed pairs interpenetrate at the very pose every episode starts from, an engine that honoured the calls would push those
links apart during reset()'s 8 settle substeps -- and the reset observation could not agree with this simulator's (which has no self-collision) to the
~1e-4 the pin measures (tests/pybullet_pin.py).  Prints the overlapping listed pairs; writes profiles/r03_self_collision_check.json."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.oracle import OracleEnv
import pybullet_pin as P

M = json.load(open(os.path.join(ROOT, "plen_ml_walk_amd", "model", "plen_model.json")))
moving = M["moving_joints"]; numj = 32


def listed_pairs():
    pairs = []
    for j in range(6):
        for cj in range(6, numj):
            if moving[j] != cj and cj != 4: pairs.append((moving[j], cj))
    for j in range(6, 12):
        for cj in range(12, numj):
            if moving[j] != cj and cj != 12: pairs.append((moving[j], cj))
        for cj in range(6):
            if moving[j] != cj and cj != 12: pairs.append((moving[j], cj))
    for j in range(12, 15):
        for cj in range(15, numj):
            if moving[j] != cj and cj != 2: pairs.append((moving[j], cj))
        for cj in range(12):
            if moving[j] != cj and cj != 2: pairs.append((moving[j], cj))
    for j in range(15, 18):
        for cj in range(18, numj):
            if moving[j] != cj and cj != 3: pairs.append((moving[j], cj))
        for cj in range(15):
            if moving[j] != cj and cj != 3: pairs.append((moving[j], cj))
    return pairs


def link_box(l):
    c = l["collider"]
    if c["type"] == "box":
        return np.array(c["R"]), np.array(c["t"]), np.array(c["half"])
    v = np.array(c["verts"]); lo, hi = v.min(0), v.max(0)          # hull: bounding box in the link frame
    return np.eye(3), 0.5 * (lo + hi), 0.5 * (hi - lo)


def obb_overlap(Ca, Ra, ha, Cb, Rb, hb):
    """Separating-axis test of two oriented boxes; returns the penetration depth along the best axis (<= 0: separated)."""
    axes = [Ra[:, i] for i in range(3)] + [Rb[:, i] for i in range(3)]
    for i in range(3):
        for j in range(3):
            a = np.cross(Ra[:, i], Rb[:, j]); n = np.linalg.norm(a)
            if n > 1e-9: axes.append(a / n)
    d = Cb - Ca; pen = np.inf
    for a in axes:
        ra = sum(ha[i] * abs(a @ Ra[:, i]) for i in range(3)); rb = sum(hb[i] * abs(a @ Rb[:, i]) for i in range(3))
        pen = min(pen, ra + rb - abs(a @ d))
    return pen


def overlaps(e, pairs):
    R, O, _ = e.link_frames()
    out = []
    for a, b in pairs:
        la, lb = M["links"][a], M["links"][b]
        Ra_, ta, ha = link_box(la); Rb_, tb, hb = link_box(lb)
        Ca = O[a + 1] + R[a + 1] @ ta; Cb = O[b + 1] + R[b + 1] @ tb
        pen = obb_overlap(Ca, R[a + 1] @ Ra_, ha, Cb, R[b + 1] @ Rb_, hb)
        if pen > 0: out.append((la["name"], lb["name"], float(pen), bool(la["parent"] == b or lb["parent"] == a)))
    return out


if __name__ == "__main__":
    pairs = listed_pairs()
    assert len(pairs) == 457, len(pairs)           # what the reference prints ("COLLISION BETWEEN LINKS" lines)
    e = OracleEnv(); e.reset()
    res = {"listed_pairs": len(pairs), "steps": []}
    for t in range(9):
        ov = overlaps(e, pairs)
        res["steps"].append({"after_control_steps": t, "overlapping_listed_pairs": len(ov), "parent_child_among_them": sum(1 for o in ov if o[3]),
                             "deepest": sorted(ov, key=lambda o: -o[2])[:6]})
        print("after %d control steps: %3d of 457 listed pairs overlap (%d parent-child), deepest %s" % (
            t, len(ov), sum(1 for o in ov if o[3]), ", ".join("%s/%s %.1f mm" % (o[0], o[1], o[2] * 1e3) for o in sorted(ov, key=lambda o: -o[2])[:4])))
        e.step(P.ACTS[t].astype(np.float64))
    res["pin_R0"] = P.oracle_residuals(K=0)[0][0]
    res["conclusion"] = ("listed pairs interpenetrate by millimetres at the reset pose itself; an engine honouring the calls would separate them during reset()'s settle "
                         "substeps, yet the reset observation agrees with PyBullet's (pin R_0, observation errors ~1e-4) WITHOUT self-collision: the calls have no effect, "
                         "as btMultiBodyLinkCollider::checkCollideWithOverride predicts for a body loaded without URDF_USE_SELF_COLLISION")
    json.dump(res, open(os.path.join(ROOT, "profiles", "r03_self_collision_check.json"), "w"), indent=1)
