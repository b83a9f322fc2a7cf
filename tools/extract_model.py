#!/usr/bin/env python3
"""Build this repo's own PLEN model tables from the reference's robot DATA files.

Inputs (read-only, only in the build container):
  /root/reference/plen_bullet/src/plen.urdf               (links :504-1274, joints :1275-1488)
  /root/reference/plen_ros/meshes_bin/{r,l}foot.stl       (the two mesh colliders, plen.urdf:1097,1263)

Outputs (committed; these are what travel to the GPU box):
  plen_ml_walk_amd/model/plen_model.json   everything below, human readable
  oracle/plen_model_raw.h                  33-link un-merged tree for the C oracle
  plen_ml_walk_amd/csrc/plen_model_gen.h   19 merged composite bodies for the HIP library

What the tables encode (SURVEY.md section 8a row A2, and DESIGN.md "Model"):
  * link order = Bullet's multibody link index = DFS over children in joint-file order,
    which reproduces movingJoints of plen_env.py:318-320 (asserted below);
  * per-link mass/COM from <inertial>; per-link inertia NOT from <inertia> but recomputed from
    the collision shape the way Bullet's URDF importer does when URDF_USE_INERTIA_FROM_FILE is
    not passed (plen_env.py:314 passes no flags): box -> m/12*(ly^2+lz^2,..) of the full extents;
    compound (collision frame != inertial frame) and convex hull -> box inertia of the compound's AABB
    (child AABB + the compound's own 1 mm margin; the hull's child AABB already includes the margin twice);
  * fixed joints folded into 19 composite bodies for the HIP path (exactly equivalent dynamics);
  * foot contact candidates: 4 sole-hull vertices per foot, extreme along the sole diagonals;
  * the order in which Bullet's solver visits the 36 non-contact constraints (18 joint limits
    then 18 motors, scrambled by btAlignedObjectArray::quickSort on all-equal island ids).
"""
import json
import os
import struct
import sys
import numpy as np
import xml.etree.ElementTree as ET
from scipy.spatial import ConvexHull

REF = os.environ.get("PLEN_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
URDF = os.path.join(REF, "plen_bullet/src/plen.urdf")
MESH_DIR = os.path.join(REF, "plen_ros/meshes_bin")

MARGIN = 0.001          # gUrdfDefaultCollisionMargin
# Hypothesis switch (round 4): btCompoundShape::getAabb grows the local AABB by the compound's OWN margin, and the URDF importer
# gives every per-link compound gUrdfDefaultCollisionMargin -- if so, the 15 links whose collision frame differs from their inertial frame get 7-17 %
# more rotational inertia (PLEN_COMPOUND_MARGIN=0.001).  The PyBullet-held pins are split on it (R_0 0.0095 -> 0.0068 and sum R_1..R_4 -7 % in its favour;
# closed-loop survival of the shipped actor at low noise 6 : 1 against), so the default stays 0: the tables of rounds 1-3.
COMPOUND_MARGIN = float(os.environ.get("PLEN_COMPOUND_MARGIN", 0.0))
BREAK_FACTOR = 0.02     # gContactBreakingThreshold / defaultContactThresholdFactor
MOVING_JOINTS_REF = [5, 6, 7, 9, 10, 11, 13, 14, 15, 17, 18, 19, 20, 21, 24, 26, 27, 30]  # plen_env.py:318-320


def rpy_mat(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def f3(s):
    return np.array([float(x) for x in s.split()], dtype=np.float64)


def load_stl_vertices(path):
    b = open(path, "rb").read()
    n = struct.unpack("<I", b[80:84])[0]
    assert len(b) == 84 + 50 * n, "binary STL expected"
    rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")])
    a = np.frombuffer(b[84:84 + 50 * n], dtype=rec)
    return a["v"].reshape(-1, 3).astype(np.float64)


def box_inertia(mass, l):
    return mass / 12.0 * np.array([l[1] ** 2 + l[2] ** 2, l[0] ** 2 + l[2] ** 2, l[0] ** 2 + l[1] ** 2])


def bullet_quicksort_equal_keys(n):
    """Permutation btAlignedObjectArray<T>::quickSort produces when every key compares equal
    (Hoare partition around the middle element, swapping on i<=j even for equal keys)."""
    a = list(range(n))

    def qs(lo, hi):
        i, j = lo, hi
        while True:
            # CompareFunc(a[i], x) and CompareFunc(x, a[j]) are always false
            if i <= j:
                a[i], a[j] = a[j], a[i]
                i += 1
                j -= 1
            if not (i <= j):
                break
        if lo < j:
            qs(lo, j)
        if i < hi:
            qs(i, hi)

    if n > 1:
        qs(0, n - 1)
    return a


def main():
    root = ET.parse(URDF).getroot()
    links_xml = {l.get("name"): l for l in root.findall("link")}
    joints_xml = root.findall("joint")
    children = {}
    child_names = set()
    for j in joints_xml:
        children.setdefault(j.find("parent").get("link"), []).append(j)
        child_names.add(j.find("child").get("link"))
    base_name = [n for n in links_xml if n not in child_names]
    assert base_name == ["torso"], base_name
    base_name = base_name[0]

    # ---- Bullet link order: DFS pre-order over children in joint-file order ----
    order = []   # (joint_xml, child_name, parent_index)
    index_of = {base_name: -1}

    def dfs(pname):
        for j in children.get(pname, []):
            c = j.find("child").get("link")
            index_of[c] = len(order)
            order.append((j, c, index_of[pname]))
            dfs(c)

    dfs(base_name)
    assert len(order) == 32

    def link_props(name):
        l = links_xml[name]
        ine = l.find("inertial")
        mass = float(ine.find("mass").get("value"))
        io = ine.find("origin")
        assert np.all(f3(io.get("rpy")) == 0), "inertial frames are axis aligned in this URDF"
        com = f3(io.get("xyz"))
        cols = l.findall("collision")
        assert len(cols) == 1
        col = cols[0]
        co = col.find("origin")
        cR = rpy_mat(*f3(co.get("rpy")))
        ct = f3(co.get("xyz"))
        g = list(col.find("geometry"))[0]
        # child transform relative to the inertial frame (inertial rotation is identity)
        rel_t = ct - com
        rel_identity = bool(np.all(rel_t == 0) and np.all(cR == np.eye(3)))
        iu = ine.find("inertia")
        assert all(float(iu.get(k)) == 0 for k in ("ixy", "ixz", "iyz")), "URDF inertia tensors are diagonal in this URDF"
        # kept only for the hypothesis ablation (scripts/pin/): what URDF_USE_INERTIA_FROM_FILE would have used
        out = dict(name=name, mass=mass, com=com.tolist(), inertia_urdf=[float(iu.get(k)) for k in ("ixx", "iyy", "izz")])
        if g.tag == "box":
            size = f3(g.get("size"))
            half = 0.5 * size
            if rel_identity:
                inertia = box_inertia(mass, size)
                aabb_half = half
                aabb_center = np.zeros(3)
            else:
                # btCompoundShape::getAabb: child AABB (btTransformAabb of implicit half + margin = the full half extents) grown by the COMPOUND's
                # own margin -- BulletURDFImporter::convertLinkCollisionShapes does compoundShape->setMargin(gUrdfDefaultCollisionMargin) and
                # getAabb adds getMargin() to the local half extents (round 4: measured on the PyBullet-held pin)
                aabb_half = np.abs(cR) @ half + COMPOUND_MARGIN
                aabb_center = rel_t
                inertia = box_inertia(mass, 2.0 * aabb_half)
            out.update(collider=dict(type="box", half=half.tolist(), R=cR.tolist(), t=ct.tolist()))
        elif g.tag == "mesh":
            fn = os.path.basename(g.get("filename"))
            sc = f3(g.get("scale"))
            v = load_stl_vertices(os.path.join(MESH_DIR, fn)) * sc
            hull = ConvexHull(v)
            hv = v[hull.vertices]
            hv_link = hv @ cR.T + ct                       # hull vertices in the link frame
            lo, hi = hv.min(0), hv.max(0)
            # btPolyhedralConvexAabbCachingShape: local AABB already carries the margin and
            # getAabb() adds it again; btCompoundShape then takes the AABB of that.
            h_local = 0.5 * (hi - lo) + 2.0 * MARGIN
            c_local = 0.5 * (hi + lo)
            aabb_half = np.abs(cR) @ h_local + COMPOUND_MARGIN
            aabb_center = cR @ c_local + rel_t
            inertia = box_inertia(mass, 2.0 * aabb_half)
            out.update(collider=dict(type="hull", mesh=fn, margin=MARGIN, verts=hv_link.tolist()))
        else:
            raise ValueError(g.tag)
        # contact breaking threshold = getAngularMotionDisc() * 0.02 (btCollisionShape)
        radius = float(np.linalg.norm(2.0 * aabb_half) * 0.5)
        disc = radius + float(np.linalg.norm(aabb_center))
        out.update(inertia=inertia.tolist(), break_threshold=disc * BREAK_FACTOR)
        return out

    base = link_props(base_name)
    links = []
    dof = 0
    moving = []
    for k, (j, cname, pidx) in enumerate(order):
        o = j.find("origin")
        R = rpy_mat(*f3(o.get("rpy")))
        t = f3(o.get("xyz"))
        lp = link_props(cname)
        jt = j.get("type")
        assert jt in ("fixed", "revolute")
        d = dict(index=k, joint=j.get("name"), parent=pidx, jtype=0 if jt == "fixed" else 1,
                 R=R.tolist(), t=t.tolist(), axis=[0.0, 0.0, 0.0], dof=-1, lower=0.0, upper=0.0, effort=0.0)
        if jt == "revolute":
            ax = f3(j.find("axis").get("xyz"))
            ax = ax / np.linalg.norm(ax)
            lim = j.find("limit")
            d.update(axis=ax.tolist(), dof=dof, lower=float(lim.get("lower")), upper=float(lim.get("upper")),
                     effort=float(lim.get("effort")))
            dof += 1
            moving.append(k)
        d.update(lp)
        links.append(d)
    assert moving == MOVING_JOINTS_REF, moving
    total_mass = base["mass"] + sum(l["mass"] for l in links)
    assert abs(total_mass - 0.495834) < 1e-9, total_mass

    # ---- zero-pose FK (link frames) for sanity numbers and sole detection ----
    Rw = {-1: np.eye(3)}
    tw = {-1: np.zeros(3)}
    for l in links:
        p = l["parent"]
        Rw[l["index"]] = Rw[p] @ np.array(l["R"])
        tw[l["index"]] = tw[p] + Rw[p] @ np.array(l["t"])
    com = base["mass"] * np.array(base["com"])
    for l in links:
        com = com + l["mass"] * (tw[l["index"]] + Rw[l["index"]] @ np.array(l["com"]))
    com /= total_mass

    # ---- foot contact candidates: 4 corner-most sole vertices ----
    feet = []
    for side, name in (("right", "r_foot"), ("left", "l_foot")):
        k = index_of[name]
        l = links[k]
        hv = np.array(l["collider"]["verts"])
        w = hv @ Rw[k].T + tw[k]
        zmin = w[:, 2].min()
        sole_idx = np.where(w[:, 2] < zmin + 1e-6)[0]
        sole = w[sole_idx]
        # Bullet keeps at most 4 manifold points per collider pair and prefers the set spanning the
        # largest area around the deepest point.  Our deterministic stand-in: the four sole-hull
        # vertices that are extreme along the sole rectangle's diagonals (support points of the
        # directions (+-x/hx, +-y/hy) in the sole plane) -> a corner-aligned quadrilateral.
        ctr = 0.5 * (sole[:, :2].min(0) + sole[:, :2].max(0))
        hxy = 0.5 * (sole[:, :2].max(0) - sole[:, :2].min(0))
        nrm = (sole[:, :2] - ctr) / hxy
        pts_idx = []
        for sx, sy in ((-1, -1), (-1, 1), (1, -1), (1, 1)):   # ordered by world (x, y) at zero pose
            score = sx * nrm[:, 0] + sy * nrm[:, 1]
            # tie-break on the larger |x| so that the choice is unique and mirror-symmetric
            top = np.where(score > score.max() - 1e-9)[0]
            pick = top[np.argmax(np.abs(nrm[top, 0]))]
            pts_idx.append(int(sole_idx[pick]))
        pts_local = hv[pts_idx]
        # Round 2: ALL sole-hull vertices are contact candidates; per collision pass the ones within the breaking threshold are reduced to
        # <= 4 manifold points: for each of the four sole diagonals the in-range vertex that is extreme along it (Bullet's manifold keeps the
        # set spanning the largest area).  sole_order[k][j] = vertex with the j-th highest key along diagonal k (ties: larger |x|, then lower
        # index), so "first in-range entry of sole_order[k]" is the winner; with the whole sole in range the winners are pts_idx above.
        order2d = ConvexHull(sole[:, :2]).vertices            # counter-clockwise outline
        assert len(order2d) == len(sole_idx) == 32, "the HIP path assumes 32 sole vertices per foot (one lane each)"
        sole_local = hv[sole_idx[order2d]]
        nrm2 = nrm[order2d]
        # The outline is an octagon whose 8 corners are 4-vertex fillets ~1.5 mm long.  Bullet's manifold merges points closer than the
        # breaking threshold (~1 mm, btPersistentManifold::getCacheEntry), so a fillet is ONE contact location: its corner-most vertex
        # represents it (sole_rep = 1) and only the 8 representatives are candidates (one tilted corner = one point, one edge = two).
        xy = sole[order2d][:, :2]
        gap = np.linalg.norm(xy - np.roll(xy, 1, axis=0), axis=1) > 1.2e-3          # True where a new fillet starts
        start = int(np.where(gap)[0][0])
        groups, cur = [], []
        for i in range(32):
            v = (start + i) % 32
            if gap[v] and cur:
                groups.append(cur); cur = []
            cur.append(v)
        groups.append(cur)
        assert len(groups) == 8 and all(len(g) == 4 for g in groups), [len(g) for g in groups]
        sole_rep = [0] * 32
        for g in groups:
            sole_rep[max(g, key=lambda v: (round(float(abs(nrm2[v, 0]) + abs(nrm2[v, 1])), 9), -v))] = 1
        sole_order = []
        for kdir, (sx, sy) in enumerate(((-1, -1), (-1, 1), (1, -1), (1, 1))):
            score = sx * nrm2[:, 0] + sy * nrm2[:, 1]
            o = sorted(range(32), key=lambda v: (-round(float(score[v]), 9), -abs(float(nrm2[v, 0])), v))
            o = [v for v in o if sole_rep[v]] + [v for v in o if not sole_rep[v]]        # representatives first (the others never come into range)
            assert np.allclose(sole_local[o[0]], pts_local[kdir]), "flat stance must reproduce the round-1 corner points"
            sole_order.append([int(v) for v in o])
        q = w[pts_idx][[0, 1, 3, 2], :2]
        best_area = 0.5 * abs(np.dot(q[:, 0], np.roll(q[:, 1], -1)) - np.dot(q[:, 1], np.roll(q[:, 0], -1)))
        feet.append(dict(side=side, link=k, n_hull=len(hv), n_sole=len(sole_idx),
                         sole_z_at_spawn=float(zmin + 0.158), quad_area=float(best_area),
                         sole_polygon_area=float(ConvexHull(sole[:, :2]).volume), sole_bbox=(2 * hxy).tolist(),
                         points=pts_local.tolist(), sole=sole_local.tolist(), sole_order=sole_order, sole_rep=sole_rep,
                         margin=MARGIN, break_threshold=l["break_threshold"]))

    # ---- merged composite bodies (fixed joints folded) ----
    # body 0 = base composite; bodies 1..18 in DoF order; frame of body b = link frame of its moving link
    body_of_link = {-1: 0}
    for l in links:
        if l["jtype"] == 1:
            body_of_link[l["index"]] = l["dof"] + 1
        else:
            body_of_link[l["index"]] = body_of_link[l["parent"]]
    # transform of every link frame relative to its body's frame
    Rb = {-1: np.eye(3)}
    tb = {-1: np.zeros(3)}
    for l in links:
        if l["jtype"] == 1:
            Rb[l["index"]] = np.eye(3)
            tb[l["index"]] = np.zeros(3)
        else:
            p = l["parent"]
            Rb[l["index"]] = Rb[p] @ np.array(l["R"])
            tb[l["index"]] = tb[p] + Rb[p] @ np.array(l["t"])
    bodies = []
    for b in range(19):
        members = [(-1, base)] if b == 0 else []
        members += [(l["index"], l) for l in links if body_of_link[l["index"]] == b]
        m = sum(x["mass"] for _, x in members)
        c = sum(x["mass"] * (tb[i] + Rb[i] @ np.array(x["com"])) for i, x in members) / m
        I = np.zeros((3, 3))
        for i, x in members:
            ci = tb[i] + Rb[i] @ np.array(x["com"])
            d = ci - c
            I += Rb[i] @ np.diag(x["inertia"]) @ Rb[i].T + x["mass"] * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        if b == 0:
            bd = dict(body=0, parent=-1, link=-1, R=np.eye(3).tolist(), t=[0, 0, 0], axis=[0, 0, 0])
        else:
            ml = links[moving[b - 1]]
            # joint placement relative to the parent BODY frame (through any fixed links in between)
            p = ml["parent"]
            Rj = Rb[p] @ np.array(ml["R"])
            tj = tb[p] + Rb[p] @ np.array(ml["t"])
            bd = dict(body=b, parent=body_of_link[p], link=ml["index"], R=Rj.tolist(), t=tj.tolist(), axis=ml["axis"])
        bd.update(mass=m, com=c.tolist(), inertia=[I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]],
                  members=[x["name"] for _, x in members],
                  member_mass=[x["mass"] for _, x in members],
                  member_com=[(tb[i] + Rb[i] @ np.array(x["com"])).tolist() for i, x in members])
        bodies.append(bd)
    assert abs(sum(b["mass"] for b in bodies) - total_mass) < 1e-12

    # ---- ground-contact boxes of the non-foot links (plen.urdf:504-1274: 31 box colliders; the plane is loaded at plen_env.py:306-309) ----
    # box 0 = the base link (torso), then the non-foot links in link order.  Pose in the LINK frame for the oracle and in the composite BODY
    # frame for the HIP library.  Link restitution: 0.5 for links 0..31 (changeDynamics loop plen_env.py:472-481), the base keeps Bullet's
    # default 0; lateral friction: Bullet's URDF default 0.5 (only the feet are changed, :439-467); no spinning/rolling friction.
    boxes = []
    for idx, x in [(-1, base)] + [(l["index"], l) for l in links]:
        c = x["collider"]
        if c["type"] != "box":
            continue
        cR, ct = np.array(c["R"]), np.array(c["t"])
        boxes.append(dict(box=len(boxes), link=idx, name=x["name"], body=body_of_link[idx], half=c["half"], R_link=cR.tolist(), t_link=ct.tolist(),
                          R_body=(Rb[idx] @ cR).tolist(), t_body=(tb[idx] + Rb[idx] @ ct).tolist(), break_threshold=x["break_threshold"],
                          link_restitution=0.0 if idx < 0 else 0.5, link_lateral_friction=0.5))
    assert len(boxes) == 31

    # ---- non-contact constraint visiting order ----
    # world array: 18 limit constraints (DoF order) then 18 motors (DoF order); ids 0..17 limit, 18..35 motor
    perm = bullet_quicksort_equal_keys(36)
    noncontact_order = [dict(kind="limit" if p < 18 else "motor", dof=p % 18) for p in perm]

    model = dict(
        source=dict(urdf="plen_bullet/src/plen.urdf", meshes=["plen_ros/meshes_bin/rfoot.stl", "plen_ros/meshes_bin/lfoot.stl"],
                    note="generated by tools/extract_model.py; do not edit"),
        total_mass=total_mass, zero_pose_com_at_spawn=(com + np.array([0, 0, 0.158])).tolist(),
        moving_joints=moving, base=base, links=links, feet=feet, bodies=bodies, boxes=boxes,
        noncontact_order=noncontact_order, margin=MARGIN)

    # PLEN_MODEL_OUT=<dir>: write the three tables there instead of over the committed ones (model-table hypotheses, scripts/pin/margin_pooled.py)
    alt = os.environ.get("PLEN_MODEL_OUT")
    out_json = os.path.join(alt, "plen_model.json") if alt else os.path.join(ROOT, "plen_ml_walk_amd/model/plen_model.json")
    with open(out_json, "w") as f:
        json.dump(model, f, indent=1)
    write_raw_header(model, os.path.join(alt, "plen_model_raw.h") if alt else os.path.join(ROOT, "oracle/plen_model_raw.h"))
    write_merged_header(model, os.path.join(alt, "plen_model_gen.h") if alt else os.path.join(ROOT, "plen_ml_walk_amd/csrc/plen_model_gen.h"))
    print("total mass", total_mass)
    print("zero-pose COM at spawn", model["zero_pose_com_at_spawn"])
    for ft in feet:
        print(ft["side"], "sole z at spawn", ft["sole_z_at_spawn"], "quad/polygon area",
              ft["quad_area"], ft["sole_polygon_area"], "break thr", ft["break_threshold"])
        print("   points", np.round(np.array(ft["points"]) * 1000, 3).tolist())
    print("non-contact order", [("L" if o["kind"] == "limit" else "M") + str(o["dof"]) for o in noncontact_order])


def carr(vals, per_line=6, nested=None):
    """C initializer body; with nested=k every k values are wrapped in braces (rows of a 2-D array)."""
    if nested:
        rows = ["  {" + ", ".join(repr(float(v)) for v in vals[i:i + nested]) + "}" for i in range(0, len(vals), nested)]
        return ",\n".join(rows)
    s = []
    for i in range(0, len(vals), per_line):
        s.append("  " + ", ".join(repr(float(v)) for v in vals[i:i + per_line]))
    return ",\n".join(s)


def write_raw_header(m, path):
    L = m["links"]
    with open(path, "w") as f:
        f.write("/* GENERATED by tools/extract_model.py from the reference's plen.urdf + foot STLs (data only).\n"
                " * Un-merged 33-link tree in Bullet link order for the C oracle. Do not edit. */\n"
                "#ifndef PLEN_MODEL_RAW_H\n#define PLEN_MODEL_RAW_H\n")
        f.write("#define RAW_NLINKS 32\n#define RAW_NDOF 18\n")
        f.write("static const double RAW_BASE_MASS = %r;\n" % m["base"]["mass"])
        f.write("static const double RAW_BASE_COM[3] = {%s};\n" % ", ".join(repr(x) for x in m["base"]["com"]))
        f.write("static const double RAW_BASE_INERTIA[3] = {%s};\n" % ", ".join(repr(x) for x in m["base"]["inertia"]))
        f.write("static const int RAW_PARENT[32] = {%s};\n" % ", ".join(str(l["parent"]) for l in L))
        f.write("static const int RAW_JTYPE[32] = {%s};\n" % ", ".join(str(l["jtype"]) for l in L))
        f.write("static const int RAW_DOF[32] = {%s};\n" % ", ".join(str(l["dof"]) for l in L))
        f.write("static const double RAW_MASS[32] = {\n%s};\n" % carr([l["mass"] for l in L]))
        f.write("static const double RAW_COM[32][3] = {\n%s};\n" % carr([x for l in L for x in l["com"]], nested=3))
        f.write("static const double RAW_INERTIA[32][3] = {\n%s};\n" % carr([x for l in L for x in l["inertia"]], nested=3))
        f.write("static const double RAW_JR[32][9] = {\n%s};\n" % carr([x for l in L for r in l["R"] for x in r], nested=9))
        f.write("static const double RAW_JT[32][3] = {\n%s};\n" % carr([x for l in L for x in l["t"]], nested=3))
        f.write("static const double RAW_AXIS[32][3] = {\n%s};\n" % carr([x for l in L for x in l["axis"]], nested=3))
        f.write("static const double RAW_LOWER[32] = {\n%s};\n" % carr([l["lower"] for l in L]))
        f.write("static const double RAW_UPPER[32] = {\n%s};\n" % carr([l["upper"] for l in L]))
        f.write("static const int RAW_MOVING[18] = {%s};\n" % ", ".join(str(x) for x in m["moving_joints"]))
        for ft in m["feet"]:
            tag = "RFOOT" if ft["side"] == "right" else "LFOOT"
            hv = L[ft["link"]]["collider"]["verts"]
            f.write("#define RAW_%s_LINK %d\n#define RAW_%s_NHULL %d\n" % (tag, ft["link"], tag, len(hv)))
            f.write("static const double RAW_%s_HULL[%d][3] = {\n%s};\n" % (tag, len(hv), carr([x for v in hv for x in v], nested=3)))
            f.write("static const double RAW_%s_POINTS[4][3] = {\n%s};\n" % (tag, carr([x for v in ft["points"] for x in v], nested=3)))
            f.write("static const double RAW_%s_BREAK = %r;\n" % (tag, ft["break_threshold"]))
            f.write("/* the 32 sole-plane hull vertices (link frame, outline order) and, per sole diagonal, the vertices by descending key */\n")
            f.write("static const double RAW_%s_SOLE[32][3] = {\n%s};\n" % (tag, carr([x for v in ft["sole"] for x in v], nested=3)))
            f.write("static const int RAW_%s_SOLE_ORDER[4][32] = {%s};\n" % (tag, ", ".join("{" + ", ".join(str(v) for v in o) + "}" for o in ft["sole_order"])))
            f.write("/* 1: the vertex represents its corner fillet and is a contact candidate */\n")
            f.write("static const int RAW_%s_SOLE_REP[32] = {%s};\n" % (tag, ", ".join(str(v) for v in ft["sole_rep"])))
        f.write("static const double RAW_MARGIN = %r;\n" % m["margin"])
        X = m["boxes"]
        f.write("/* box colliders of the non-foot links (pose in the LINK frame; link -1 = base); half extents already include Bullet's margin */\n")
        f.write("#define RAW_NBOX %d\n" % len(X))
        f.write("static const int RAW_BOX_LINK[%d] = {%s};\n" % (len(X), ", ".join(str(b["link"]) for b in X)))
        f.write("static const double RAW_BOX_R[%d][9] = {\n%s};\n" % (len(X), carr([v for b in X for r in b["R_link"] for v in r], nested=9)))
        f.write("static const double RAW_BOX_T[%d][3] = {\n%s};\n" % (len(X), carr([v for b in X for v in b["t_link"]], nested=3)))
        f.write("static const double RAW_BOX_H[%d][3] = {\n%s};\n" % (len(X), carr([v for b in X for v in b["half"]], nested=3)))
        f.write("static const double RAW_BOX_BREAK[%d] = {\n%s};\n" % (len(X), carr([b["break_threshold"] for b in X])))
        f.write("static const double RAW_BOX_LINK_RESTITUTION[%d] = {\n%s};\n" % (len(X), carr([b["link_restitution"] for b in X])))
        f.write("/* solver visiting order of the 36 non-contact constraints: kind 0=limit 1=motor, dof */\n")
        f.write("static const int RAW_NC_KIND[36] = {%s};\n" % ", ".join("0" if o["kind"] == "limit" else "1" for o in m["noncontact_order"]))
        f.write("static const int RAW_NC_DOF[36] = {%s};\n" % ", ".join(str(o["dof"]) for o in m["noncontact_order"]))
        f.write("#endif\n")


def write_merged_header(m, path):
    B = m["bodies"]
    with open(path, "w") as f:
        f.write("/* GENERATED by tools/extract_model.py from the reference's plen.urdf + foot STLs (data only).\n"
                " * 19 composite bodies (fixed joints folded), DoF order, for the HIP library. Do not edit. */\n"
                "#ifndef PLEN_MODEL_GEN_H\n#define PLEN_MODEL_GEN_H\n")
        f.write("#define GEN_NBODY 19\n")
        f.write("static const int GEN_PARENT[19] = {%s};\n" % ", ".join(str(b["parent"]) for b in B))
        f.write("static const double GEN_MASS[19] = {\n%s};\n" % carr([b["mass"] for b in B]))
        f.write("static const double GEN_COM[19][3] = {\n%s};\n" % carr([x for b in B for x in b["com"]], nested=3))
        f.write("/* xx yy zz xy xz yz about the composite COM, body-frame axes */\n")
        f.write("static const double GEN_INERTIA[19][6] = {\n%s};\n" % carr([x for b in B for x in b["inertia"]], nested=6))
        f.write("/* joint placement in the parent body frame (row-major R) and axis in the child frame */\n")
        f.write("static const double GEN_JR[19][9] = {\n%s};\n" % carr([x for b in B for r in b["R"] for x in r], nested=9))
        f.write("static const double GEN_JT[19][3] = {\n%s};\n" % carr([x for b in B for x in b["t"]], nested=3))
        f.write("static const double GEN_AXIS[19][3] = {\n%s};\n" % carr([x for b in B for x in b["axis"]], nested=3))
        for ft in m["feet"]:
            tag = "RFOOT" if ft["side"] == "right" else "LFOOT"
            body = [b["body"] for b in B if b["link"] == ft["link"]][0]
            f.write("#define GEN_%s_BODY %d\n" % (tag, body))
            f.write("static const double GEN_%s_POINTS[4][3] = {\n%s};\n" % (tag, carr([x for v in ft["points"] for x in v], nested=3)))
            f.write("static const double GEN_%s_BREAK = %r;\n" % (tag, ft["break_threshold"]))
            f.write("/* the 32 sole-plane hull vertices (foot body frame, outline order) and, per sole diagonal, the vertices by descending key */\n")
            f.write("static const double GEN_%s_SOLE[32][3] = {\n%s};\n" % (tag, carr([x for v in ft["sole"] for x in v], nested=3)))
            f.write("static const int GEN_%s_SOLE_ORDER[4][32] = {%s};\n" % (tag, ", ".join("{" + ", ".join(str(v) for v in o) + "}" for o in ft["sole_order"])))
            f.write("/* 1: the vertex represents its corner fillet and is a contact candidate */\n")
            f.write("static const int GEN_%s_SOLE_REP[32] = {%s};\n" % (tag, ", ".join(str(v) for v in ft["sole_rep"])))
        nmax = max(len(b["member_mass"]) for b in B)
        f.write("/* member links of every composite body (mass, COM in the body frame): Bullet applies its linear damping per LINK */\n")
        f.write("#define GEN_MAXMEMB %d\n" % nmax)
        f.write("static const int GEN_NMEMB[19] = {%s};\n" % ", ".join(str(len(b["member_mass"])) for b in B))
        f.write("static const double GEN_MEMB_MASS[19][%d] = {\n%s};\n" % (nmax, carr([x for b in B for x in (b["member_mass"] + [0.0] * nmax)[:nmax]], nested=nmax)))
        f.write("static const double GEN_MEMB_COM[19][%d] = {\n%s};\n" % (nmax * 3, carr([x for b in B for v in (b["member_com"] + [[0.0, 0.0, 0.0]] * nmax)[:nmax] for x in v], nested=nmax * 3)))
        f.write("static const double GEN_MARGIN = %r;\n" % m["margin"])
        X = m["boxes"]
        f.write("/* box colliders of the non-foot links: composite body, pose in that BODY's frame (row-major R), half extents (margin included),\n"
                " * contact breaking threshold, link restitution (0.5; base 0) */\n")
        f.write("#define GEN_NBOX %d\n" % len(X))
        f.write("static const int GEN_BOX_BODY[%d] = {%s};\n" % (len(X), ", ".join(str(b["body"]) for b in X)))
        f.write("static const double GEN_BOX_R[%d][9] = {\n%s};\n" % (len(X), carr([v for b in X for r in b["R_body"] for v in r], nested=9)))
        f.write("static const double GEN_BOX_T[%d][3] = {\n%s};\n" % (len(X), carr([v for b in X for v in b["t_body"]], nested=3)))
        f.write("static const double GEN_BOX_H[%d][3] = {\n%s};\n" % (len(X), carr([v for b in X for v in b["half"]], nested=3)))
        f.write("static const double GEN_BOX_BREAK[%d] = {\n%s};\n" % (len(X), carr([b["break_threshold"] for b in X])))
        f.write("static const double GEN_BOX_LINK_RESTITUTION[%d] = {\n%s};\n" % (len(X), carr([b["link_restitution"] for b in X])))
        lows = set(l["lower"] for l in m["links"] if l["jtype"] == 1); ups = set(l["upper"] for l in m["links"] if l["jtype"] == 1)
        assert len(lows) == 1 and len(ups) == 1, "the HIP path assumes one common joint limit"
        f.write("#define GEN_LOWER_LIMIT %r\n#define GEN_UPPER_LIMIT %r\n" % (lows.pop(), ups.pop()))
        f.write("static const int GEN_NC_KIND[36] = {%s};\n" % ", ".join("0" if o["kind"] == "limit" else "1" for o in m["noncontact_order"]))
        f.write("static const int GEN_NC_DOF[36] = {%s};\n" % ", ".join(str(o["dof"]) for o in m["noncontact_order"]))
        f.write("#endif\n")


if __name__ == "__main__":
    main()
