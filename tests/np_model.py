"""NumPy statement of the formulation the HIP kernels use (tests only).

The HIP path does NOT follow Bullet's articulated-body recursion; it uses
  * 19 composite bodies (fixed joints folded),
  * world-axes composite-rigid-body mass matrix M (24x24) about the base origin,
  * a classical-acceleration recursive Newton-Euler bias,
  * sparse factorization M = L^T L (leaves first: no fill-in),
  * a 48-"port" Delassus matrix A = J M^-1 J^T and projected Gauss-Seidel in port space
    visiting rows in Bullet's order.
This file states that formulation in float64 NumPy so the tests can check, on the CPU, that it is
mathematically the same thing as the C oracle's articulated-body / generalized-velocity-space
solver (oracle/plen_oracle.c), and so GPU debug dumps have something to be compared with.
"""
import json
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODEL = json.load(open(os.path.join(ROOT, "plen_ml_walk_amd/model/plen_model.json")))

NB = 19
ND = 18
NV = 24
NPORT = 48   # 18 joints + 2 feet x (3 torsional axes + 4 points x 3 directions)

DT = 1.0 / 240.0
G = np.array([0.0, 0.0, -9.81])
N_W = np.array([0.0, 0.0, 1.0])
DIR1 = np.array([0.0, -1.0, 0.0])
DIR2 = np.array([1.0, 0.0, 0.0])


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def quat_to_mat(q):
    x, y, z, w = q
    s = 2.0 / (x * x + y * y + z * z + w * w)
    return np.array([[1 - s * (y * y + z * z), s * (x * y - w * z), s * (x * z + w * y)],
                     [s * (x * y + w * z), 1 - s * (x * x + z * z), s * (y * z - w * x)],
                     [s * (x * z - w * y), s * (y * z + w * x), 1 - s * (x * x + y * y)]])


def axis_angle(a, q):
    K = skew(a)
    return np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * (K @ K)


class Body(object):
    pass


def bodies():
    out = []
    for b in MODEL["bodies"]:
        B = Body()
        B.parent = b["parent"]
        B.mass = b["mass"]
        B.com = np.array(b["com"])
        xx, yy, zz, xy, xz, yz = b["inertia"]
        B.I = np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])
        B.R = np.array(b["R"], dtype=float).reshape(3, 3)
        B.t = np.array(b["t"], dtype=float)
        B.axis = np.array(b["axis"], dtype=float)
        B.member_mass = np.array(b["member_mass"]); B.member_com = np.array(b["member_com"])
        out.append(B)
    return out


BODIES = bodies()
FOOT_BODY = [6, 12]   # right, left (DoF order: right leg 1..6, left leg 7..12)
FOOT_POINTS = [np.array(MODEL["feet"][0]["points"]), np.array(MODEL["feet"][1]["points"])]
FOOT_BREAK = [MODEL["feet"][0]["break_threshold"], MODEL["feet"][1]["break_threshold"]]
FOOT_SOLE = [np.array(MODEL["feet"][0]["sole"]), np.array(MODEL["feet"][1]["sole"])]       # 32 sole-plane hull vertices per foot
FOOT_SOLE_ORDER = [MODEL["feet"][0]["sole_order"], MODEL["feet"][1]["sole_order"]]         # per sole diagonal: vertices by descending key
FOOT_SOLE_REP = [np.array(MODEL["feet"][0]["sole_rep"], bool), np.array(MODEL["feet"][1]["sole_rep"], bool)]   # the 8 corner representatives
MARGIN = MODEL["margin"]
BOXES = MODEL["boxes"]          # box colliders of the non-foot links, pose in their composite body's frame
MAXNEAR = 8


def ancestors(b):
    """bodies on the path root..b (excluding the base), i.e. the joints that move body b"""
    out = []
    while b > 0:
        out.append(b)
        b = BODIES[b].parent
    return out[::-1]


def fk(pos, quat, q):
    """world rotation, frame origin, COM and joint axis of every composite body"""
    R = [None] * NB; O = [None] * NB; Cw = [None] * NB; A = [None] * NB
    R[0] = quat_to_mat(quat); O[0] = np.array(pos, dtype=float); A[0] = np.zeros(3)
    Cw[0] = O[0] + R[0] @ BODIES[0].com
    for b in range(1, NB):
        B = BODIES[b]; p = B.parent
        R[b] = R[p] @ B.R @ axis_angle(B.axis, q[b - 1])
        O[b] = O[p] + R[p] @ B.t
        Cw[b] = O[b] + R[b] @ B.com
        A[b] = R[b] @ B.axis
    return R, O, Cw, A


def mass_matrix_and_bias(pos, quat, omega, vel, q, qd, lin_damp=0.0):
    """M (24x24) and generalized bias force tau such that  M vdot = tau  (motors are constraints).
    Generalized velocity = [omega_world, v_base_origin_world, qd]."""
    R, O, Cw, A = fk(pos, quat, q)
    O0 = O[0]
    # per-body spatial inertia about O0, world axes: (m, m*c, Io)
    m = np.array([B.mass for B in BODIES])
    mc = np.zeros((NB, 3)); Io = np.zeros((NB, 3, 3))
    Iw = [None] * NB
    for b in range(NB):
        c = Cw[b] - O0
        Iw[b] = R[b] @ BODIES[b].I @ R[b].T
        mc[b] = m[b] * c
        Io[b] = Iw[b] + m[b] * (np.dot(c, c) * np.eye(3) - np.outer(c, c))
    # composite sums over subtrees (children have larger indices)
    cm = m.copy(); cmc = mc.copy(); cIo = Io.copy()
    for b in range(NB - 1, 0, -1):
        p = BODIES[b].parent
        cm[p] += cm[b]; cmc[p] += cmc[b]; cIo[p] += cIo[b]
    # motion subspaces about O0: base = identity; joint b: [a; (O_b - O0) x a]
    S = np.zeros((NV, 6))
    S[:6, :6] = np.eye(6)
    for b in range(1, NB):
        S[5 + b, :3] = A[b]
        S[5 + b, 3:] = np.cross(O[b] - O0, A[b])
    sub = [0] * 6 + list(range(1, NB))       # body whose composite inertia column k uses
    M = np.zeros((NV, NV))
    for k in range(NV):
        b = sub[k]
        sw, sv = S[k, :3], S[k, 3:]
        n = cIo[b] @ sw + np.cross(cmc[b], sv)
        f = cm[b] * sv + np.cross(sw, cmc[b])
        # rows: this DoF and every DoF that supports body b
        rows = list(range(6)) + [5 + a for a in ancestors(b)] if b > 0 else list(range(6))
        for r_ in rows:
            val = S[r_, :3] @ n + S[r_, 3:] @ f
            M[r_, k] = val; M[k, r_] = val
    # ---- bias: classical recursive Newton-Euler with vdot = 0 ----
    om = [None] * NB; al = [None] * NB; ao = [None] * NB; ac = [None] * NB; vc = [None] * NB
    om[0] = np.array(omega, dtype=float); al[0] = np.zeros(3); ao[0] = np.zeros(3)
    vo = [None] * NB; vo[0] = np.array(vel, dtype=float)
    for b in range(NB):
        if b > 0:
            p = BODIES[b].parent
            rel = A[b] * qd[b - 1]
            om[b] = om[p] + rel
            al[b] = al[p] + np.cross(om[p], rel)
            d = O[b] - O[p]
            vo[b] = vo[p] + np.cross(om[p], d)
            ao[b] = ao[p] + np.cross(al[p], d) + np.cross(om[p], np.cross(om[p], d))
        e = Cw[b] - O[b]
        vc[b] = vo[b] + np.cross(om[b], e)
        ac[b] = ao[b] + np.cross(al[b], e) + np.cross(om[b], np.cross(om[b], e))
    F = np.zeros((NB, 3)); N = np.zeros((NB, 3))
    for b in range(NB):
        f = m[b] * (ac[b] - G)
        n = Iw[b] @ al[b] + np.cross(om[b], Iw[b] @ om[b])
        if lin_damp > 0:           # Bullet damps every LINK: m_i v_i (k + k|v_i|) at the link COM
            for mi, ci in zip(BODIES[b].member_mass, BODIES[b].member_com):
                cw = R[b] @ ci
                vi = vo[b] + np.cross(om[b], cw)
                fi = mi * vi * (lin_damp + lin_damp * np.linalg.norm(vi))
                f = f + fi
                n = n + np.cross(O[b] + cw - Cw[b], fi)
        F[b] = f
        N[b] = n + np.cross(Cw[b] - O0, f)
    for b in range(NB - 1, 0, -1):
        p = BODIES[b].parent
        F[p] += F[b]; N[p] += N[b]
    tau = np.zeros(NV)
    tau[:3] = -N[0]; tau[3:6] = -F[0]
    for b in range(1, NB):
        tau[5 + b] = -A[b] @ (N[b] - np.cross(O[b] - O0, F[b]))
    kin = dict(R=R, O=O, C=Cw, A=A)
    return M, tau, kin


def box_candidates(kin):
    """Corners of the near boxes (at most MAXNEAR, box order) within their link's breaking threshold: (dist, id, box, body, P), deepest first."""
    R, O = kin["R"], kin["O"]
    out, near = [], 0
    for x in BOXES:
        if near >= MAXNEAR:
            break
        b = x["body"]
        ax = R[b] @ np.array(x["R_body"])              # columns: box axes in world
        c = O[b] + R[b] @ np.array(x["t_body"])
        h = np.array(x["half"])
        zmin = c[2] - (h[0] * abs(ax[2, 0]) + h[1] * abs(ax[2, 1]) + h[2] * abs(ax[2, 2]))
        if not zmin <= x["break_threshold"]:
            continue
        r = near; near += 1
        for cn in range(8):
            sg = np.array([h[0] if cn & 1 else -h[0], h[1] if cn & 2 else -h[1], h[2] if cn & 4 else -h[2]])
            w = c + ax @ sg
            if w[2] <= x["break_threshold"]:
                out.append((w[2], 8 * r + cn, x["box"], b, w))
    out.sort(key=lambda t: (t[0], t[1]))
    return out


def port_jacobians(kin, pos, body_contacts=True):
    """48 x 24 port Jacobian + bookkeeping.  Port layout:
       0..17   joint d
       18+15f + {0,1,2}            foot f torsional about (n, dir1, dir2)
       18+15f + 3 + 3k + {0,1,2}   contact slot 4f+k linear along (n, dir1, dir2): foot f's point k while that is in range, else lent to
                                   a box corner of another link (deepest first), else unused"""
    R, O, A = kin["R"], kin["O"], kin["A"]
    O0 = O[0]
    J = np.zeros((NPORT, NV))
    for d in range(ND):
        J[d, 6 + d] = 1.0
    pts = np.zeros((2, 4, 3)); dist = np.zeros((2, 4)); foot_active = np.zeros((2, 4), bool)
    for f in range(2):
        fb = FOOT_BODY[f]
        anc = ancestors(fb)
        base = 18 + 15 * f
        for a_i, ax in enumerate((N_W, DIR1, DIR2)):
            J[base + a_i, 0:3] = ax
            for b in anc:
                J[base + a_i, 5 + b] = A[b] @ ax
        # manifold of this foot: per sole diagonal the in-range sole vertex extreme along it, no vertex twice
        wv = O[fb] + FOOT_SOLE[f] @ R[fb].T
        in_range = FOOT_SOLE_REP[f] & (wv[:, 2] - MARGIN <= FOOT_BREAK[f])
        chosen = []
        for k in range(4):
            win = next((v for v in FOOT_SOLE_ORDER[f][k] if in_range[v]), None)
            if win in chosen:
                win = None
            chosen.append(win)
        packed = [w_ for w_ in chosen if w_ is not None]              # the foot's points take its lowest slots, in diagonal order
        for k in range(4):
            win = packed[k] if k < len(packed) else None
            foot_active[f, k] = win is not None
            w = wv[win] if win is not None else O[fb] + R[fb] @ FOOT_POINTS[f][k]        # (an unused slot keeps the corner point: its rows do not exist)
            dist[f, k] = w[2] - MARGIN
            P = np.array([w[0], w[1], dist[f, k]])
            pts[f, k] = P
            for a_i, ax in enumerate((N_W, DIR1, DIR2)):
                row = base + 3 + 3 * k + a_i
                J[row, 0:3] = np.cross(P - O0, ax)
                J[row, 3:6] = ax
                for b in anc:
                    J[row, 5 + b] = A[b] @ np.cross(P - O[b], ax)
    # slots whose foot point is out of range are lent to box corners, deepest first
    slots = [dict(kind="foot", f=c // 4, k=c % 4) if foot_active[c // 4, c % 4] else None for c in range(8)]
    if body_contacts:
        cands = box_candidates(kin)
        if cands and all(sl is not None for sl in slots):
            # both feet fully planted (8 foot points) AND another link near the ground: each foot gives up the slot of its fourth point
            # (a flat foot stands on three corners as well) so that the link that touches down is held up too (round 3)
            keep = {3: slots[3], 7: slots[7]}
            slots[3] = slots[7] = None
        else:
            keep = {}
        for c in range(8):
            if slots[c] is None and cands:
                d_, _, box, b, P = cands.pop(0)
                slots[c] = dict(kind="box", box=box, body=b, P=P, dist=d_)
                f, k = c // 4, c % 4
                pts[f, k] = P; dist[f, k] = d_
                for a_i, ax in enumerate((N_W, DIR1, DIR2)):
                    row = 18 + 15 * f + 3 + 3 * k + a_i
                    J[row, :] = 0
                    J[row, 0:3] = np.cross(P - O0, ax)
                    J[row, 3:6] = ax
                    for bb in ancestors(b):
                        J[row, 5 + bb] = A[bb] @ np.cross(P - O[bb], ax)
        for c, sl in keep.items():           # a released slot nobody took keeps its foot point (rows, position and distance are still the foot's)
            if slots[c] is None:
                slots[c] = sl
    return J, pts, dist, slots


class World(object):
    def __init__(self, joint_act=False):
        self.num_iterations = 50
        self.erp = 0.2; self.erp2 = 0.08
        self.linear_slop = 1e-5; self.residual_threshold = 1e-7
        self.restitution_velocity_threshold = 0.2; self.max_coordinate_velocity = 100.0
        self.box_lateral_friction = 0.5 * 0.8; self.body_contacts = True
        self.lateral_friction = 0.8 * 0.8; self.spinning_friction = 0.1 * 0.8
        self.rolling_friction = (0.01 if joint_act else 0.1) * 0.8
        self.restitution = 0.25
        self.linear_damping = 0.1 if joint_act else 0.0
        self.kp = 0.1; self.kd = 1.0; self.max_force = 0.15


NC_ORDER = [(o["kind"], o["dof"]) for o in MODEL["noncontact_order"]]
LOWER, UPPER = -1.7, 1.7


def substep(state, target, w=None, info=None):
    """state = (pos3, quat4, omega3, vel3, q18, qd18) as one 49-vector; returns the next state."""
    w = w or World()
    s = np.array(state, dtype=float)
    pos, quat, omega, vel, q, qd = s[0:3], s[3:7], s[7:10], s[10:13], s[13:31], s[31:49]
    M, tau, kin = mass_matrix_and_bias(pos, quat, omega, vel, q, qd, w.linear_damping)
    # M = L^T L (Featherstone's sparsity-preserving LTL factorization, as the kernel does it): L lower triangular
    L = np.linalg.cholesky(M[::-1, ::-1])[::-1, ::-1].T
    v = np.concatenate([omega, vel, qd])
    acc = np.linalg.solve(L, np.linalg.solve(L.T, tau))
    v = np.clip(v + DT * acc, -w.max_coordinate_velocity, w.max_coordinate_velocity)
    J, pts, dist, slots = port_jacobians(kin, pos, w.body_contacts)
    Y = np.linalg.solve(L.T, J.T)          # 24 x 48,  A = J M^-1 J^T = Y^T Y
    A = Y.T @ Y                            # port Delassus
    b = J @ v                              # port relative velocities
    diag = np.diag(A)
    jdi = 1.0 / diag
    # ---- rows ----
    lam = {}    # row id -> applied impulse
    r = np.zeros(NPORT)                    # J * deltaV per port
    active = [(c // 4, c % 4) for c in range(8) if slots[c] is not None]
    # non-contact rows in solver order
    nc_rows = []
    for kind, d in NC_ORDER:
        if kind == "motor":
            desired = w.kp * (target[d] - q[d]) / DT + v[6 + d] + w.kd * (0 - v[6 + d])
            nc_rows.append(dict(port=d, sign=1.0, rhs=(desired - b[d]) * jdi[d], lo=-w.max_force * DT, hi=w.max_force * DT, lam=0.0))
        else:
            for side, pen, sign in ((0, q[d] - LOWER, 1.0), (1, UPPER - q[d], -1.0)):
                if pen > 0:
                    continue
                pos_err = -pen * w.erp / DT if pen > -0.04 else 0.0
                nc_rows.append(dict(port=d, sign=sign, rhs=(pos_err - sign * b[d]) * jdi[d], lo=0.0, hi=100.0, lam=0.0))
    nrm, spin, roll, fric = [], [], [], []
    for (f, k) in active:
        sl = slots[4 * f + k]
        is_foot = sl["kind"] == "foot"
        mu_lat = w.lateral_friction if is_foot else w.box_lateral_friction
        restitution = w.restitution if is_foot else BOXES[sl["box"]]["link_restitution"] * 0.5
        base = 18 + 15 * f
        pn = base + 3 + 3 * k
        rel = b[pn]
        distance = dist[f, k] + w.linear_slop
        rest = 0.0 if abs(rel) < w.restitution_velocity_threshold else restitution * -rel
        rest = max(rest, 0.0)
        pos_err, vel_err = 0.0, rest - rel
        if distance > 0:
            vel_err -= distance / DT
        else:
            pos_err = -distance * w.erp2 / DT
        ni = len(nrm)
        nrm.append(dict(port=pn, sign=1.0, rhs=(pos_err + vel_err) * jdi[pn], lo=0.0, hi=1e10, lam=0.0))
        if is_foot and w.spinning_friction > 0:
            spin.append(dict(port=base, sign=1.0, rhs=-b[base] * jdi[base], mu=w.spinning_friction, n=ni, lam=0.0))
        if is_foot and w.rolling_friction > 0:
            for a_i in (1, 2):
                roll.append(dict(port=base + a_i, sign=1.0, rhs=-b[base + a_i] * jdi[base + a_i], mu=w.rolling_friction, n=ni, lam=0.0))
        for a_i in (1, 2):
            fric.append(dict(port=pn + a_i, sign=1.0, rhs=-b[pn + a_i] * jdi[pn + a_i], mu=mu_lat, n=ni, lam=0.0))

    def resolve(row):
        p = row["port"]
        di = row["rhs"] - (row["sign"] * r[p]) * jdi[p]
        sm = row["lam"] + di
        if sm < row["lo"]:
            di = row["lo"] - row["lam"]; row["lam"] = row["lo"]
        elif sm > row["hi"]:
            di = row["hi"] - row["lam"]; row["lam"] = row["hi"]
        else:
            row["lam"] = sm
        r[:] += A[:, p] * (row["sign"] * di)
        return di / jdi[p]

    its = 0
    for it in range(w.num_iterations):
        res = 0.0
        n = len(nc_rows)
        for j in range(n):
            idx = j if (it & 1) else n - 1 - j
            res = max(res, resolve(nc_rows[idx]) ** 2)
        for row in nrm:
            res = max(res, resolve(row) ** 2)
        for row in spin + roll:
            tot = nrm[row["n"]]["lam"]
            if tot > 0:
                row["lo"] = -row["mu"] * tot; row["hi"] = row["mu"] * tot
                res = max(res, resolve(row) ** 2)
        for j in range(0, len(fric), 2):
            ca, cb = fric[j], fric[j + 1]
            lim = ca["mu"] * nrm[ca["n"]]["lam"]
            pa, pb = ca["port"], cb["port"]
            dA = ca["rhs"] - r[pa] * jdi[pa]; dB = cb["rhs"] - r[pb] * jdi[pb]
            sA = ca["lam"] + dA; sB = cb["lam"] + dB
            if sA * sA + sB * sB >= lim * lim:
                ang = np.arctan2(sA, sB)
                cA_ = abs(lim * np.sin(ang)); cB_ = abs(lim * np.cos(ang))
                if sA < -cA_:
                    dA = -cA_ - ca["lam"]; ca["lam"] = -cA_
                elif sA > cA_:
                    dA = cA_ - ca["lam"]; ca["lam"] = cA_
                else:
                    ca["lam"] = sA
                if sB < -cB_:
                    dB = -cB_ - cb["lam"]; cb["lam"] = -cB_
                elif sB > cB_:
                    dB = cB_ - cb["lam"]; cb["lam"] = cB_
                else:
                    cb["lam"] = sB
            else:
                ca["lam"] = sA; cb["lam"] = sB
            r[:] += A[:, pa] * dA + A[:, pb] * dB
            res = max(res, (dA / jdi[pa] + dB / jdi[pb]) ** 2)
        its = it + 1
        if res <= w.residual_threshold or it >= w.num_iterations - 1:
            break
    # total impulse per port -> delta v = M^-1 J^T lambda = L^-1 (Y lambda)
    lam_port = np.zeros(NPORT)
    for row in nc_rows + nrm + spin + roll + fric:
        lam_port[row["port"]] += row["sign"] * row["lam"]
    dv = np.linalg.solve(L, Y @ lam_port)
    v = np.clip(v + dv, -w.max_coordinate_velocity, w.max_coordinate_velocity)
    omega, vel, qd = v[0:3], v[3:6], v[6:]
    # ---- integrate (btMultiBody::stepPositionsMultiDof) ----
    pos = pos + DT * vel
    fa = np.linalg.norm(omega)
    if fa * DT > 0.5 * (np.pi / 2):
        fa = 0.5 * (np.pi / 2) / DT
    if fa < 0.001:
        k = 0.5 * DT - DT ** 3 * 0.020833333333 * fa * fa
    else:
        k = np.sin(0.5 * fa * DT) / fa
    dq = np.array([omega[0] * k, omega[1] * k, omega[2] * k, np.cos(fa * DT * 0.5)])
    x1, y1, z1, w1 = dq; x2, y2, z2, w2 = quat
    nq = np.array([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                   w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2,
                   w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2,
                   w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2])
    nq /= np.linalg.norm(nq)
    q = q + DT * qd
    if info is not None:
        info.update(M=M, tau=tau, L=L, A=A, b=b, J=J, dist=dist, iterations=its, active=active, slots=slots,
                    right=any(sl is not None and sl["kind"] == "foot" and sl["f"] == 0 for sl in slots),
                    left=any(sl is not None and sl["kind"] == "foot" and sl["f"] == 1 for sl in slots))
    return np.concatenate([pos, nq, omega, vel, q, qd])
