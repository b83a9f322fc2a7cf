/*
 * plen_oracle.c -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * A plain-C restatement of the algorithm the reference's hot path runs:
 *   plen_bullet/src/plen_bullet/plen_env.py  PlenWalkEnv.step  (:638-692), reset (:558-614),
 *   agent_to_env (:694-714), move_joints (:716-753), compute_observation (:768-871),
 *   compute_reward (:873-1070), compute_done (:1072-1093),
 * including what `p.stepSimulation()` (:667) does inside the third-party `pybullet` module.
 *
 * PHYSICS PARITY: PARTIAL.  pybullet / Bullet is NOT vendored by the reference, NOT version pinned (no requirements file) and NOT
 * installed on any machine this was built or run on, and the reference has no tests or golden vectors for it, so the physics below cannot
 * be compared with PyBullet output component by component.  What pins it (round 3, tests/pybullet_pin.py, tests/test_pybullet_pin.py): the
 * reference's recorded command log trajectories/<joint>_cmd.npy is the shipped actor's deterministic output along a PyBullet episode,
 * a_t = actor_3229999(obs_t^PyBullet), i.e. 500 x 18 equations on PyBullet's own observations: this oracle's reset observation satisfies
 * them to observation errors of ~1e-4 and its observation after one full-range control step to a few 1e-3.  Of the Bullet defaults assumed
 * below, the iteration count, the row order and erp2 are sharp optima of the reset residual R_0; the motor gains, joint damping, maximum
 * force and velocity clamp are NOT (nominal R_0 / R_1 prefer 5-10 % more damping, the robust score and the stance pin do not), and no variant
 * of any of them wins on the chaos-robust objectives (round 4: profiles/r04_ablation.json, r04_ablation_pooled.json; round 5, the compound
 * margin in the link inertias: profiles/r05_margin_pooled.json) -- the documented values stay.  The physics
 * restates Bullet's published multibody algorithm (era ~2.89, early 2020) as documented in DESIGN.md section "Oracle":
 *   btMultiBody::computeAccelerationsArticulatedBodyAlgorithmMultiDof  (Featherstone ABA, explicit
 *       gyroscopic term, semi-implicit Euler: v += dt*a before the constraint solve),
 *   btMultiBody::calcAccelerationDeltasMultiDof                         (unit-impulse response),
 *   btMultiBodyJointMotor / btMultiBodyJointLimitConstraint::createConstraintRows,
 *   btMultiBodyConstraintSolver::setupMultiBodyContactConstraint / ...TorsionalFriction...,
 *   btMultiBodyConstraintSolver::solveSingleIteration / resolveSingleConstraintRowGeneric /
 *       resolveConeFrictionConstraintRows, residual early-out,
 *   btMultiBody::stepPositionsMultiDof (exponential-map quaternion update),
 *   pybullet.c getEulerFromQuaternion,
 * with PyBullet's world defaults (50 iterations, erp2 0.08, linearSlop 1e-5, residual 1e-7, ...).
 * The Python-level arithmetic (action map, reward, termination, gait bookkeeping) IS pinned: it is
 * checked against golden vectors captured from the reference itself (tests/golden/, made by
 * tools/make_golden.py) in tests/test_oracle_golden.py.
 *
 * Known, documented deviation: contact generation.  Bullet runs GJK/EPA foot-hull vs ground-box
 * with a persistent 4-point manifold whose point positions depend on solver history; that is not
 * reproducible without Bullet itself.  Here the candidates of a foot are the 8 corners of its
 * octagonal sole outline (one representative hull vertex per rounded corner, 1 mm spherical
 * margin) tested against the half-space z<=0; those within the foot's
 * contact-breaking threshold are reduced to <= 4 manifold points per collision pass: per sole
 * diagonal the in-range vertex extreme along it (area-spanning like btPersistentManifold's
 * reduction, but memoryless); a flat foot yields its four corner-most vertices.
 * The 31 box colliders of the other links (plen.urdf:504-1274) collide with the ground too (Bullet:
 * btBoxBoxDetector against plane.urdf's box -> the penetrating corners of the face turned to the
 * ground): a box corner is a contact point while its height is below the link's breaking threshold.
 * The solver has 8 contact-point slots (the HIP kernel's port layout): a foot's points take its lowest
 * slots in diagonal order; box corners, deepest first, take the slots whose foot point is inactive; corners
 * beyond that are dropped (both feet flat AND other links on the ground: see DESIGN.md).
 * Self-collision is off in the reference.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Build:  make -C oracle     (gcc -O2 -fPIC -shared; -DORACLE_REAL=float for the f32 variant; `make native` =
 *         -O3 -march=native -fopenmp on the machine that times it, bench.py's cpu_baseline)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "plen_model_raw.h"

#ifndef ORACLE_REAL
#define ORACLE_REAL double
#endif
typedef ORACLE_REAL real;

#define NL RAW_NLINKS          /* 32 non-base links                               */
#define ND RAW_NDOF            /* 18 joint DoF                                    */
#define NV (6 + ND)            /* generalized velocity: base omega(3), v(3), qd   */
#define MAXCP 8                /* contact-point slots: 2 feet x 4 home points; a slot whose foot point is inactive may be lent to a box corner */
#define MAXNEAR 8              /* at most this many near boxes (box index order) are enumerated per collision pass */
#define MAXROWS (36 + MAXCP * 6)
#define HIST 1024

/* ---------------------------------------------------------------- small math helpers */
static inline void v3set(real *a, real x, real y, real z) { a[0] = x; a[1] = y; a[2] = z; }
static inline void v3cpy(real *a, const real *b) { a[0] = b[0]; a[1] = b[1]; a[2] = b[2]; }
static inline real v3dot(const real *a, const real *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void v3cross(real *c, const real *a, const real *b) {
    real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    c[0] = x; c[1] = y; c[2] = z;
}
static inline void v3sub(real *c, const real *a, const real *b) { c[0] = a[0] - b[0]; c[1] = a[1] - b[1]; c[2] = a[2] - b[2]; }
static inline void v3add(real *c, const real *a, const real *b) { c[0] = a[0] + b[0]; c[1] = a[1] + b[1]; c[2] = a[2] + b[2]; }
static inline void v3axpy(real *y, real a, const real *x) { y[0] += a * x[0]; y[1] += a * x[1]; y[2] += a * x[2]; }
static inline real v3norm(const real *a) { return (real)sqrt((double)v3dot(a, a)); }
static inline void m3mulv(real *o, const real *M, const real *v) {
    real x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
    real y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
    real z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3mul(real *o, const real *A, const real *B) {
    real t[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
        t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(o, t, sizeof t);
}
/* rotation about a unit axis by angle (Rodrigues) */
static void axis_angle(real *R, const real *a, real q) {
    real c = (real)cos((double)q), s = (real)sin((double)q), t = 1 - c;
    R[0] = c + a[0] * a[0] * t;        R[1] = a[0] * a[1] * t - a[2] * s; R[2] = a[0] * a[2] * t + a[1] * s;
    R[3] = a[1] * a[0] * t + a[2] * s; R[4] = c + a[1] * a[1] * t;        R[5] = a[1] * a[2] * t - a[0] * s;
    R[6] = a[2] * a[0] * t - a[1] * s; R[7] = a[2] * a[1] * t + a[0] * s; R[8] = c + a[2] * a[2] * t;
}
/* quaternion (x,y,z,w), body->world */
static void quat_to_mat(real *R, const real *q) {
    real x = q[0], y = q[1], z = q[2], w = q[3];
    real d = x * x + y * y + z * z + w * w, s = 2 / d;
    real xs = x * s, ys = y * s, zs = z * s;
    real wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
    R[0] = 1 - (yy + zz); R[1] = xy - wz;       R[2] = xz + wy;
    R[3] = xy + wz;       R[4] = 1 - (xx + zz); R[5] = yz - wx;
    R[6] = xz - wy;       R[7] = yz + wx;       R[8] = 1 - (xx + yy);
}
static void mat_to_quat(real *q, const real *R) {   /* btMatrix3x3::getRotation */
    real tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        real s = (real)sqrt((double)(tr + 1));
        q[3] = s * (real)0.5; s = (real)0.5 / s;
        q[0] = (R[7] - R[5]) * s; q[1] = (R[2] - R[6]) * s; q[2] = (R[3] - R[1]) * s;
    } else {
        int i = R[0] < R[4] ? (R[4] < R[8] ? 2 : 1) : (R[0] < R[8] ? 2 : 0);
        int j = (i + 1) % 3, k = (i + 2) % 3;
        real s = (real)sqrt((double)(R[4 * i] - R[4 * j] - R[4 * k] + 1));
        q[i] = s * (real)0.5; s = (real)0.5 / s;
        q[3] = (R[3 * k + j] - R[3 * j + k]) * s;
        q[j] = (R[3 * j + i] + R[3 * i + j]) * s;
        q[k] = (R[3 * k + i] + R[3 * i + k]) * s;
    }
}
/* pybullet.c getEulerFromQuaternion (plen_env.py:799, :1017, :1030) */
static void quat_to_euler(real *rpy, const real *q) {
    real sqx = q[0] * q[0], sqy = q[1] * q[1], sqz = q[2] * q[2], squ = q[3] * q[3];
    real sarg = -2 * (q[0] * q[2] - q[3] * q[1]);
    const real PI = (real)3.14159265358979323846;
    if (sarg <= (real)-0.99999) {
        rpy[0] = 0; rpy[1] = (real)-0.5 * PI; rpy[2] = 2 * (real)atan2((double)q[0], (double)-q[1]);
    } else if (sarg >= (real)0.99999) {
        rpy[0] = 0; rpy[1] = (real)0.5 * PI; rpy[2] = 2 * (real)atan2((double)-q[0], (double)q[1]);
    } else {
        rpy[0] = (real)atan2((double)(2 * (q[1] * q[2] + q[3] * q[0])), (double)(squ - sqx - sqy + sqz));
        rpy[1] = (real)asin((double)sarg);
        rpy[2] = (real)atan2((double)(2 * (q[0] * q[1] + q[3] * q[2])), (double)(squ + sqx - sqy - sqz));
    }
}

/* 6x6 helpers; spatial vectors are [angular(3); linear(3)] in WORLD axes about a body's COM */
static void m6mulv(real *o, const real *M, const real *v) {
    real t[6];
    for (int i = 0; i < 6; i++) { real s = 0; for (int j = 0; j < 6; j++) s += M[6 * i + j] * v[j]; t[i] = s; }
    memcpy(o, t, sizeof t);
}
static real v6dot(const real *a, const real *b) { real s = 0; for (int i = 0; i < 6; i++) s += a[i] * b[i]; return s; }
/* shift a wrench given about point c (child) to point p (parent), r = c - p:  n_p = n_c + r x f */
static void wrench_shift(real *o, const real *w, const real *r) {
    real t[3]; v3cross(t, r, w + 3);
    o[0] = w[0] + t[0]; o[1] = w[1] + t[1]; o[2] = w[2] + t[2]; o[3] = w[3]; o[4] = w[4]; o[5] = w[5];
}
/* shift an (acceleration-like) motion vector from p to c = p + r:  a_c = a_p + alpha x r */
static void motion_shift(real *o, const real *m, const real *r) {
    real t[3]; v3cross(t, m, r);
    o[0] = m[0]; o[1] = m[1]; o[2] = m[2]; o[3] = m[3] + t[0]; o[4] = m[4] + t[1]; o[5] = m[5] + t[2];
}
/* I_p += X^T I X with X the motion shift p->c (r = c - p) */
static void inertia_shift_add(real *Ip, const real *Ic, const real *r) {
    /* X = [[1,0],[-[r]x,1]];  X^T = [[1,[r]x],[0,1]] */
    real X[36] = {0}, T[36], O[36];
    for (int i = 0; i < 6; i++) X[7 * i] = 1;
    /* -[r]x in the lower-left block */
    X[6 * 3 + 1] = r[2];  X[6 * 3 + 2] = -r[1];
    X[6 * 4 + 0] = -r[2]; X[6 * 4 + 2] = r[0];
    X[6 * 5 + 0] = r[1];  X[6 * 5 + 1] = -r[0];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) { real s = 0; for (int k = 0; k < 6; k++) s += Ic[6 * i + k] * X[6 * k + j]; T[6 * i + j] = s; }
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) { real s = 0; for (int k = 0; k < 6; k++) s += X[6 * k + i] * T[6 * k + j]; O[6 * i + j] = s; }
    for (int i = 0; i < 36; i++) Ip[i] += O[i];
}
/* solve 6x6 SPD system by Cholesky */
static void solve6(real *x, const real *A, const real *b) {
    real L[36] = {0};
    for (int j = 0; j < 6; j++) {
        real s = A[7 * j];
        for (int k = 0; k < j; k++) s -= L[6 * j + k] * L[6 * j + k];
        real d = (real)sqrt((double)s); L[7 * j] = d;
        for (int i = j + 1; i < 6; i++) {
            real t = A[6 * i + j];
            for (int k = 0; k < j; k++) t -= L[6 * i + k] * L[6 * j + k];
            L[6 * i + j] = t / d;
        }
    }
    real y[6];
    for (int i = 0; i < 6; i++) { real t = b[i]; for (int k = 0; k < i; k++) t -= L[6 * i + k] * y[k]; y[i] = t / L[7 * i]; }
    for (int i = 5; i >= 0; i--) { real t = y[i]; for (int k = i + 1; k < 6; k++) t -= L[6 * k + i] * x[k]; x[i] = t / L[7 * i]; }
}

/* ---------------------------------------------------------------- world parameters */
typedef struct {
    real dt;                 /* 1/240: PyBullet default, never changed (plen_env.py:298-305 commented) */
    real gravity[3];         /* plen_env.py:296 */
    int num_iterations;      /* PyBullet default 50 */
    real erp;                /* 0.2  (non-contact rows) */
    real erp2;               /* 0.08 (contact rows, PyBullet override) */
    real friction_erp;       /* 0.2 */
    real global_cfm;         /* 0 */
    real linear_slop;        /* 1e-5 (PyBullet override) */
    real residual_threshold; /* 1e-7 (PyBullet override) */
    real restitution_velocity_threshold; /* 0.2 */
    real max_coordinate_velocity;        /* btMultiBody m_maxCoordinateVelocity 100 */
    real lateral_friction;   /* foot 0.8 x plane 0.8 (plen_env.py:309,444) */
    real box_lateral_friction; /* other links: Bullet's URDF default 0.5 x plane 0.8 */
    int body_contacts;       /* 1: the box colliders of the non-foot links collide with the ground (reference behaviour); 0: feet only */
    real spinning_friction;  /* foot 0.1 x plane lateral 0.8 */
    real rolling_friction;   /* foot 0.1 (0.01 joint_act) x 0.8 (plen_env.py:439-442) */
    real restitution;        /* 0.5 x 0.5 (plen_env.py:309,481) */
    real linear_damping;     /* 0 (0.1 joint_act) (plen_env.py:472-480) */
    real angular_damping;    /* 0 */
    real motor_kp, motor_kd; /* PyBullet POSITION_CONTROL defaults 0.1 / 1.0 */
    real motor_max_force;    /* 0.15 (plen_env.py:753) */
    /* ---- hypothesis switches (scripts/pin/, DESIGN.md "Hypotheses against the PyBullet-held pin"); 0 = the model described above ---- */
    int manifold_mode;       /* 0: memoryless 8-corner reduction; 1: btPersistentManifold restated (one deepest hull vertex per collision pass,
                                getCacheEntry / sortCachedPoints / refreshContactPoints, history dependent) */
    real warmstart;          /* normal impulse of a persisting manifold point carried over, times this factor (Bullet: m_warmstartingFactor) */
    int pyramid_friction;    /* 1: the two lateral rows solved independently (SOLVER_DISABLE_IMPLICIT_CONE_FRICTION) */
    int base_gyro_off;       /* 1: no gyroscopic term on the base (btMultiBody::m_useGyroTerm false) */
    real motor_rhs_clamp;    /* > 0: btMultiBodyJointMotor::m_rhsClamp (the motor's desired velocity clamped to +- this) */
    real joint_damping;      /* btMultibodyLink::m_jointDamping of every revolute joint (URDF <dynamics damping>; the reference's URDF has none) */
    real sole_grow, sole_dz; /* memoryless model: sole candidate vertices moved outward by sole_grow (m, along the direction from the sole centre) and up by sole_dz */
    int man_cand;            /* manifold_mode 1: candidates of the per-pass new point: 0 all 209 hull vertices, 1 the 32 sole vertices, 2 the 8 corner representatives */
    real man_drift;          /* manifold_mode 1: the drift test's threshold as a multiple of the breaking threshold (0 -> 1) */
    int man_add_all;         /* manifold_mode 1: 1 = every in-range candidate goes through addContactPoint each pass, deepest last (multi-point generation) */
    int man_order;           /* manifold_mode 1, add_all: 0 = candidates inserted from the highest to the lowest, 1 = lowest first */
    real man_cache, man_range;   /* manifold_mode 1: getCacheEntry's merge radius / the in-range threshold as multiples of the breaking threshold (0 -> 1) */
    int man_p1;              /* manifold_mode 1: 1 = the FIRST point of an empty manifold is the sole-plane point (man_p1x, man_p1y) of the foot link frame
                                (what EPA returns for the flat-on-flat start of every episode is an artefact of its polytope expansion: scanned, not known) */
    real man_p1x, man_p1y;
    int man_fresh;           /* manifold_mode 1: 1 = the manifold is rebuilt from nothing every pass (memoryless; no carried impulse either) */
    int nc_order;            /* 1: motors visited in DoF order instead of btAlignedObjectArray::quickSort's scramble of equal island keys; 2: reverse DoF order */
    int no_order_flip;       /* 1: the non-contact rows are not reversed on even iterations */
    int torsional_points;    /* > 0: only the first n points of a manifold get spinning / rolling rows */
    int man_key_ground;      /* manifold_mode 1: 1 = getCacheEntry / sortCachedPoints work on m_localPointA of body A = the PLANE (loaded first: lower broadphase id),
                                i.e. on the points' GROUND positions, not on their foot-frame positions */
    int fric_order;          /* torsional rows: 0 = Bullet >= 2.87 (all spinning rows, then all rolling rows: two arrays); 1 = per point interleaved (spin, roll1, roll2: the single
                                m_multiBodyTorsionalFrictionContactConstraints array of Bullet <= 2.86); 2 = all rolling rows before the spinning rows */
    int lever_on_plane;      /* 1: the contact rows' Jacobians are taken at the point on the GROUND (cp.getPositionWorldOnA for body A = plane) instead of the point on the foot */
    int tors_freeze;         /* > 0: the bounds of the spinning / rolling rows are no longer rewritten from the normal impulse after this many iterations
                                (mechanism A/B of the expanding mode, scripts/pin/expanding_mode.py; Bullet rewrites them in every iteration) */
} World;

typedef struct {
    /* ---- physics state (49 reals) ---- */
    real base_pos[3], base_quat[4], base_omega[3], base_vel[3], q[ND], qd[ND];
    real target[ND];
    /* ---- model (mutable for domain randomisation) ---- */
    real mass[NL + 1], inertia[NL + 1][3];   /* index 0 = base, i+1 = link i */
    World w;
    int joint_act;
    /* ---- kinematics cache ---- */
    real Rw[NL + 1][9], Ow[NL + 1][3], Cw[NL + 1][3], Aw[NL + 1][3];
    real Iw[NL + 1][9];
    /* ---- ABA cache ---- */
    real IA[NL + 1][36], U[NL + 1][6], Dinv[NL + 1], S[NL + 1][6];
    /* ---- contact state of the last collision pass ---- */
    int ncp; int cp_foot[MAXCP]; real cp_pos[MAXCP][3]; real cp_dist[MAXCP];
    int cp_slot[MAXCP], cp_link[MAXCP], cp_box[MAXCP];      /* slot of contact c; link it is on; box index or -1 for a foot point */
    real cp_mu[MAXCP], cp_rest[MAXCP];                      /* combined lateral friction and restitution of contact c */
    int reward_head;                 /* 0: PlenWalkEnv-v1 (plen_env.py), 1: PlenWalkEnv-v0 contract (plen_walk.py:346-396, 597-650) */
    /* persistent foot manifolds (manifold_mode 1): point on the foot in the foot link frame, its anchor on the ground, carried normal impulse */
    struct { int n; real lA[4][3], wB[4][3], imp[4]; } man[2];
    real cp_drift[MAXCP][2];                  /* (posA - posB) along the two friction directions: the friction rows' positional error */
    real cp_warm[MAXCP]; int cp_man[MAXCP];   /* warm-start impulse of contact c and its manifold entry (-1: none) */
    real foot_force[2][3];           /* contact force on each foot over the last substep (impulses / dt), right then left */
    int right_contact, left_contact;
    int last_iterations; real last_residual;
    /* ---- env-level state (plen_env.py attributes) ---- */
    int gait_period_counter, double_support_counter, episode_timestep, dead, first_pass;
    real torso_z, torso_y, torso_vx, roll, pitch, yaw;
    real diffs[6];
    int nhist; real hist[6][HIST];  /* lhip,rhip,lknee,rknee,lankle,rankle = joints 2,8,3,9,4,10 */
} Oracle;

static const real ENV_RANGES[ND][2] = {   /* plen_env.py:148-167 */
    {-1.57, 1.57}, {-0.15, 1.5}, {-0.95, 0.75}, {-0.9, 0.3}, {-0.95, 1.2}, {-0.8, 0.4},
    {-1.57, 1.57}, {-1.5, 0.15}, {-0.75, 0.95}, {-0.3, 0.9}, {-1.2, 0.95}, {-0.4, 0.8},
    {-1.57, 1.57}, {-0.15, 1.57}, {-0.2, 0.35}, {-1.57, 1.57}, {-0.15, 1.57}, {-0.2, 0.35}};

static void world_defaults(World *w, int joint_act) {
    w->dt = (real)(1.0 / 240.0);
    v3set(w->gravity, 0, 0, (real)-9.81);
    w->num_iterations = 50;
    w->erp = (real)0.2; w->erp2 = (real)0.08; w->friction_erp = (real)0.2; w->global_cfm = 0;
    w->linear_slop = (real)0.00001; w->residual_threshold = (real)1e-7;
    w->restitution_velocity_threshold = (real)0.2; w->max_coordinate_velocity = 100;
    w->lateral_friction = (real)(0.8 * 0.8);
    w->box_lateral_friction = (real)(0.5 * 0.8); w->body_contacts = 1;
    w->spinning_friction = (real)(0.1 * 0.8);
    w->rolling_friction = (real)((joint_act ? 0.01 : 0.1) * 0.8);
    w->restitution = (real)(0.5 * 0.5);
    w->linear_damping = (real)(joint_act ? 0.1 : 0.0); w->angular_damping = 0;
    w->motor_kp = (real)0.1; w->motor_kd = 1; w->motor_max_force = (real)0.15;
}

/* ---------------------------------------------------------------- forward kinematics */
static void fk(Oracle *o) {
    quat_to_mat(o->Rw[0], o->base_quat);
    v3cpy(o->Ow[0], o->base_pos);
    real c[3] = {(real)RAW_BASE_COM[0], (real)RAW_BASE_COM[1], (real)RAW_BASE_COM[2]}, t[3];
    m3mulv(t, o->Rw[0], c); v3add(o->Cw[0], o->Ow[0], t);
    v3set(o->Aw[0], 0, 0, 0);
    for (int i = 0; i < NL; i++) {
        int p = RAW_PARENT[i] + 1, b = i + 1;
        real JR[9], jt[3], ax[3], Rq[9], Rl[9];
        for (int k = 0; k < 9; k++) JR[k] = (real)RAW_JR[i][k];
        for (int k = 0; k < 3; k++) { jt[k] = (real)RAW_JT[i][k]; ax[k] = (real)RAW_AXIS[i][k]; }
        if (RAW_JTYPE[i] == 1) { axis_angle(Rq, ax, o->q[RAW_DOF[i]]); m3mul(Rl, JR, Rq); }
        else memcpy(Rl, JR, sizeof Rl);
        m3mul(o->Rw[b], o->Rw[p], Rl);
        m3mulv(t, o->Rw[p], jt); v3add(o->Ow[b], o->Ow[p], t);
        real cl[3] = {(real)RAW_COM[i][0], (real)RAW_COM[i][1], (real)RAW_COM[i][2]};
        m3mulv(t, o->Rw[b], cl); v3add(o->Cw[b], o->Ow[b], t);
        m3mulv(o->Aw[b], o->Rw[b], ax);
    }
    /* world inertia tensors R diag(I) R^T */
    for (int b = 0; b <= NL; b++) {
        const real *R = o->Rw[b]; const real *I = o->inertia[b];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
            o->Iw[b][3 * i + j] = R[3 * i] * I[0] * R[3 * j] + R[3 * i + 1] * I[1] * R[3 * j + 1] + R[3 * i + 2] * I[2] * R[3 * j + 2];
    }
}

/* ---------------------------------------------------------------- ABA
 * World-axes formulation, every body's spatial quantities taken about its own COM, classical
 * accelerations (see DESIGN.md "Oracle dynamics").  Generalized velocity layout follows btMultiBody:
 * v[0..2] base angular velocity (world), v[3..5] base COM linear velocity (world), v[6+d] joint rates.
 * qdd receives the generalized accelerations for zero joint torque (motors act as constraints).
 */
static void aba(Oracle *o, real *qdd) {
    real om[NL + 1][3], vc[NL + 1][3], zeta[NL + 1][6], p[NL + 1][6], acc[NL + 1][6];
    const World *w = &o->w;
    v3cpy(om[0], o->base_omega); v3cpy(vc[0], o->base_vel);
    /* pass 1: velocities, velocity-product accelerations, rigid-body bias wrenches */
    for (int b = 0; b <= NL; b++) {
        if (b > 0) {
            int i = b - 1, pb = RAW_PARENT[i] + 1;
            real qd = RAW_JTYPE[i] == 1 ? o->qd[RAW_DOF[i]] : 0;
            real rel[3] = {o->Aw[b][0] * qd, o->Aw[b][1] * qd, o->Aw[b][2] * qd};
            v3add(om[b], om[pb], rel);
            real rpo[3], roc[3], t[3], t2[3];
            v3sub(rpo, o->Ow[b], o->Cw[pb]);        /* parent COM -> joint origin */
            v3sub(roc, o->Cw[b], o->Ow[b]);         /* joint origin -> child COM  */
            v3cross(t, om[pb], rpo); v3add(vc[b], vc[pb], t);
            v3cross(t, om[b], roc);  v3add(vc[b], vc[b], t);
            /* zeta_ang = om_p x (a qd);  zeta_lin = om_p x (om_p x rpo) + zeta_ang x roc + om_c x (om_c x roc) */
            v3cross(zeta[b], om[pb], rel);
            v3cross(t, om[pb], rpo); v3cross(t2, om[pb], t); v3cpy(zeta[b] + 3, t2);
            v3cross(t, zeta[b], roc); v3add(zeta[b] + 3, zeta[b] + 3, t);
            v3cross(t, om[b], roc); v3cross(t2, om[b], t); v3add(zeta[b] + 3, zeta[b] + 3, t2);
            /* joint motion subspace about the child COM: [a; a x roc] */
            v3cpy(o->S[b], o->Aw[b]); v3cross(o->S[b] + 3, o->Aw[b], roc);
        }
        /* articulated inertia starts as the rigid-body inertia; bias = gyroscopic - gravity + damping */
        real *IA = o->IA[b]; memset(IA, 0, 36 * sizeof(real));
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) IA[6 * i + j] = o->Iw[b][3 * i + j];
        IA[21] = IA[28] = IA[35] = o->mass[b];
        real Iom[3], g[3];
        m3mulv(Iom, o->Iw[b], om[b]); v3cross(g, om[b], Iom);
        if (b == 0 && w->base_gyro_off) v3set(g, 0, 0, 0);
        real wn = v3norm(om[b]), vn = v3norm(vc[b]);
        for (int k = 0; k < 3; k++) {
            p[b][k] = g[k] + Iom[k] * (w->angular_damping + w->angular_damping * wn);
            p[b][3 + k] = -o->mass[b] * w->gravity[k] + o->mass[b] * vc[b][k] * (w->linear_damping + w->linear_damping * vn);
        }
    }
    /* pass 2: leaves -> root (links are in DFS pre-order, so reverse index order works) */
    for (int b = NL; b >= 1; b--) {
        int i = b - 1, pb = RAW_PARENT[i] + 1;
        real r[3]; v3sub(r, o->Cw[b], o->Cw[pb]);
        real Ia[36], pa[6], t6[6];
        memcpy(Ia, o->IA[b], sizeof Ia);
        if (RAW_JTYPE[i] == 1) {
            m6mulv(o->U[b], o->IA[b], o->S[b]);
            real D = v6dot(o->S[b], o->U[b]);
            o->Dinv[b] = 1 / D;
            real u = -v6dot(o->S[b], p[b]) - w->joint_damping * o->qd[RAW_DOF[i]];   /* tau = 0 (+ joint damping, 0 in the reference) */
            for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++) Ia[6 * a + c] -= o->U[b][a] * o->Dinv[b] * o->U[b][c];
            m6mulv(t6, Ia, zeta[b]);
            for (int a = 0; a < 6; a++) pa[a] = p[b][a] + t6[a] + o->U[b][a] * o->Dinv[b] * u;
        } else {
            m6mulv(t6, Ia, zeta[b]);
            for (int a = 0; a < 6; a++) pa[a] = p[b][a] + t6[a];
        }
        inertia_shift_add(o->IA[pb], Ia, r);
        wrench_shift(t6, pa, r);
        for (int a = 0; a < 6; a++) p[pb][a] += t6[a];
    }
    /* base: IA a0 = -p0 */
    real rhs[6]; for (int a = 0; a < 6; a++) rhs[a] = -p[0][a];
    solve6(acc[0], o->IA[0], rhs);
    for (int a = 0; a < 6; a++) qdd[a] = acc[0][a];
    /* pass 3: root -> leaves */
    for (int b = 1; b <= NL; b++) {
        int i = b - 1, pb = RAW_PARENT[i] + 1;
        real r[3]; v3sub(r, o->Cw[b], o->Cw[pb]);
        real ap[6]; motion_shift(ap, acc[pb], r);
        for (int a = 0; a < 6; a++) ap[a] += zeta[b][a];
        if (RAW_JTYPE[i] == 1) {
            real u = -v6dot(o->S[b], p[b]) - o->w.joint_damping * o->qd[RAW_DOF[i]];
            real qa = o->Dinv[b] * (u - v6dot(o->U[b], ap));
            qdd[6 + RAW_DOF[i]] = qa;
            for (int a = 0; a < 6; a++) acc[b][a] = ap[a] + o->S[b][a] * qa;
        } else memcpy(acc[b], ap, sizeof ap);
    }
}

/* btMultiBody::calcAccelerationDeltasMultiDof: out = M^-1 force, using the cached ABA quantities */
static void aba_delta(const Oracle *o, const real *force, real *out) {
    real p[NL + 1][6], acc[NL + 1][6], u[NL + 1];
    memset(p, 0, sizeof p);
    for (int b = NL; b >= 1; b--) {
        int i = b - 1, pb = RAW_PARENT[i] + 1;
        real r[3]; v3sub(r, o->Cw[b], o->Cw[pb]);
        real pa[6], t6[6];
        if (RAW_JTYPE[i] == 1) {
            u[b] = force[6 + RAW_DOF[i]] - v6dot(o->S[b], p[b]);
            for (int a = 0; a < 6; a++) pa[a] = p[b][a] + o->U[b][a] * o->Dinv[b] * u[b];
        } else memcpy(pa, p[b], sizeof pa);
        wrench_shift(t6, pa, r);
        for (int a = 0; a < 6; a++) p[pb][a] += t6[a];
    }
    real rhs[6]; for (int a = 0; a < 6; a++) rhs[a] = force[a] - p[0][a];
    solve6(acc[0], o->IA[0], rhs);
    for (int a = 0; a < 6; a++) out[a] = acc[0][a];
    for (int b = 1; b <= NL; b++) {
        int i = b - 1, pb = RAW_PARENT[i] + 1;
        real r[3]; v3sub(r, o->Cw[b], o->Cw[pb]);
        real ap[6]; motion_shift(ap, acc[pb], r);
        if (RAW_JTYPE[i] == 1) {
            real qa = o->Dinv[b] * (u[b] - v6dot(o->U[b], ap));
            out[6 + RAW_DOF[i]] = qa;
            for (int a = 0; a < 6; a++) acc[b][a] = ap[a] + o->S[b][a] * qa;
        } else memcpy(acc[b], ap, sizeof ap);
    }
}

/* ---------------------------------------------------------------- constraint rows */
typedef struct {
    real jac[NV], delta[NV];
    real jac_diag_inv, rhs, cfm, lo, hi, applied, friction;
    int friction_index;
} Row;

/* btMultiBody::fillContactJacobianMultiDof / fillConstraintJacobianMultiDof:
 * row for direction (n_ang, n_lin) at world point P on link `link` (link index, base = -1) */
static void fill_jacobian(const Oracle *o, int link, const real *P, const real *n_ang, const real *n_lin, real *jac) {
    memset(jac, 0, NV * sizeof(real));
    real r[3], t[3];
    v3sub(r, P, o->Cw[0]); v3cross(t, r, n_lin);
    for (int k = 0; k < 3; k++) { jac[k] = t[k] + n_ang[k]; jac[3 + k] = n_lin[k]; }
    for (int i = link; i >= 0; i = RAW_PARENT[i]) {
        if (RAW_JTYPE[i] != 1) continue;
        int b = i + 1;
        v3sub(r, P, o->Ow[b]); v3cross(t, r, n_lin);
        jac[6 + RAW_DOF[i]] = v3dot(o->Aw[b], t) + v3dot(o->Aw[b], n_ang);
    }
}

static void gen_vel(const Oracle *o, real *v) {
    v3cpy(v, o->base_omega); v3cpy(v + 3, o->base_vel);
    for (int d = 0; d < ND; d++) v[6 + d] = o->qd[d];
}

static real row_finish(const Oracle *o, Row *r, real cfm_in) {
    aba_delta(o, r->jac, r->delta);
    real d = 0; for (int k = 0; k < NV; k++) d += r->jac[k] * r->delta[k];
    d += cfm_in;
    const real EPS = (sizeof(real) == 8) ? (real)2.220446049250313e-16 : (real)1.1920929e-07f;
    r->jac_diag_inv = d > EPS ? 1 / d : 0;
    real v[NV], rel = 0; gen_vel(o, v);
    for (int k = 0; k < NV; k++) rel += v[k] * r->jac[k];
    r->applied = 0;
    return rel;
}

/* btSequentialImpulseConstraintSolver::restitutionCurve */
static real restitution_curve(real rel_vel, real restitution, real thr) {
    if (fabs((double)rel_vel) < thr) return 0;
    return restitution * -rel_vel;
}

/* btMultiBodyConstraintSolver::resolveSingleConstraintRowGeneric */
static real resolve_row(Row *c, real *dv) {
    real di = c->rhs - c->applied * c->cfm;
    real dvn = 0; for (int k = 0; k < NV; k++) dvn += c->jac[k] * dv[k];
    di -= dvn * c->jac_diag_inv;
    real sum = c->applied + di;
    if (sum < c->lo) { di = c->lo - c->applied; c->applied = c->lo; }
    else if (sum > c->hi) { di = c->hi - c->applied; c->applied = c->hi; }
    else c->applied = sum;
    for (int k = 0; k < NV; k++) dv[k] += c->delta[k] * di;
    return di / c->jac_diag_inv;
}

/* btMultiBodyConstraintSolver::resolveConeFrictionConstraintRows */
static real resolve_cone(Row *cA, Row *cB, real *dv) {
    real dB = cB->rhs - cB->applied * cB->cfm, dA = cA->rhs - cA->applied * cA->cfm;
    real nB = 0, nA = 0;
    for (int k = 0; k < NV; k++) { nB += cB->jac[k] * dv[k]; nA += cA->jac[k] * dv[k]; }
    dB -= nB * cB->jac_diag_inv; dA -= nA * cA->jac_diag_inv;
    real sumB = cB->applied + dB, sumA = cA->applied + dA;
    if (sumA * sumA + sumB * sumB >= cA->lo * cB->lo) {
        real angle = (real)atan2((double)sumA, (double)sumB);
        real ca = (real)fabs((double)(cA->lo * (real)sin((double)angle)));
        real cb = (real)fabs((double)(cB->lo * (real)cos((double)angle)));
        if (sumA < -ca) { dA = -ca - cA->applied; cA->applied = -ca; }
        else if (sumA > ca) { dA = ca - cA->applied; cA->applied = ca; }
        else cA->applied = sumA;
        if (sumB < -cb) { dB = -cb - cB->applied; cB->applied = -cb; }
        else if (sumB > cb) { dB = cb - cB->applied; cB->applied = cb; }
        else cB->applied = sumB;
    } else { cA->applied = sumA; cB->applied = sumB; }
    for (int k = 0; k < NV; k++) dv[k] += cA->delta[k] * dA + cB->delta[k] * dB;
    return dA / cA->jac_diag_inv + dB / cB->jac_diag_inv;
}

/* ---------------------------------------------------------------- collision: feet and link boxes vs ground */
typedef struct { real dist; int id, box, link; real pos[3]; } BoxCand;
static int cand_less(const BoxCand *a, const BoxCand *b) { return a->dist < b->dist || (a->dist == b->dist && a->id < b->id); }

static void collide(Oracle *o) {
    o->ncp = 0; o->right_contact = 0; o->left_contact = 0;
    int slot_used[MAXCP] = {0};
    struct { int used, foot, link, box, man; real pos[3], dist, mu, rest, warm, drift[2]; } slot[MAXCP];
    memset(slot, 0, sizeof slot);
    for (int c = 0; c < MAXCP; c++) slot[c].man = -1;
    if (o->w.manifold_mode == 1) for (int f = 0; f < 2; f++) {
        /* btPersistentManifold restated (foot = body A of its manifold: btCompoundCollisionAlgorithm hands the child hull in first) */
        int link = f == 0 ? RAW_RFOOT_LINK : RAW_LFOOT_LINK, b = link + 1;
        const real thr = (real)(f == 0 ? RAW_RFOOT_BREAK : RAW_LFOOT_BREAK), thr2 = thr * thr;
        const real *R = o->Rw[b], *O = o->Ow[b];
        if (o->w.man_fresh) o->man[f].n = 0;
        /* refreshContactPoints: distance along the normal and drift in the plane, from the stored local points */
        for (int i = o->man[f].n - 1; i >= 0; i--) {
            real pa[3]; m3mulv(pa, R, o->man[f].lA[i]); v3add(pa, pa, O);
            real dist = pa[2] - o->man[f].wB[i][2];
            real dx = o->man[f].wB[i][0] - pa[0], dy = o->man[f].wB[i][1] - pa[1];
            const real dthr = o->w.man_drift > 0 ? thr * o->w.man_drift : thr;
            if (dist > thr || dx * dx + dy * dy > dthr * dthr) {          /* removeContactPoint: the last entry takes its place */
                int last = --o->man[f].n;
                if (i != last) { v3cpy(o->man[f].lA[i], o->man[f].lA[last]); v3cpy(o->man[f].wB[i], o->man[f].wB[last]); o->man[f].imp[i] = o->man[f].imp[last]; }
            }
        }
        /* the collision pass's new point(s): GJK's closest point = the lowest candidate vertex, sphere-swept by the margin */
        const int ncand_v = o->w.man_cand == 0 ? (f == 0 ? RAW_RFOOT_NHULL : RAW_LFOOT_NHULL) : 32;
        real cw[209][3]; int cok[209];
        for (int v = 0; v < ncand_v; v++) {
            const double *pl = o->w.man_cand == 0 ? (f == 0 ? RAW_RFOOT_HULL[v] : RAW_LFOOT_HULL[v]) : (f == 0 ? RAW_RFOOT_SOLE[v] : RAW_LFOOT_SOLE[v]);
            real l[3] = {(real)pl[0], (real)pl[1], (real)pl[2]};
            m3mulv(cw[v], R, l); v3add(cw[v], cw[v], O);
            cok[v] = o->w.man_cand == 2 ? (f == 0 ? RAW_RFOOT_SOLE_REP[v] : RAW_LFOOT_SOLE_REP[v]) : 1;
        }
        /* order of insertion: one point (the lowest) or all in-range candidates from the highest to the lowest (the deepest added last) */
        int ord[209], nord = 0;
        const real rthr = o->w.man_range > 0 ? thr * o->w.man_range : thr;
        for (int v = 0; v < ncand_v; v++) if (cok[v] && cw[v][2] - (real)RAW_MARGIN <= rthr) ord[nord++] = v;
        for (int i = 1; i < nord; i++) { int k = ord[i], j = i - 1; while (j >= 0 && cw[ord[j]][2] < cw[k][2]) { ord[j + 1] = ord[j]; j--; } ord[j + 1] = k; }
        if (o->w.man_order == 1 && o->w.man_add_all) for (int i = 0; i < nord / 2; i++) { int t_ = ord[i]; ord[i] = ord[nord - 1 - i]; ord[nord - 1 - i] = t_; }
        real p1w[3];
        const int use_p1 = o->w.man_p1 && o->man[f].n == 0 && nord > 0;
        if (use_p1) {
            const double *pl = f == 0 ? RAW_RFOOT_SOLE[0] : RAW_LFOOT_SOLE[0];
            real l[3] = {o->w.man_p1x, o->w.man_p1y, (real)pl[2]};
            m3mulv(p1w, R, l); v3add(p1w, p1w, O);
        }
        for (int oi = (o->w.man_add_all ? 0 : (nord > 0 ? nord - 1 : 0)); oi < nord; oi++) {
            const real *lw = use_p1 ? p1w : cw[ord[oi]];
            real depth = lw[2] - (real)RAW_MARGIN;
            {
            real pa[3] = {lw[0], lw[1], depth}, d[3], la[3];
            v3sub(d, pa, O);
            for (int k = 0; k < 3; k++) la[k] = R[k] * d[0] + R[3 + k] * d[1] + R[6 + k] * d[2];      /* R^T (pa - O) */
            int near = -1; real best = o->w.man_cache > 0 ? thr2 * o->w.man_cache * o->w.man_cache : thr2;                          /* getCacheEntry */
            const real gnd[3] = {lw[0], lw[1], 0};
            for (int i = 0; i < o->man[f].n; i++) { real e[3]; if (o->w.man_key_ground) v3sub(e, o->man[f].wB[i], gnd); else v3sub(e, o->man[f].lA[i], la); real q = v3dot(e, e); if (q < best) { best = q; near = i; } }
            int ins;
            if (near >= 0) ins = near;                                /* replaceContactPoint keeps the applied impulse */
            else if (o->man[f].n < 4) { ins = o->man[f].n++; o->man[f].imp[ins] = 0; }
            else {                                                    /* sortCachedPoints (gContactCalcArea3Points), deepest point kept */
                int deep = -1; real maxpen = depth;
                for (int i = 0; i < 4; i++) {
                    real pi[3]; m3mulv(pi, R, o->man[f].lA[i]); v3add(pi, pi, O);
                    real di = pi[2] - o->man[f].wB[i][2];
                    if (di < maxpen) { deep = i; maxpen = di; }
                }
                real res[4] = {0, 0, 0, 0}, a[3], bb[3], c[3];
                const real (*P)[3] = o->w.man_key_ground ? o->man[f].wB : o->man[f].lA;
                const real *la_k = o->w.man_key_ground ? gnd : la;
                if (deep != 0) { v3sub(a, la_k, P[1]); v3sub(bb, P[3], P[2]); v3cross(c, a, bb); res[0] = v3dot(c, c); }
                if (deep != 1) { v3sub(a, la_k, P[0]); v3sub(bb, P[3], P[2]); v3cross(c, a, bb); res[1] = v3dot(c, c); }
                if (deep != 2) { v3sub(a, la_k, P[0]); v3sub(bb, P[3], P[1]); v3cross(c, a, bb); res[2] = v3dot(c, c); }
                if (deep != 3) { v3sub(a, la_k, P[0]); v3sub(bb, P[2], P[1]); v3cross(c, a, bb); res[3] = v3dot(c, c); }
                ins = 0; for (int i = 1; i < 4; i++) if (res[i] > res[ins]) ins = i;     /* btVector4::closestAxis4 */
                o->man[f].imp[ins] = 0;
            }
            v3cpy(o->man[f].lA[ins], la); v3set(o->man[f].wB[ins], lw[0], lw[1], 0);
            }
        }
        for (int i = 0; i < o->man[f].n; i++) {
            int c = 4 * f + i;
            real pa[3]; m3mulv(pa, R, o->man[f].lA[i]); v3add(pa, pa, O);
            slot[c].used = 1; slot[c].foot = f; slot[c].link = link; slot[c].box = -1; slot[c].dist = pa[2] - o->man[f].wB[i][2];
            v3cpy(slot[c].pos, pa);
            slot[c].mu = o->w.lateral_friction; slot[c].rest = o->w.restitution;
            slot[c].warm = o->man[f].imp[i] * o->w.warmstart; slot[c].man = i;
            slot[c].drift[0] = -(pa[1] - o->man[f].wB[i][1]); slot[c].drift[1] = pa[0] - o->man[f].wB[i][0];    /* along (0,-1,0) and (1,0,0) */
            slot_used[c] = 1;
            if (f == 0) o->right_contact = 1; else o->left_contact = 1;
        }
    }
    else for (int f = 0; f < 2; f++) {
        int link = f == 0 ? RAW_RFOOT_LINK : RAW_LFOOT_LINK, b = link + 1;
        real thr = (real)(f == 0 ? RAW_RFOOT_BREAK : RAW_LFOOT_BREAK);
        /* candidates: the representatives of the sole outline's 8 corner fillets (Bullet's manifold merges points closer than the breaking
         * threshold, and a fillet is ~1.5 mm long), sphere-swept by the margin; in range while distance <= breaking threshold */
        real wv[32][3]; int in_range[32];
        for (int v = 0; v < 32; v++) {
            const double *pl = f == 0 ? RAW_RFOOT_SOLE[v] : RAW_LFOOT_SOLE[v];
            real l[3] = {(real)pl[0], (real)pl[1], (real)pl[2]};
            if (o->w.sole_grow != 0 || o->w.sole_dz != 0) {
                double cx = 0, cy = 0; for (int q = 0; q < 32; q++) { const double *pq = f == 0 ? RAW_RFOOT_SOLE[q] : RAW_LFOOT_SOLE[q]; cx += pq[0] / 32; cy += pq[1] / 32; }
                double dx = pl[0] - cx, dy = pl[1] - cy, n = sqrt(dx * dx + dy * dy);
                l[0] += (real)(o->w.sole_grow * dx / n); l[1] += (real)(o->w.sole_grow * dy / n); l[2] += o->w.sole_dz;
            }
            m3mulv(wv[v], o->Rw[b], l); v3add(wv[v], wv[v], o->Ow[b]);
            in_range[v] = (f == 0 ? RAW_RFOOT_SOLE_REP[v] : RAW_LFOOT_SOLE_REP[v]) && (wv[v][2] - (real)RAW_MARGIN) <= thr;
        }
        /* manifold reduction: per sole diagonal k the in-range vertex extreme along it (first in-range entry of the precomputed order);
         * a vertex already chosen for an earlier diagonal is not repeated.  Whole sole in range -> the four corner-most vertices. */
        int chosen[4] = {-1, -1, -1, -1}, nfoot = 0;     /* the foot's points take its lowest slots, in diagonal order (row order unchanged; as the kernel) */
        for (int k = 0; k < 4; k++) {
            const int *ord = f == 0 ? RAW_RFOOT_SOLE_ORDER[k] : RAW_LFOOT_SOLE_ORDER[k];
            int win = -1;
            for (int j = 0; j < 32; j++) if (in_range[ord[j]]) { win = ord[j]; break; }
            for (int k2 = 0; k2 < k; k2++) if (win >= 0 && chosen[k2] == win) win = -1;
            chosen[k] = win;
            if (win < 0) continue;
            int c = 4 * f + nfoot++;
            real dist = wv[win][2] - (real)RAW_MARGIN;
            slot[c].used = 1; slot[c].foot = f; slot[c].link = link; slot[c].box = -1; slot[c].dist = dist;
            v3set(slot[c].pos, wv[win][0], wv[win][1], dist);   /* position on the robot (sphere-swept vertex) */
            slot[c].mu = o->w.lateral_friction; slot[c].rest = o->w.restitution;
            slot_used[c] = 1;
            if (f == 0) o->right_contact = 1; else o->left_contact = 1;
        }
    }
    if (o->w.body_contacts) {
        /* near boxes in box order (lowest point of the box within its breaking threshold), then their corners */
        BoxCand cand[MAXNEAR * 8]; int ncand = 0, nnear = 0;
        for (int x = 0; x < RAW_NBOX && nnear < MAXNEAR; x++) {
            int b = RAW_BOX_LINK[x] + 1;
            const real *Rw = o->Rw[b];
            real Rb[9], t[3], h[3], c[3], ax[9];
            for (int i = 0; i < 9; i++) Rb[i] = (real)RAW_BOX_R[x][i];
            for (int i = 0; i < 3; i++) { t[i] = (real)RAW_BOX_T[x][i]; h[i] = (real)RAW_BOX_H[x][i]; }
            m3mulv(c, Rw, t); v3add(c, c, o->Ow[b]);
            m3mul(ax, Rw, Rb);                         /* columns = box axes in world */
            real zmin = c[2] - (h[0] * (real)fabs((double)ax[6]) + h[1] * (real)fabs((double)ax[7]) + h[2] * (real)fabs((double)ax[8]));
            real thr = (real)RAW_BOX_BREAK[x];
            if (!(zmin <= thr)) continue;
            int r = nnear++;
            for (int cn = 0; cn < 8; cn++) {
                real sg[3] = {(cn & 1) ? h[0] : -h[0], (cn & 2) ? h[1] : -h[1], (cn & 4) ? h[2] : -h[2]}, w[3];
                m3mulv(w, ax, sg); v3add(w, w, c);
                if (w[2] <= thr) {
                    BoxCand *q = &cand[ncand++];
                    q->dist = w[2]; q->id = 8 * r + cn; q->box = x; q->link = RAW_BOX_LINK[x]; v3cpy(q->pos, w);
                }
            }
        }
        /* both feet fully planted (all 8 slots hold foot points) and another link near the ground: each foot gives up the slot of its FOURTH
         * point -- a flat foot stands on three corners as well, and its occupied slots stay a prefix -- so that the link that touches down
         * (a hand pressed to the floor while standing) is held up too.  Bullet keeps every manifold; this is the 8-slot layout's way of never
         * leaving a touching link without a contact (round 3; before, such corners were dropped). */
        int released = 0;
        if (ncand > 0) {
            int all = 1; for (int c = 0; c < MAXCP; c++) all = all && slot_used[c];
            if (all) { slot_used[3] = slot_used[7] = 0; released = 1; }        /* slot[3], slot[7] keep the foot points: a released slot nobody takes gets its point back */
        }
        /* deepest first (ties: candidate id) into the free slots in slot order */
        for (int i = 1; i < ncand; i++) { BoxCand k = cand[i]; int j = i - 1; while (j >= 0 && cand_less(&k, &cand[j])) { cand[j + 1] = cand[j]; j--; } cand[j + 1] = k; }
        int next = 0;
        for (int c = 0; c < MAXCP && next < ncand; c++) {
            if (slot_used[c]) continue;
            const BoxCand *q = &cand[next++];
            slot[c].used = 1; slot[c].foot = -1; slot[c].link = q->link; slot[c].box = q->box; slot[c].dist = q->dist;
            v3cpy(slot[c].pos, q->pos);
            slot[c].mu = o->w.box_lateral_friction;
            slot[c].rest = (real)RAW_BOX_LINK_RESTITUTION[q->box] * (real)0.5;      /* x plane restitution 0.5 (plen_env.py:309) */
            slot_used[c] = 1;
        }
        (void)released;              /* an untaken released slot still holds its foot point (slot[c].used stayed 1) */
    }
    for (int c = 0; c < MAXCP; c++) {
        if (!slot[c].used) continue;
        int n = o->ncp++;
        o->cp_slot[n] = c; o->cp_foot[n] = slot[c].foot; o->cp_link[n] = slot[c].link; o->cp_box[n] = slot[c].box;
        o->cp_dist[n] = slot[c].dist; v3cpy(o->cp_pos[n], slot[c].pos); o->cp_mu[n] = slot[c].mu; o->cp_rest[n] = slot[c].rest;
        o->cp_warm[n] = slot[c].warm; o->cp_man[n] = slot[c].man; o->cp_drift[n][0] = slot[c].drift[0]; o->cp_drift[n][1] = slot[c].drift[1];
    }
}

/* ---------------------------------------------------------------- one 1/240 s substep */
/* optional solver trace (scripts/pin/expanding_mode.py): per iteration the residual and the delta-velocity vector; not thread safe, debugging only */
static double *g_trace = 0; static int g_trace_cap = 0;
static void substep(Oracle *o) {
    const World *w = &o->w;
    const real dt = w->dt;
    fk(o);
    collide(o);                                   /* performDiscreteCollisionDetection */
    /* stepVelocities: v += dt * ABA(q, v) */
    real qdd[NV], v[NV];
    aba(o, qdd);
    gen_vel(o, v);
    for (int k = 0; k < NV; k++) {
        v[k] += qdd[k] * dt;
        if (v[k] > w->max_coordinate_velocity) v[k] = w->max_coordinate_velocity;
        if (v[k] < -w->max_coordinate_velocity) v[k] = -w->max_coordinate_velocity;
    }
    v3cpy(o->base_omega, v); v3cpy(o->base_vel, v + 3);
    for (int d = 0; d < ND; d++) o->qd[d] = v[6 + d];

    /* ---- row setup ---- */
    static const real Z3[3] = {0, 0, 0};
    Row nc[36], nrm[MAXCP], spin[MAXCP], roll[2 * MAXCP], fric[2 * MAXCP];
    int n_nc = 0, n_n = 0, n_spin = 0, n_roll = 0, n_fric = 0;
    int spin_of[MAXCP], roll_of[MAXCP];
    for (int s = 0; s < 36; s++) {
        int d = RAW_NC_DOF[s];
        if (w->nc_order == 1) d = s % 18; else if (w->nc_order == 2) d = 17 - s % 18;
        int link = RAW_MOVING[d];
        if (RAW_NC_KIND[s] == 1) {
            /* btMultiBodyJointMotor::createConstraintRows (POSITION_CONTROL, kp 0.1, kd 1, target vel 0) */
            Row *r = &nc[n_nc++];
            memset(r->jac, 0, sizeof r->jac); r->jac[6 + d] = 1;
            real rel = row_finish(o, r, 0);
            real pos_stab = (o->target[d] - o->q[d]) / dt;               /* motor erp = 1 */
            real desired = w->motor_kp * pos_stab + o->qd[d] + w->motor_kd * (0 - o->qd[d]);
            if (w->motor_rhs_clamp > 0) { if (desired > w->motor_rhs_clamp) desired = w->motor_rhs_clamp; if (desired < -w->motor_rhs_clamp) desired = -w->motor_rhs_clamp; }
            real vel_err = desired - rel;
            r->rhs = vel_err * r->jac_diag_inv; r->cfm = 0;
            r->hi = w->motor_max_force * dt; r->lo = -r->hi;
        } else {
            /* btMultiBodyJointLimitConstraint: a row only while the limit is violated */
            real lo = (real)RAW_LOWER[link], hi = (real)RAW_UPPER[link];
            for (int side = 0; side < 2; side++) {
                real pen = side == 0 ? o->q[d] - lo : hi - o->q[d];
                if (pen > 0) continue;
                Row *r = &nc[n_nc++];
                memset(r->jac, 0, sizeof r->jac); r->jac[6 + d] = side == 0 ? 1 : -1;
                real rel = row_finish(o, r, 0);
                /* split impulse is on and unimplemented for multibodies: beyond the -0.04 threshold
                 * the positional part goes to m_rhsPenetration, which the multibody solver ignores */
                real pos_err = pen > (real)-0.04 ? -pen * w->erp / dt : 0;
                r->rhs = pos_err * r->jac_diag_inv + (0 - rel) * r->jac_diag_inv; r->cfm = 0;
                r->lo = 0; r->hi = 100;           /* btMultiBodyConstraint m_maxAppliedImpulse default */
            }
        }
    }
    /* convertMultiBodyContact: per manifold point normal, spinning, 2 rolling, 2 lateral rows */
    const real nrmW[3] = {0, 0, 1};
    const real dir1[3] = {0, -1, 0}, dir2[3] = {1, 0, 0};          /* btPlaneSpace1((0,0,1)) */
    for (int c = 0; c < o->ncp; c++) {
        const int link = o->cp_link[c], is_foot = o->cp_foot[c] >= 0;
        real Pl[3] = {o->cp_pos[c][0], o->cp_pos[c][1], w->lever_on_plane ? (real)0 : o->cp_pos[c][2]};
        const real *P = Pl;
        Row *r = &nrm[n_n];
        fill_jacobian(o, link, P, Z3, nrmW, r->jac);
        real cfm = w->global_cfm / dt;
        real rel = row_finish(o, r, cfm);
        real distance = o->cp_dist[c] + w->linear_slop;
        real rest = restitution_curve(rel, o->cp_rest[c], w->restitution_velocity_threshold);
        if (rest <= 0) rest = 0;
        real pos_err = 0, vel_err = rest - rel;
        if (distance > 0) vel_err -= distance / dt; else pos_err = -distance * w->erp2 / dt;
        r->rhs = pos_err * r->jac_diag_inv + vel_err * r->jac_diag_inv;
        r->cfm = cfm * r->jac_diag_inv; r->lo = 0; r->hi = (real)1e10; r->friction = o->cp_mu[c];
        r->friction_index = n_n;
        const int tors_idx = o->cp_man[c] >= 0 ? o->cp_man[c] : (o->cp_slot[c] & 3);          /* position of the point in its foot's manifold */
        const int tors_ok = !(w->torsional_points > 0 && tors_idx >= w->torsional_points);
        spin_of[n_n] = roll_of[n_n] = -1;
        if (is_foot && tors_ok && w->spinning_friction > 0) {      /* spinning / rolling friction is set on the two foot links only (plen_env.py:439-467) */
            spin_of[n_n] = n_spin;
            Row *t = &spin[n_spin++];
            fill_jacobian(o, link, P, nrmW, Z3, t->jac);
            real rv = row_finish(o, t, 0);
            t->rhs = (0 - rv) * t->jac_diag_inv; t->cfm = 0; t->friction = w->spinning_friction; t->friction_index = n_n;
            t->lo = -t->friction; t->hi = t->friction;
        }
        if (is_foot && tors_ok && w->rolling_friction > 0) {
            roll_of[n_n] = n_roll;
            for (int a = 0; a < 2; a++) {
                Row *t = &roll[n_roll++];
                fill_jacobian(o, link, P, a == 0 ? dir1 : dir2, Z3, t->jac);
                real rv = row_finish(o, t, 0);
                t->rhs = (0 - rv) * t->jac_diag_inv; t->cfm = 0; t->friction = w->rolling_friction; t->friction_index = n_n;
                t->lo = -t->friction; t->hi = t->friction;
            }
        }
        for (int a = 0; a < 2; a++) {
            Row *t = &fric[n_fric++];
            fill_jacobian(o, link, P, Z3, a == 0 ? dir1 : dir2, t->jac);
            real rv = row_finish(o, t, 0);         /* frictionCFM = 0; positional error = drift from the manifold point's anchor x frictionERP (persistent manifolds only) */
            real perr = w->manifold_mode == 1 ? -o->cp_drift[c][a] * w->friction_erp / dt : 0;
            t->rhs = (perr + (0 - rv)) * t->jac_diag_inv; t->cfm = 0; t->friction = o->cp_mu[c]; t->friction_index = n_n;
            t->lo = -t->friction; t->hi = t->friction;
        }
        n_n++;
    }

    /* ---- solveGroupCacheFriendlyIterations ---- */
    real dv[NV]; memset(dv, 0, sizeof dv);
    if (w->warmstart > 0) for (int j = 0; j < n_n; j++) if (o->cp_warm[j] > 0) {     /* setupMultiBodyContactConstraint: warm starting */
        nrm[j].applied = o->cp_warm[j];
        for (int k = 0; k < NV; k++) dv[k] += nrm[j].delta[k] * o->cp_warm[j];
    }
    int it; real residual = 0;
    for (it = 0; it < w->num_iterations; it++) {
        residual = 0;
        for (int j = 0; j < n_nc; j++) {
            int idx = ((it & 1) || w->no_order_flip) ? j : n_nc - 1 - j;
            real r = resolve_row(&nc[idx], dv); if (r * r > residual) residual = r * r;
        }
        for (int j = 0; j < n_n; j++) { real r = resolve_row(&nrm[j], dv); if (r * r > residual) residual = r * r; }
        if (w->fric_order == 1) {          /* per point: spin, roll1, roll2 (rows of point c: spin_of[c], roll_of[c], roll_of[c] + 1; -1 = none) */
            for (int c = 0; c < n_n; c++) {
                Row *rs[3] = {spin_of[c] >= 0 ? &spin[spin_of[c]] : 0, roll_of[c] >= 0 ? &roll[roll_of[c]] : 0, roll_of[c] >= 0 ? &roll[roll_of[c] + 1] : 0};
                for (int q = 0; q < 3; q++) if (rs[q]) {
                    real tot = nrm[rs[q]->friction_index].applied;
                    if (tot > 0) { if (!(w->tors_freeze > 0 && it >= w->tors_freeze)) { rs[q]->lo = -rs[q]->friction * tot; rs[q]->hi = rs[q]->friction * tot; }
                        real r = resolve_row(rs[q], dv); if (r * r > residual) residual = r * r; }
                }
            }
        } else for (int pass = 0; pass < 2; pass++) {
        if ((pass == 0) == (w->fric_order != 2))
        for (int j = 0; j < n_spin; j++) {
            real tot = nrm[spin[j].friction_index].applied;
            if (tot > 0) { if (!(w->tors_freeze > 0 && it >= w->tors_freeze)) { spin[j].lo = -spin[j].friction * tot; spin[j].hi = spin[j].friction * tot; }
                real r = resolve_row(&spin[j], dv); if (r * r > residual) residual = r * r; }
        }
        else
        for (int j = 0; j < n_roll; j++) {
            real tot = nrm[roll[j].friction_index].applied;
            if (tot > 0) { if (!(w->tors_freeze > 0 && it >= w->tors_freeze)) { roll[j].lo = -roll[j].friction * tot; roll[j].hi = roll[j].friction * tot; }
                real r = resolve_row(&roll[j], dv); if (r * r > residual) residual = r * r; }
        }
        }
        for (int j = 0; j + 1 < n_fric; j += 2) {
            real tot = nrm[fric[j].friction_index].applied;
            fric[j].lo = -fric[j].friction * tot; fric[j].hi = fric[j].friction * tot;
            fric[j + 1].lo = fric[j].lo; fric[j + 1].hi = fric[j].hi;
            if (w->pyramid_friction) {
                if (tot > 0) for (int a = 0; a < 2; a++) { real r = resolve_row(&fric[j + a], dv); if (r * r > residual) residual = r * r; }
            } else { real r = resolve_cone(&fric[j], &fric[j + 1], dv); if (r * r > residual) residual = r * r; }
        }
        if (g_trace && it < g_trace_cap) { double *tr = g_trace + (size_t)it * (2 + NV + 8); tr[0] = residual; tr[1] = n_nc; for (int k = 0; k < NV; k++) tr[2 + k] = dv[k]; for (int j = 0; j < 8; j++) tr[2 + NV + j] = j < n_n ? nrm[j].applied : 0; }
        if (residual <= w->residual_threshold || it >= w->num_iterations - 1) { it++; break; }
    }
    o->last_iterations = it; o->last_residual = residual;
    if (w->manifold_mode == 1) for (int c = 0; c < o->ncp; c++) if (o->cp_foot[c] >= 0 && o->cp_man[c] >= 0)
        o->man[o->cp_foot[c]].imp[o->cp_man[c]] = nrm[c].applied;               /* solveGroupCacheFriendlyFinish: pt->m_appliedImpulse */
    /* contact force per foot (what a Gazebo bumper reports): normal + lateral impulses of its points / dt */
    memset(o->foot_force, 0, sizeof o->foot_force);
    for (int c = 0; c < o->ncp; c++) {
        if (o->cp_foot[c] < 0) continue;
        real *F = o->foot_force[o->cp_foot[c]];
        v3axpy(F, nrm[c].applied / dt, nrmW); v3axpy(F, fric[2 * c].applied / dt, dir1); v3axpy(F, fric[2 * c + 1].applied / dt, dir2);
    }
    /* processDeltaVeeMultiDof2 */
    for (int k = 0; k < NV; k++) {
        v[k] += dv[k];
        if (v[k] > w->max_coordinate_velocity) v[k] = w->max_coordinate_velocity;
        if (v[k] < -w->max_coordinate_velocity) v[k] = -w->max_coordinate_velocity;
    }
    v3cpy(o->base_omega, v); v3cpy(o->base_vel, v + 3);
    for (int d = 0; d < ND; d++) o->qd[d] = v[6 + d];

    /* ---- btMultiBody::stepPositionsMultiDof ---- */
    v3axpy(o->base_pos, dt, o->base_vel);
    {
        const real *angvel = o->base_omega;
        real fAngle = v3norm(angvel), axis[3];
        const real ANGULAR_MOTION_THRESHOLD = (real)(0.5 * 1.5707963267948966);
        if (fAngle * dt > ANGULAR_MOTION_THRESHOLD) fAngle = (real)0.5 * (real)1.5707963267948966 / dt;
        real k;
        if (fAngle < (real)0.001) k = (real)0.5 * dt - (dt * dt * dt) * (real)0.020833333333 * fAngle * fAngle;
        else k = (real)sin((double)((real)0.5 * fAngle * dt)) / fAngle;
        v3set(axis, angvel[0] * k, angvel[1] * k, angvel[2] * k);
        real dq[4] = {axis[0], axis[1], axis[2], (real)cos((double)(fAngle * dt * (real)0.5))};
        const real *q0 = o->base_quat; real r[4];
        /* r = dq * q0 */
        r[3] = dq[3] * q0[3] - dq[0] * q0[0] - dq[1] * q0[1] - dq[2] * q0[2];
        r[0] = dq[3] * q0[0] + dq[0] * q0[3] + dq[1] * q0[2] - dq[2] * q0[1];
        r[1] = dq[3] * q0[1] + dq[1] * q0[3] + dq[2] * q0[0] - dq[0] * q0[2];
        r[2] = dq[3] * q0[2] + dq[2] * q0[3] + dq[0] * q0[1] - dq[1] * q0[0];
        real n = (real)sqrt((double)(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]));
        for (int i = 0; i < 4; i++) o->base_quat[i] = r[i] / n;
    }
    for (int d = 0; d < ND; d++) o->q[d] += dt * o->qd[d];
}

/* ---------------------------------------------------------------- env level (plen_env.py) */
/* plen_env.py:694-714 */
static double agent_to_env(const real *range, double agent_val) {
    double lo = range[0], hi = range[1];
    double m = (hi - lo) / (1.0 - (-1.0));
    double b = hi - (m * 1.0);
    double v = m * agent_val + b;
    if (v >= hi) v = hi - 0.001; else if (v <= lo) v = lo + 0.001;
    return v;
}

static int gazebo_contact(const real *F);

/* plen_env.py:768-871 */
static void compute_observation(Oracle *o, real *obs) {
    real rpy[3]; quat_to_euler(rpy, o->base_quat);
    o->torso_z = o->base_pos[2]; o->torso_y = o->base_pos[1]; o->torso_vx = o->base_vel[0];
    o->roll = rpy[0]; o->pitch = rpy[1]; o->yaw = rpy[2];
    for (int d = 0; d < ND; d++) obs[d] = o->q[d];
    obs[18] = o->torso_z; obs[19] = o->torso_vx; obs[20] = o->roll; obs[21] = o->pitch; obs[22] = o->yaw;
    if (o->reward_head == 1) { o->right_contact = gazebo_contact(o->foot_force[0]); o->left_contact = gazebo_contact(o->foot_force[1]); }
    obs[23] = o->torso_y; obs[24] = (real)o->right_contact; obs[25] = (real)o->left_contact;
    static const int J[6] = {2, 8, 3, 9, 4, 10};
    if (o->nhist > 0) { for (int k = 0; k < 6; k++) o->diffs[k] = o->hist[k][o->nhist - 1] - o->q[J[k]]; o->first_pass = 0; }
    else { o->first_pass = 1; for (int k = 0; k < 6; k++) o->diffs[k] = 0; }
    if (o->nhist < HIST) { for (int k = 0; k < 6; k++) o->hist[k][o->nhist] = o->q[J[k]]; o->nhist++; }
}

/* plen_env.py:1072-1093 */
static int compute_done(Oracle *o) {
    const double PI3 = 3.14159265358979323846 / 3.0;
    int done = (o->roll > PI3) || (o->pitch > PI3) || (o->torso_z < (real)0.08) || (o->torso_y > 1);
    o->dead = done;
    return done;
}

static void clear_gait(Oracle *o) {
    o->nhist = 0; o->gait_period_counter = 0; o->double_support_counter = 0;
}

/* world orientation of a foot link as roll/pitch (plen_env.py:1016-1018, :1029-1031) */
static void foot_rp(const Oracle *o, int link, real *r, real *p) {
    real q[4], rpy[3]; mat_to_quat(q, o->Rw[link + 1]); quat_to_euler(rpy, q); *r = rpy[0]; *p = rpy[1];
}

/* plen_env.py:873-1070.  foot_roll/pitch are passed in so the function can be pinned against the
 * reference with arbitrary inputs. */
static double reward_core(Oracle *o, double lr, double lp, double rr, double rp) {
    double reward = 0;
    double vx = o->torso_vx;
    reward += 0.0;                                          /* alive_reward == 0 */
    if (vx < 0) reward -= exp(vx * 3.0); else reward += (vx * 3.0) * (vx * 3.0);
    { double h = fabs(0.160178937611 - (double)o->torso_z) * 40.0; reward -= h * h; }
    reward -= fabs((double)o->torso_y) * fabs((double)o->torso_y) * 1;
    reward -= fabs((double)o->roll) * fabs((double)o->roll) * 1.0;
    reward -= fabs((double)o->pitch) * fabs((double)o->pitch) * 0.5;
    reward -= fabs((double)o->yaw) * fabs((double)o->yaw) * 1.0;
    double jar = 0, jap = 0;
    const int gps = 80;
    if (o->gait_period_counter >= gps && o->right_contact == 1) {
        clear_gait(o);
    } else if (o->gait_period_counter >= 1.5 * gps) {
        reward -= 2;
    } else if (o->gait_period_counter > 0) {
        for (int pr = 0; pr < 3; pr++) {
            double dot = 0, nl = 0, nr = 0;
            for (int i = 0; i < o->nhist; i++) {
                double l = o->hist[2 * pr][i], r = o->hist[2 * pr + 1][i];
                dot += l * r; nl += l * l; nr += r * r;
            }
            jar += dot / (sqrt(nl) * sqrt(nr));
        }
        jar *= 1.0 / 3.0;
        if (!o->first_pass) {
            for (int k = 0; k < 6; k++) jap -= 1.0 / exp(fabs((double)o->diffs[k]));
            jap *= 0.5 * (1.0 / 3.0);
        }
    }
    reward += jar; reward += jap;
    if (o->left_contact == 1) {
        double x = (o->gait_period_counter * 10 / (double)gps) - 0.5 * 10;
        reward += 0.5 * (1 - tanh(x * x));
    }
    const double gpr = 0.1;
    if (o->gait_period_counter < gps / 2.0) {
        if (o->right_contact == 1 && o->left_contact == 0) reward += gpr;
        else if (o->right_contact == 0) reward -= gpr;
    } else if (o->gait_period_counter < gps) {
        if (o->left_contact == 1 && o->right_contact == 0) reward += gpr;
        else if (o->left_contact == 0) reward -= gpr;
    }
    if (o->right_contact == 1 && o->left_contact == 1) {
        o->double_support_counter += 1;
        if (o->double_support_counter >= 16) reward -= 2;
    }
    if (o->left_contact == 1 && fabs(lr) <= 0.1 && fabs(lp) <= 0.1) reward += 0.1;
    if (o->right_contact == 1 && fabs(rr) <= 0.1 && fabs(rp) <= 0.1) reward += 0.1;
    if (o->dead) { reward -= 100.0; o->dead = 0; }
    return reward;
}

/* ---- PlenWalkEnv-v0 (Gazebo) contract on the same state: plen_walk.py:346-396 contact rule, :597-618 done, :620-650 reward ---- */
static int gazebo_contact(const real *F) {
    double m = sqrt((double)F[0] * F[0] + (double)F[1] * F[1] + (double)F[2] * F[2]);
    return m > 4.8559 / 3.0;
}
static int gazebo_done(Oracle *o, double torso_x, int episode_timestep, int *dead) {
    const double PI3 = fabs(3.14159265358979323846 / 3.);
    int done;
    if (o->roll > PI3 || o->pitch > PI3 || o->torso_z < (real)0.08 || o->torso_y > 1) { done = 1; *dead = 1; }
    else if (episode_timestep > 500 && torso_x < 1) { done = 1; *dead = 0; }
    else { done = 0; *dead = 0; }
    return done;
}
static double gazebo_reward(const Oracle *o, int dead) {
    double reward = 0, vx = o->torso_vx;
    reward += 100. / 500;                                             /* alive_reward = dead_penalty / max_episode_steps */
    reward += (vx > 0 ? 1.0 : vx < 0 ? -1.0 : 0.0) * (vx * 3.) * (vx * 3.);
    { double h = fabs(0.158 - (double)o->torso_z) * 20.; reward -= h * h; }
    reward -= fabs((double)o->torso_y) * fabs((double)o->torso_y) * 1;
    reward -= fabs((double)o->roll) * fabs((double)o->roll) * 1.;
    reward -= fabs((double)o->pitch) * fabs((double)o->pitch) * 0.5;
    reward -= fabs((double)o->yaw) * fabs((double)o->yaw) * 1.;
    if (dead) reward -= 100.;
    return reward;
}

/* ---------------------------------------------------------------- exported C API */
#define API __attribute__((visibility("default")))

API int oracle_real_size(void) { return (int)sizeof(real); }

API Oracle *oracle_create(int joint_act) {
    Oracle *o = (Oracle *)calloc(1, sizeof(Oracle));
    o->joint_act = joint_act;
    world_defaults(&o->w, joint_act);
    o->mass[0] = (real)RAW_BASE_MASS;
    for (int k = 0; k < 3; k++) o->inertia[0][k] = (real)RAW_BASE_INERTIA[k];
    for (int i = 0; i < NL; i++) { o->mass[i + 1] = (real)RAW_MASS[i]; for (int k = 0; k < 3; k++) o->inertia[i + 1][k] = (real)RAW_INERTIA[i][k]; }
    o->base_quat[3] = 1; o->base_pos[2] = (real)0.158;
    o->first_pass = 1;
    return o;
}
API void oracle_destroy(Oracle *o) { free(o); }
/* whole-environment copy (physics, env-level and manifold state, parameters): finite differences of a step (tests/pybullet_pin.py) */
API void oracle_copy(Oracle *dst, const Oracle *src) { memcpy(dst, src, sizeof(Oracle)); }

/* domain randomisation hooks: scale every link mass (and inertia with it), override lateral friction */
API void oracle_set_params(Oracle *o, double mass_scale, double lateral_friction) {
    o->mass[0] = (real)(RAW_BASE_MASS * mass_scale);
    for (int k = 0; k < 3; k++) o->inertia[0][k] = (real)(RAW_BASE_INERTIA[k] * mass_scale);
    for (int i = 0; i < NL; i++) { o->mass[i + 1] = (real)(RAW_MASS[i] * mass_scale); for (int k = 0; k < 3; k++) o->inertia[i + 1][k] = (real)(RAW_INERTIA[i][k] * mass_scale); }
    if (lateral_friction >= 0) o->w.lateral_friction = (real)lateral_friction;
}
/* hypothesis sweeps (scripts/pin/): any World scalar by key, any link's inertia diagonal */
API int oracle_set_hyp(Oracle *o, int key, double v) {
    World *w = &o->w;
    switch (key) {
    case 0: w->erp = (real)v; break;            case 1: w->erp2 = (real)v; break;
    case 2: w->friction_erp = (real)v; break;   case 3: w->global_cfm = (real)v; break;
    case 4: w->linear_slop = (real)v; break;    case 5: w->residual_threshold = (real)v; break;
    case 6: w->restitution_velocity_threshold = (real)v; break;
    case 7: w->max_coordinate_velocity = (real)v; break;
    case 8: w->lateral_friction = (real)v; break;  case 9: w->box_lateral_friction = (real)v; break;
    case 10: w->spinning_friction = (real)v; break; case 11: w->rolling_friction = (real)v; break;
    case 12: w->restitution = (real)v; break;   case 13: w->linear_damping = (real)v; break;
    case 14: w->angular_damping = (real)v; break; case 15: w->motor_kp = (real)v; break;
    case 16: w->motor_kd = (real)v; break;      case 17: w->motor_max_force = (real)v; break;
    case 18: w->num_iterations = (int)v; break; case 19: w->body_contacts = (int)v; break;
    case 20: w->dt = (real)v; break;
    case 21: w->manifold_mode = (int)v; break;  case 22: w->warmstart = (real)v; break;
    case 23: w->pyramid_friction = (int)v; break; case 24: w->base_gyro_off = (int)v; break;
    case 25: w->torsional_points = (int)v; break;
    case 42: w->tors_freeze = (int)v; break;
    case 43: w->fric_order = (int)v; break; case 44: w->lever_on_plane = (int)v; break; case 45: w->man_key_ground = (int)v; break;
    case 30: w->man_cand = (int)v; break; case 31: w->man_drift = (real)v; break; case 32: w->man_add_all = (int)v; break;
    case 37: w->sole_grow = (real)v; break; case 38: w->sole_dz = (real)v; break;
    case 39: w->man_p1 = (int)v; break; case 40: w->man_p1x = (real)v; break; case 41: w->man_p1y = (real)v; break;
    case 33: w->man_fresh = (int)v; break; case 34: w->man_order = (int)v; break; case 35: w->man_cache = (real)v; break; case 36: w->man_range = (real)v; break;
    case 28: w->nc_order = (int)v; break; case 29: w->no_order_flip = (int)v; break;
    case 26: w->motor_rhs_clamp = (real)v; break; case 27: w->joint_damping = (real)v; break;
    default: return -1;
    }
    return 0;
}
API void oracle_set_link_inertia(Oracle *o, int b, double ixx, double iyy, double izz) {   /* b = 0 base, i + 1 link i */
    o->inertia[b][0] = (real)ixx; o->inertia[b][1] = (real)iyy; o->inertia[b][2] = (real)izz;
}
API void oracle_set_world(Oracle *o, int num_iterations, double residual_threshold) {
    if (num_iterations > 0) o->w.num_iterations = num_iterations;
    if (residual_threshold >= 0) o->w.residual_threshold = (real)residual_threshold;
}

/* state = pos3 quat4 omega3 vel3 q18 qd18 (49 doubles) */
API void oracle_get_state(const Oracle *o, double *s) {
    int k = 0;
    for (int i = 0; i < 3; i++) s[k++] = o->base_pos[i];
    for (int i = 0; i < 4; i++) s[k++] = o->base_quat[i];
    for (int i = 0; i < 3; i++) s[k++] = o->base_omega[i];
    for (int i = 0; i < 3; i++) s[k++] = o->base_vel[i];
    for (int i = 0; i < ND; i++) s[k++] = o->q[i];
    for (int i = 0; i < ND; i++) s[k++] = o->qd[i];
}
API void oracle_set_state(Oracle *o, const double *s) {
    int k = 0;
    for (int i = 0; i < 3; i++) o->base_pos[i] = (real)s[k++];
    for (int i = 0; i < 4; i++) o->base_quat[i] = (real)s[k++];
    for (int i = 0; i < 3; i++) o->base_omega[i] = (real)s[k++];
    for (int i = 0; i < 3; i++) o->base_vel[i] = (real)s[k++];
    for (int i = 0; i < ND; i++) o->q[i] = (real)s[k++];
    for (int i = 0; i < ND; i++) o->qd[i] = (real)s[k++];
}
/* aux = gait counter, double-support counter, episode step, dead, first_pass, nhist */
API void oracle_get_aux(const Oracle *o, int *a) {
    a[0] = o->gait_period_counter; a[1] = o->double_support_counter; a[2] = o->episode_timestep;
    a[3] = o->dead; a[4] = o->first_pass; a[5] = o->nhist;
}
API void oracle_set_targets(Oracle *o, const double *t) { for (int d = 0; d < ND; d++) o->target[d] = (real)t[d]; }
API void oracle_substep(Oracle *o) { substep(o); }
API void oracle_set_trace(double *buf, int cap_iterations) { g_trace = buf; g_trace_cap = cap_iterations; }   /* rows of 2 + 24 + 8 doubles */
API void oracle_contacts(const Oracle *o, int *flags) { flags[0] = o->right_contact; flags[1] = o->left_contact; flags[2] = o->ncp; flags[3] = o->last_iterations; }
/* contact slots of the last collision pass: box[8] = -2 empty, -1 foot point, else box index; pos[8][3]; also runs a collision pass on demand */
API void oracle_contact_slots(Oracle *o, int run_collide, int *box8, double *pos24) {
    if (run_collide) { fk(o); collide(o); }
    for (int c = 0; c < MAXCP; c++) { box8[c] = -2; for (int k = 0; k < 3; k++) pos24[3 * c + k] = 0; }
    for (int n = 0; n < o->ncp; n++) { int c = o->cp_slot[n]; box8[c] = o->cp_box[n]; for (int k = 0; k < 3; k++) pos24[3 * c + k] = o->cp_pos[n][k]; }
}
API void oracle_set_body_contacts(Oracle *o, int on) { o->w.body_contacts = on; }

/* low-level hooks for cross-checks in tests */
API void oracle_forward_dynamics(Oracle *o, double *qdd) {   /* accelerations at the current state, no constraints */
    real a[NV]; fk(o); aba(o, a); for (int k = 0; k < NV; k++) qdd[k] = a[k];
}
API void oracle_minv_times(Oracle *o, const double *force, double *out) {   /* M^-1 f (after forward_dynamics) */
    real f[NV], r[NV]; for (int k = 0; k < NV; k++) f[k] = (real)force[k];
    aba_delta(o, f, r); for (int k = 0; k < NV; k++) out[k] = r[k];
}
API void oracle_link_frames(Oracle *o, double *R, double *O, double *C) {  /* [33][9], [33][3], [33][3] */
    fk(o);
    for (int b = 0; b <= NL; b++) { for (int k = 0; k < 9; k++) R[9 * b + k] = o->Rw[b][k]; for (int k = 0; k < 3; k++) { O[3 * b + k] = o->Ow[b][k]; C[3 * b + k] = o->Cw[b][k]; } }
}

API double oracle_agent_to_env(int joint, double a) { return agent_to_env(ENV_RANGES[joint], a); }

/* reward/done pin: drive the env-level state machine with externally supplied observations
 * (joint angles, base pose, contacts, foot roll/pitch) exactly as tools/make_golden.py drives the
 * reference through stubbed pybullet calls.  Mirrors step() order: obs -> done -> reward -> counter++ */
API void oracle_script_reset(Oracle *o) {
    clear_gait(o); o->episode_timestep = 0;   /* dead and first_pass are NOT reset (plen_env.py:576-590) */
}
API double oracle_script_step(Oracle *o, const double *q18, double z, double vx, double roll, double pitch, double yaw, double y,
                              int rc, int lc, double lroll, double lpitch, double rroll, double rpitch, int *done_out) {
    /* install the scripted readings where compute_observation would read them from the engine */
    for (int d = 0; d < ND; d++) o->q[d] = (real)q18[d];
    o->right_contact = rc; o->left_contact = lc;
    o->torso_z = (real)z; o->torso_vx = (real)vx; o->roll = (real)roll; o->pitch = (real)pitch; o->yaw = (real)yaw; o->torso_y = (real)y;
    static const int J[6] = {2, 8, 3, 9, 4, 10};
    if (o->nhist > 0) { for (int k = 0; k < 6; k++) o->diffs[k] = o->hist[k][o->nhist - 1] - o->q[J[k]]; o->first_pass = 0; }
    else { o->first_pass = 1; for (int k = 0; k < 6; k++) o->diffs[k] = 0; }
    if (o->nhist < HIST) { for (int k = 0; k < 6; k++) o->hist[k][o->nhist] = o->q[J[k]]; o->nhist++; }
    int done = compute_done(o);
    double r = reward_core(o, lroll, lpitch, rroll, rpitch);
    o->episode_timestep += 1; o->gait_period_counter += 1;
    *done_out = done;
    return r;
}

/* plen_env.py:558-614 */
API void oracle_reset(Oracle *o, double *obs_out) {
    v3set(o->base_pos, 0, 0, (real)0.158);
    o->base_quat[0] = o->base_quat[1] = o->base_quat[2] = 0; o->base_quat[3] = 1;
    v3set(o->base_omega, 0, 0, 0); v3set(o->base_vel, 0, 0, 0);
    for (int d = 0; d < ND; d++) { o->q[d] = 0; o->qd[d] = 0; o->target[d] = 0; }
    o->man[0].n = o->man[1].n = 0;            /* the teleport moves every cached point beyond its breaking threshold */
    for (int i = 0; i < 8; i++) substep(o);                 /* 2 * sim_stepsize */
    real obs[26]; fk(o); compute_observation(o, obs);
    o->episode_timestep = 0;
    clear_gait(o);
    if (obs_out) for (int k = 0; k < 26; k++) obs_out[k] = obs[k];
}

/* plen_env.py:638-692.  action is float32-valued (Box dtype) carried in doubles.  Returns reward. */
API double oracle_step(Oracle *o, const double *action, double *obs_out, int *done_out) {
    for (int d = 0; d < ND; d++)
        o->target[d] = (real)(o->joint_act ? action[d] : agent_to_env(ENV_RANGES[d], action[d]));
    for (int i = 0; i < 4; i++) substep(o);
    real obs[26]; fk(o); compute_observation(o, obs);
    int done; double reward;
    if (o->reward_head == 1) {
        int dead; done = gazebo_done(o, (double)o->base_pos[0], o->episode_timestep, &dead); o->dead = dead;
        reward = gazebo_reward(o, dead);
    } else {
        done = compute_done(o);
        real lr, lp, rr, rp;
        foot_rp(o, RAW_LFOOT_LINK, &lr, &lp); foot_rp(o, RAW_RFOOT_LINK, &rr, &rp);
        reward = reward_core(o, lr, lp, rr, rp);
    }
    o->episode_timestep += 1; o->gait_period_counter += 1;
    for (int k = 0; k < 26; k++) obs_out[k] = obs[k];
    *done_out = done;
    return reward;
}

/* gym TimeLimit(500) + the driver's auto-reset (plen_td3.py:108-133), for the CPU baseline and the
 * rollout parity tests: runs `nsteps` control steps of one env, resetting on done/time-limit.
 * actions [nsteps][18]; outputs obs [nsteps][26] (post-step, pre-reset), rew, done flags
 * (bit0 = terminal, bit1 = time-limit). Returns the number of env steps executed. */
API int oracle_rollout(Oracle *o, int nsteps, const float *actions, double *obs, double *rew, uint8_t *flags) {
    double a[ND], ob[26];
    for (int t = 0; t < nsteps; t++) {
        for (int d = 0; d < ND; d++) a[d] = actions[t * ND + d];
        int done; double r = oracle_step(o, a, ob, &done);
        int trunc = o->episode_timestep >= 500;
        if (obs) memcpy(obs + 26 * t, ob, sizeof ob);
        if (rew) rew[t] = r;
        if (flags) flags[t] = (uint8_t)((done ? 1 : 0) | (trunc ? 2 : 0));
        if (done || trunc) oracle_reset(o, NULL);
    }
    return nsteps;
}

/* ---------------------------------------------------------------- CPU baseline (bench.py cpu_baseline leg)
 * n_envs environments stepped with uniform random actions (xorshift per env, float32-valued like the Box) and the
 * driver's auto-reset, whole vector steps until `budget_s` seconds have passed.  With OpenMP (the native build,
 * gcc -O3 -march=native -fopenmp) the envs of a vector step are partitioned over `threads` threads. Returns env-steps. */
#ifdef _OPENMP
#include <omp.h>
#else
#include <time.h>
#endif
static double wall_now(void) {
#ifdef _OPENMP
    return omp_get_wtime();
#else
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec;
#endif
}
API long long oracle_throughput(int n_envs, int threads, double budget_s, unsigned seed, double *seconds_out, int *vector_steps_out) {
    Oracle **envs = (Oracle **)calloc((size_t)n_envs, sizeof(Oracle *));
    uint64_t *rng = (uint64_t *)calloc((size_t)n_envs, sizeof(uint64_t));
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
#endif
    for (int e = 0; e < n_envs; e++) { envs[e] = oracle_create(0); oracle_reset(envs[e], NULL); rng[e] = 0x9E3779B97F4A7C15ull * (uint64_t)(seed + 1 + e) | 1ull; }
    int vsteps = 0;
    (void)threads;
    const double t0 = wall_now();
    double t1 = t0;
    do {
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
#endif
        for (int e = 0; e < n_envs; e++) {
            double a[ND], ob[26]; int done;
            uint64_t x = rng[e];
            for (int d = 0; d < ND; d++) {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                a[d] = (double)(float)(((double)(x >> 11) * (1.0 / 9007199254740992.0)) * 2.0 - 1.0);
            }
            rng[e] = x;
            oracle_step(envs[e], a, ob, &done);
            if (done || envs[e]->episode_timestep >= 500) oracle_reset(envs[e], NULL);
        }
        vsteps++;
        t1 = wall_now();
    } while (t1 - t0 < budget_s);
    for (int e = 0; e < n_envs; e++) oracle_destroy(envs[e]);
    free(envs); free(rng);
    if (seconds_out) *seconds_out = t1 - t0;
    if (vector_steps_out) *vector_steps_out = vsteps;
    return (long long)vsteps * n_envs;
}

/* ---------------------------------------------------------------- full-size parity helper (tests): n_envs independent environments,
 * each created, given its own parameters, reset and rolled out for T control steps with the driver's auto-reset (oracle_rollout), the
 * envs partitioned over `threads` POSIX threads.  actions [T][n_envs][18] float32; outputs obs [T][n_envs][26], rew [T][n_envs],
 * flags [T][n_envs].  mass_scale / lateral_friction: per-env arrays or NULL.  rolling < 0 keeps the reference's rolling friction. */
#include <pthread.h>
typedef struct { int e0, e1, n, T; const float *act; const double *ms, *mu; double rolling; int body_contacts; double *obs, *rew; uint8_t *flags; } BatchJob;
static void *batch_worker(void *arg) {
    BatchJob *j = (BatchJob *)arg;
    for (int e = j->e0; e < j->e1; e++) {
        Oracle *o = oracle_create(0);
        if (j->ms || j->mu) oracle_set_params(o, j->ms ? j->ms[e] : 1.0, j->mu ? j->mu[e] : -1.0);
        if (j->rolling >= 0) o->w.rolling_friction = (real)j->rolling;
        o->w.body_contacts = j->body_contacts;
        oracle_reset(o, NULL);
        for (int t = 0; t < j->T; t++) {
            double a[ND], ob[26]; int done;
            for (int d = 0; d < ND; d++) a[d] = j->act[((size_t)t * j->n + e) * ND + d];
            double r = oracle_step(o, a, ob, &done);
            int trunc = o->episode_timestep >= 500;
            memcpy(j->obs + ((size_t)t * j->n + e) * 26, ob, sizeof ob);
            j->rew[(size_t)t * j->n + e] = r;
            j->flags[(size_t)t * j->n + e] = (uint8_t)((done ? 1 : 0) | (trunc ? 2 : 0));
            if (done || trunc) oracle_reset(o, NULL);
        }
        oracle_destroy(o);
    }
    return NULL;
}
API void oracle_batch_rollout(int n_envs, int T, const float *actions, const double *mass_scale, const double *lateral_friction, double rolling,
                              int body_contacts, int threads, double *obs, double *rew, uint8_t *flags) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256]; BatchJob jobs[256];
    int per = (n_envs + threads - 1) / threads;
    for (int k = 0; k < threads; k++) {
        BatchJob *j = &jobs[k];
        j->e0 = k * per < n_envs ? k * per : n_envs; j->e1 = (k + 1) * per < n_envs ? (k + 1) * per : n_envs;
        j->n = n_envs; j->T = T; j->act = actions; j->ms = mass_scale; j->mu = lateral_friction; j->rolling = rolling; j->body_contacts = body_contacts;
        j->obs = obs; j->rew = rew; j->flags = flags;
        pthread_create(&th[k], NULL, batch_worker, j);
    }
    for (int k = 0; k < threads; k++) pthread_join(th[k], NULL);
}


/* ---------------------------------------------------------------- closed-loop ensembles (tests/test_distribution_gpu.py, scripts/pin/): n_episodes
 * independent episodes from reset, each until done or the 500-step limit, the envs partitioned over POSIX threads.  Actions: the actor MLP
 * 26-256-256-18 (td3.py:19-57; weights row major [out][in] as torch stores them, NULL = uniform random actions U[-1, 1) like the benchmark's)
 * plus N(0, sigma) noise clipped to [-1, 1] (plen_td3.py:101-104) and rounded to float32 (the Box dtype).  Per-episode xorshift streams seeded
 * from `seed`: a statistical sample, not a replay of anybody else's noise.  hyp_keys / hyp_vals: oracle_set_hyp switches applied to every env. */
typedef struct { int e0, e1; const double *W1, *b1, *W2, *b2, *W3, *b3; double sigma; unsigned seed; int nh; const int *hk; const double *hv;
                 int *len; double *ret; } EnsJob;
static double ens_uniform(uint64_t *x) { *x ^= *x << 13; *x ^= *x >> 7; *x ^= *x << 17; return (double)(*x >> 11) * (1.0 / 9007199254740992.0); }
static double ens_normal(uint64_t *x) {
    double u1 = ens_uniform(x), u2 = ens_uniform(x);
    if (u1 < 1e-300) u1 = 1e-300;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}
static void *ens_worker(void *arg) {
    EnsJob *j = (EnsJob *)arg;
    Oracle *o = oracle_create(0);
    for (int k = 0; k < j->nh; k++) oracle_set_hyp(o, j->hk[k], j->hv[k]);
    double *h1 = (double *)malloc(256 * sizeof(double)), *h2 = (double *)malloc(256 * sizeof(double));
    for (int e = j->e0; e < j->e1; e++) {
        uint64_t rng = 0x9E3779B97F4A7C15ull * (uint64_t)(j->seed + 1u + (unsigned)e) | 1ull;
        for (int w = 0; w < 8; w++) ens_uniform(&rng);
        double ob[26], a[ND], ret = 0; int done = 0, t = 0;
        oracle_reset(o, ob);
        for (t = 0; t < 500 && !done; t++) {
            if (j->W1) {
                for (int i = 0; i < 256; i++) { double s = j->b1[i]; for (int k = 0; k < 26; k++) s += j->W1[i * 26 + k] * ob[k]; h1[i] = s > 0 ? s : 0; }
                for (int i = 0; i < 256; i++) { double s = j->b2[i]; for (int k = 0; k < 256; k++) s += j->W2[i * 256 + k] * h1[k]; h2[i] = s > 0 ? s : 0; }
                for (int i = 0; i < ND; i++) { double s = j->b3[i]; for (int k = 0; k < 256; k++) s += j->W3[i * 256 + k] * h2[k]; a[i] = tanh(s); }
            } else for (int i = 0; i < ND; i++) a[i] = ens_uniform(&rng) * 2.0 - 1.0;
            for (int i = 0; i < ND; i++) {
                if (j->W1 && j->sigma > 0) a[i] += j->sigma * ens_normal(&rng);
                if (a[i] > 1) a[i] = 1;
                if (a[i] < -1) a[i] = -1;
                a[i] = (double)(float)a[i];
            }
            ret += oracle_step(o, a, ob, &done);
        }
        j->len[e] = t; j->ret[e] = ret;
    }
    free(h1); free(h2); oracle_destroy(o);
    return NULL;
}
API void oracle_ensemble(int n_episodes, const double *W1, const double *b1, const double *W2, const double *b2, const double *W3, const double *b3,
                         double sigma, unsigned seed, int threads, int n_hyp, const int *hyp_keys, const double *hyp_vals, int *lengths, double *returns) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256]; EnsJob jobs[256];
    int per = (n_episodes + threads - 1) / threads;
    for (int k = 0; k < threads; k++) {
        EnsJob *j = &jobs[k];
        j->e0 = k * per < n_episodes ? k * per : n_episodes; j->e1 = (k + 1) * per < n_episodes ? (k + 1) * per : n_episodes;
        j->W1 = W1; j->b1 = b1; j->W2 = W2; j->b2 = b2; j->W3 = W3; j->b3 = b3; j->sigma = sigma; j->seed = seed; j->nh = n_hyp; j->hk = hyp_keys; j->hv = hyp_vals;
        j->len = lengths; j->ret = returns;
        pthread_create(&th[k], NULL, ens_worker, j);
    }
    for (int k = 0; k < threads; k++) pthread_join(th[k], NULL);
}

API double oracle_last_residual(const Oracle *o) { return (double)o->last_residual; }
API void oracle_set_reward_head(Oracle *o, int head) { o->reward_head = head; }
API void oracle_foot_forces(const Oracle *o, double *f6) { for (int i = 0; i < 6; i++) f6[i] = o->foot_force[i / 3][i % 3]; }
/* pins of the PlenWalkEnv-v0 head against tests/golden/gazebo_reward_done.npz */
API int oracle_gazebo_contact(const double *force3) { real F[3] = {(real)force3[0], (real)force3[1], (real)force3[2]}; return gazebo_contact(F); }
API double oracle_gazebo_script(Oracle *o, double vx, double z, double y, double roll, double pitch, double yaw, double x, int episode_timestep,
                                int *done_out, int *dead_out) {
    o->torso_vx = (real)vx; o->torso_z = (real)z; o->torso_y = (real)y; o->roll = (real)roll; o->pitch = (real)pitch; o->yaw = (real)yaw;
    *done_out = gazebo_done(o, x, episode_timestep, dead_out);
    return gazebo_reward(o, *dead_out);
}

/* single-state pin of compute_done + compute_reward (tests/golden/a78_reward_done.npz): install the
 * env attributes the reference's test harness set by hand, then run done -> reward. */
API void oracle_script_inject(Oracle *o, int cnt, int ds, int nh, const double *hist, const double *diffs, int first_pass) {
    o->gait_period_counter = cnt; o->double_support_counter = ds; o->nhist = nh; o->first_pass = first_pass;
    for (int k = 0; k < 6; k++) { o->diffs[k] = (real)diffs[k]; for (int i = 0; i < nh; i++) o->hist[k][i] = (real)hist[k * nh + i]; }
}
API double oracle_script_reward(Oracle *o, double z, double vx, double roll, double pitch, double yaw, double y, int rc, int lc,
                                double lroll, double lpitch, double rroll, double rpitch, int *done_out) {
    o->right_contact = rc; o->left_contact = lc;
    o->torso_z = (real)z; o->torso_vx = (real)vx; o->roll = (real)roll; o->pitch = (real)pitch; o->yaw = (real)yaw; o->torso_y = (real)y;
    *done_out = compute_done(o);
    return reward_core(o, lroll, lpitch, rroll, rpitch);
}

/* contact-parameter overrides for the well-conditioned parity configurations used by the tests
 * (negative = keep).  Coefficients are the COMBINED ones the solver uses. */
API void oracle_set_friction(Oracle *o, double lateral, double spinning, double rolling) {
    if (lateral >= 0) o->w.lateral_friction = (real)lateral;
    if (spinning >= 0) o->w.spinning_friction = (real)spinning;
    if (rolling >= 0) o->w.rolling_friction = (real)rolling;
}
