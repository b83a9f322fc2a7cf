/*
 * plenvec.h -- C ABI of libplenvec.so, the MI355X-native vectorised PLEN walking environment.
 *
 * The reference has no C/FFI interface of its own: its hot path is the Python class
 * PlenWalkEnv (plen_bullet/src/plen_bullet/plen_env.py) calling the third-party `pybullet`
 * C extension.  Each entry point below therefore cites the reference *Python call site(s)* whose
 * work it replaces; the Python facades in plen_ml_walk_amd/ (same names/kwargs as the reference)
 * bind these symbols with ctypes (INTEGRATION.md shows the binding a maintainer would add).
 *
 * Conventions
 *   - all functions return 0 on success, a negative PLENVEC_E_* code on failure;
 *     plenvec_last_error() returns a human-readable message for the calling thread;
 *   - every array argument is a caller-owned DEVICE pointer unless its name ends in _host;
 *   - work is enqueued on the hipStream_t passed in (void* here so the header needs no HIP);
 *     nothing synchronises with the host except create/destroy and the *_host helpers;
 *   - a handle is not thread-safe: one stream per handle at a time;
 *   - real-valued device arrays are float or double according to PlenCfg.dtype.
 *
 * Deviations from the sketch in SURVEY.md section 8(b), on purpose:
 *   - plenvec_create takes no model: the PLEN tables are compiled in (generated from the reference's plen.urdf + foot STLs).
 *     plenvec_create_from_model takes a PlenModel for variants of the same tree (other masses, inertias, frames, soles, boxes);
 *   - plenvec_step reports truncation as a BIT of `done` (PLENVEC_DONE_TIMELIMIT) instead of a separate trunc[] array, because the
 *     non-finite guard needs a third state; plenvec_step2 returns the sketch's (done, trunc) pair of 0/1 bytes;
 *   - there is no device = -1 CPU backend: the only CPU implementation of this path is the test oracle, and a product path through it
 *     would void every parity claim.  Without a GPU every entry point fails with PLENVEC_E_NODEV.
 */
#ifndef PLENVEC_H
#define PLENVEC_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PLENVEC_OK 0
#define PLENVEC_E_INVAL -1     /* bad argument */
#define PLENVEC_E_HIP -2       /* HIP runtime error (message has the HIP error string) */
#define PLENVEC_E_NODEV -3     /* no usable GPU: the library never falls back to the CPU */

#define PLENVEC_OBS 26         /* plen_env.py:807-822 */
#define PLENVEC_ACT 18         /* plen_env.py:142-144 */
#define PLENVEC_STATE 49       /* pos3 quat4(x,y,z,w) omega3 vel3 q18 qd18 */
#define PLENVEC_DTYPE_F32 0
#define PLENVEC_DTYPE_F64 1

/* done flags written by plenvec_step */
#define PLENVEC_DONE_TERMINAL 1   /* compute_done() fired (plen_env.py:1072-1093) */
#define PLENVEC_DONE_TIMELIMIT 2  /* gym TimeLimit: episode step reached max_episode_steps (plen_env.py:15-19) */
#define PLENVEC_DONE_NONFINITE 4  /* cfg.nonfinite_guard: the step produced a NaN/inf (state, observation or reward); the env was put back
                                     into its reset state (even with auto_reset = 0), next_obs = the reset observation, reward = 0, and the
                                     TIMELIMIT bit is set with it so that callers treat the transition as a truncation, never as a terminal.
                                     The reference's only guard of this kind is robot_gazebo_env.py:182-185 (NaN reward -> shut down). */

typedef struct plenvec plenvec_t;

/* World / env configuration.  Zero-initialise, call plenvec_default_cfg(), then override.
 * Defaults reproduce plen_env.py:34-556 with PyBullet's own defaults for what the reference leaves
 * unset (DESIGN.md lists each with its source). */
typedef struct PlenCfg {
    int32_t dtype;               /* PLENVEC_DTYPE_F32 (default) or _F64 */
    int32_t joint_act;           /* PlenWalkEnv(joint_act=...) plen_env.py:34,439-442,472-475,652-654 */
    int32_t max_episode_steps;   /* 500, register() plen_env.py:15-19 */
    int32_t substeps;            /* 4 = sim_stepsize, plen_env.py:40-42 */
    int32_t reset_substeps;      /* 8 = 2*sim_stepsize, plen_env.py:569-570 */
    int32_t num_iterations;      /* 50 */
    int32_t auto_reset;          /* 1: an env whose episode ended restarts inside the same step call */
    int32_t reward_head;         /* 0: PlenWalkEnv-v1 reward/done/contact flags (plen_env.py, default); 1: the PlenWalkEnv-v0 contract on the same
                                    physics: contact = |foot force| > weight/3 (plen_walk.py:346-396), done :597-618, reward :620-650 */
    double dt;                   /* 1/240 */
    double gravity_z;            /* -9.81, plen_env.py:296 */
    double erp, erp2;            /* 0.2, 0.08 */
    double linear_slop;          /* 1e-5 */
    double residual_threshold;   /* 1e-7 */
    double restitution_velocity_threshold; /* 0.2 */
    double max_coordinate_velocity;        /* 100 */
    double lateral_friction;     /* 0.8*0.8, plen_env.py:309,444 */
    double spinning_friction;    /* 0.1*0.8, plen_env.py:445 */
    double rolling_friction;     /* 0.1*0.8 (0.01*0.8 if joint_act), plen_env.py:439-442 */
    double restitution;          /* 0.5*0.5, plen_env.py:309,481 */
    double linear_damping;       /* 0 (0.1 if joint_act), plen_env.py:472-480 */
    double motor_kp, motor_kd;   /* 0.1, 1.0 (PyBullet POSITION_CONTROL defaults) */
    double motor_max_force;      /* 0.15, plen_env.py:753 */
    double spawn_z;              /* 0.158, plen_env.py:312 */
    double box_lateral_friction; /* 0.5*0.8: links other than the feet keep Bullet's URDF default friction 0.5 (only links 11, 19 are changed, plen_env.py:439-467) */
    int32_t nonfinite_guard;     /* 1 (default): per-env NaN/inf guard in the step epilogue, see PLENVEC_DONE_NONFINITE; 0: NaNs propagate like in the reference */
    int32_t body_contacts;       /* 1 (default): the box colliders of the 31 non-foot links (plen.urdf:504-1274) collide with the ground like in the
                                    reference (plen_env.py:306-315): a box corner near the ground takes a contact slot whose foot point is out of range;
                                    0: only the feet collide */
} PlenCfg;

/* Fills *cfg with the reference configuration for the given joint_act mode. */
int plenvec_default_cfg(PlenCfg *cfg, int joint_act);

/* The robot as numbers (what loadURDF("plen.urdf") hands PyBullet, plen_env.py:312-315, after the importer's rules: fixed joints folded
 * into 19 composite bodies, inertia from the collision shapes).  The TOPOLOGY is fixed -- the kernels are written for the PLEN tree: body 0
 * the base, four serial chains (right leg 1-6, left leg 7-12, right arm 13-15, left arm 16-18), feet = bodies 6 and 12 -- every number may
 * change (a plen_new.urdf-style variant with other masses / link geometry, a measured robot, ...). */
#define PLENVEC_NBODY 19
#define PLENVEC_MAXMEMB 7
#define PLENVEC_MAXBOX 31
typedef struct PlenModel {
    int32_t num_bodies;                         /* 19 */
    int32_t parent[PLENVEC_NBODY];              /* must be {-1, 0,1,2,3,4,5, 0,7,8,9,10,11, 0,13,14, 0,16,17} */
    double joint_R[PLENVEC_NBODY][9];           /* joint frame in the parent body frame, row major */
    double joint_t[PLENVEC_NBODY][3];
    double axis[PLENVEC_NBODY][3];              /* revolute axis in the body frame (unit) */
    double com[PLENVEC_NBODY][3];               /* composite centre of mass, body frame */
    double inertia[PLENVEC_NBODY][6];           /* about the COM, body axes: xx yy zz xy xz yz */
    double mass[PLENVEC_NBODY];
    int32_t n_member[PLENVEC_NBODY];            /* links folded into the body (Bullet applies linear damping per LINK) */
    double member_com[PLENVEC_NBODY][PLENVEC_MAXMEMB][3];
    double member_mass[PLENVEC_NBODY][PLENVEC_MAXMEMB];
    double margin;                              /* collision margin of the foot hulls (0.001) */
    double foot_break[2];                       /* contact breaking threshold, right / left foot */
    double sole[2][32][3];                      /* the 32 sole-plane hull vertices per foot, foot body frame, outline order */
    int32_t sole_rep[2][32];                    /* 1: the vertex is a contact candidate (one per corner fillet) */
    int32_t sole_order[2][4][32];               /* per sole diagonal: vertex indices by descending key (candidates first) */
    int32_t num_boxes;                          /* <= 31 box colliders of the non-foot links */
    int32_t box_body[PLENVEC_MAXBOX];
    double box_R[PLENVEC_MAXBOX][9], box_t[PLENVEC_MAXBOX][3], box_half[PLENVEC_MAXBOX][3];   /* pose in the body frame, half extents */
    double box_break[PLENVEC_MAXBOX];           /* contact breaking threshold */
    double box_link_restitution[PLENVEC_MAXBOX];/* 0.5 (plen_env.py:472-481), base link 0 */
} PlenModel;
/* What a PlenModel can NOT change (compiled into the kernels): the tree topology, which bodies are the feet (6 and 12), the joint
 * limits (+-1.7 rad, plen.urdf:1310 -- every revolute joint of the PLEN has the same), the solver's row order.  plenvec_create_from_model
 * rejects (PLENVEC_E_INVAL) non-finite numbers, a non-orthonormal joint_R, an inertia that is not positive definite, member masses that do
 * not sum to the body's mass and sole_rep outside {0, 1}. */
/* Fills *model with the compiled-in PLEN robot (plen.urdf + rfoot.stl / lfoot.stl through tools/extract_model.py). */
int plenvec_default_model(PlenModel *model);

/* Replaces: PlenWalkEnv.__init__ (plen_env.py:34-556: connect, loadURDF, changeDynamics...) for
 * `num_envs` independent environments on HIP device `device`.  The PLEN model tables are compiled
 * in (generated from the reference's plen.urdf + foot STLs by tools/extract_model.py).
 * All envs start in the post-reset state (see plenvec_reset).  Host-synchronous. */
int plenvec_create(const PlenCfg *cfg, int num_envs, int device, plenvec_t **out);
/* The same with the robot given as numbers; PLENVEC_E_INVAL if the topology is not the PLEN tree.  plenvec_create(cfg, ...) ==
 * plenvec_create_from_model(default model, cfg, ...). */
int plenvec_create_from_model(const PlenModel *model, const PlenCfg *cfg, int num_envs, int device, plenvec_t **out);
int plenvec_destroy(plenvec_t *h);
int plenvec_num_envs(const plenvec_t *h);
int plenvec_dtype(const plenvec_t *h);

/* Replaces: PlenWalkEnv.reset (plen_env.py:558-614): base pose/joints zeroed, zero motor targets,
 * 8 settle substeps, observation, gait/episode counters cleared.
 * mask: uint8[num_envs] device pointer, nonzero = reset that env; NULL = all envs.
 * obs: real[num_envs][26] device pointer or NULL; rows of envs that were not reset are left untouched. */
int plenvec_reset(plenvec_t *h, const uint8_t *mask, void *obs, void *stream);

/* Replaces: PlenWalkEnv.step (plen_env.py:638-692) = agent_to_env (:694-714) + move_joints
 * (:716-753) + 4 x p.stepSimulation() (:665-667) + compute_observation (:768-871) + compute_done
 * (:1072-1093) + compute_reward (:873-1070) + counters (:674-678), plus gym's TimeLimit wrapper and
 * the driver's reset-on-done (plen_td3.py:108-133) when cfg.auto_reset is set.
 *   action   float[num_envs][18]   agent actions in [-1,1] (always float32, the Box dtype)
 *   next_obs real[num_envs][26]    observation after the step (terminal observation if the episode ended)
 *   reward   real[num_envs]
 *   done     uint8[num_envs]       PLENVEC_DONE_* bits
 *   cur_obs  real[num_envs][26]    observation to act on next: == next_obs unless the env was
 *                                  auto-reset, then the reset observation.  May be NULL. */
int plenvec_step(plenvec_t *h, const float *action, void *next_obs, void *reward, uint8_t *done, void *cur_obs, void *stream);
/* plenvec_step with SURVEY 8(b)'s output pair: done[e] = 1 iff compute_done() fired and the time limit did not (the `done_bool` the
 * reference driver stores, plen_td3.py:109-110), trunc[e] = 1 iff the episode was cut by the time limit (or by the non-finite guard).
 * One extra tiny launch on the same stream; the hot loops of this repository use plenvec_step. */
int plenvec_step2(plenvec_t *h, const float *action, void *next_obs, void *reward, uint8_t *done, uint8_t *trunc, void *cur_obs, void *stream);

/* State injection / extraction for parity tests (no reference equivalent; PyBullet's
 * resetBasePositionAndOrientation/resetJointState/getJointStates family).
 * state: real[num_envs][49].  set_state also clears the gait bookkeeping like reset() does. */
int plenvec_get_state(plenvec_t *h, void *state, void *stream);
int plenvec_set_state(plenvec_t *h, const void *state, void *stream);
/* aux: int32[num_envs][8] = gait counter, double-support counter, episode step, history length,
 * right contact, left contact, solver iterations of the last substep, issue-slot estimate of the last step
 * (internal: plenvec_step places envs on SIMDs by it; any value is legal, results do not depend on it) */
int plenvec_get_aux(plenvec_t *h, int32_t *aux, void *stream);

/* Test hook: run `nsub` raw physics substeps (p.stepSimulation(), plen_env.py:667) with the given
 * motor targets (real[num_envs][18], radians, i.e. AFTER agent_to_env), no observation/reward
 * bookkeeping.  dump: optional real[num_envs][PLENVEC_DUMP] of solver intermediates of the LAST substep.  Afterwards aux[4..6] hold
 * the contact flags and iteration count of the last substep and aux[7] its contact-slot masks (bits 0-7: slots lent to box corners of
 * non-foot links, bits 8-15: occupied slots). */
#define PLENVEC_DUMP 4096
int plenvec_debug_substeps(plenvec_t *h, const void *targets, int nsub, void *dump, void *stream);

/* Domain randomisation (BASELINE.json configs[4]; not in the reference): per-env scale of every
 * link mass (inertia scales with it) and per-env foot/ground lateral friction coefficient.
 * real[num_envs] each; NULL leaves that parameter unchanged.  The dynamics use the new values from the next
 * plenvec_step on; the per-env reset record (the settled stance depends on mass and friction) is re-simulated on the
 * given stream at the start of the next plenvec_step or plenvec_reset, WITHOUT touching the live state of any env, so
 * auto-resets and masked resets after set_params always restore a state settled with the current parameters.
 * (Call it outside hipGraph capture and run one step or reset before capturing.) */
int plenvec_set_params(plenvec_t *h, const void *mass_scale, const void *lateral_friction, void *stream);

/* Number of PLENVEC_DONE_NONFINITE events since create (all envs).  Host-synchronous on `stream` (diagnostic). */
int plenvec_get_nonfinite_count(plenvec_t *h, int64_t *count_host, void *stream);

/* Kernel timing hook for bench.py: HIP-event time of the step kernel launches on the stream the
 * kernel was launched on.  begin() records, end() records + synchronises and returns the elapsed
 * milliseconds and the number of step kernels launched in between (`launches` counts host-side launch calls: under hipGraph
 * replay it counts captures, not replays -- time graphs with your own events). */
int plenvec_timing_begin(plenvec_t *h, void *stream);
int plenvec_timing_end(plenvec_t *h, void *stream, double *elapsed_ms, int64_t *launches);

const char *plenvec_last_error(void);
const char *plenvec_version(void);

#ifdef __cplusplus
}
#endif
#endif
