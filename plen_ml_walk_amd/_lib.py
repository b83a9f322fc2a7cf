"""ctypes binding of libplenvec.so (include/plenvec.h).  No torch types cross this boundary: only
raw device pointers, sizes and a stream handle.  The library is built in-tree by
__graft_entry__.build() / plen_ml_walk_amd.build.build_extension(); importing fails loudly if
the shared object is missing -- there is no Python or CPU fallback for the env step."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PLENVEC_LIB: developer override used by scripts/gpu_ab.py to time two builds of the kernel side by side
LIB_PATH = os.environ.get("PLENVEC_LIB") or os.path.join(_HERE, "csrc", "libplenvec.so")

OBS, ACT, STATE, DUMP = 26, 18, 49, 4096
DTYPE_F32, DTYPE_F64 = 0, 1
DONE_TERMINAL, DONE_TIMELIMIT, DONE_NONFINITE = 1, 2, 4


class PlenCfg(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("joint_act", C.c_int32), ("max_episode_steps", C.c_int32), ("substeps", C.c_int32),
                ("reset_substeps", C.c_int32), ("num_iterations", C.c_int32), ("auto_reset", C.c_int32), ("reward_head", C.c_int32),
                ("dt", C.c_double), ("gravity_z", C.c_double), ("erp", C.c_double), ("erp2", C.c_double),
                ("linear_slop", C.c_double), ("residual_threshold", C.c_double), ("restitution_velocity_threshold", C.c_double),
                ("max_coordinate_velocity", C.c_double), ("lateral_friction", C.c_double), ("spinning_friction", C.c_double),
                ("rolling_friction", C.c_double), ("restitution", C.c_double), ("linear_damping", C.c_double),
                ("motor_kp", C.c_double), ("motor_kd", C.c_double), ("motor_max_force", C.c_double), ("spawn_z", C.c_double),
                ("box_lateral_friction", C.c_double), ("nonfinite_guard", C.c_int32), ("body_contacts", C.c_int32)]


NBODY, MAXMEMB, MAXBOX = 19, 7, 31


class PlenModel(C.Structure):
    """include/plenvec.h PlenModel: the robot as numbers (fixed PLEN topology)."""
    _d = C.c_double
    _fields_ = [("num_bodies", C.c_int32), ("parent", C.c_int32 * NBODY), ("joint_R", (_d * 9) * NBODY), ("joint_t", (_d * 3) * NBODY),
                ("axis", (_d * 3) * NBODY), ("com", (_d * 3) * NBODY), ("inertia", (_d * 6) * NBODY), ("mass", _d * NBODY),
                ("n_member", C.c_int32 * NBODY), ("member_com", ((_d * 3) * MAXMEMB) * NBODY), ("member_mass", (_d * MAXMEMB) * NBODY),
                ("margin", _d), ("foot_break", _d * 2), ("sole", ((_d * 3) * 32) * 2), ("sole_rep", (C.c_int32 * 32) * 2),
                ("sole_order", ((C.c_int32 * 32) * 4) * 2), ("num_boxes", C.c_int32), ("box_body", C.c_int32 * MAXBOX),
                ("box_R", (_d * 9) * MAXBOX), ("box_t", (_d * 3) * MAXBOX), ("box_half", (_d * 3) * MAXBOX), ("box_break", _d * MAXBOX),
                ("box_link_restitution", _d * MAXBOX)]


EXPORTS = ["plenvec_default_model", "plenvec_create_from_model", "plenvec_step2", "plenvec_default_cfg", "plenvec_create", "plenvec_destroy", "plenvec_num_envs", "plenvec_dtype", "plenvec_reset",
           "plenvec_step", "plenvec_get_state", "plenvec_set_state", "plenvec_get_aux", "plenvec_debug_substeps",
           "plenvec_set_params", "plenvec_get_nonfinite_count", "plenvec_timing_begin", "plenvec_timing_end", "plenvec_last_error", "plenvec_version"]

_lib = None


class PlenvecError(RuntimeError):
    pass


def load():
    """dlopen the HIP library; raises if it has not been built (never falls back to anything)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PlenvecError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                           "There is no CPU fallback for the environment step." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32 = C.c_void_p, C.c_int
    lib.plenvec_default_cfg.argtypes = [C.POINTER(PlenCfg), i32]
    lib.plenvec_create.argtypes = [C.POINTER(PlenCfg), i32, i32, C.POINTER(vp)]
    lib.plenvec_default_model.argtypes = [C.POINTER(PlenModel)]
    lib.plenvec_create_from_model.argtypes = [C.POINTER(PlenModel), C.POINTER(PlenCfg), i32, i32, C.POINTER(vp)]
    lib.plenvec_step2.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    lib.plenvec_destroy.argtypes = [vp]
    lib.plenvec_num_envs.argtypes = [vp]
    lib.plenvec_dtype.argtypes = [vp]
    lib.plenvec_reset.argtypes = [vp, vp, vp, vp]
    lib.plenvec_step.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.plenvec_get_state.argtypes = [vp, vp, vp]
    lib.plenvec_set_state.argtypes = [vp, vp, vp]
    lib.plenvec_get_aux.argtypes = [vp, vp, vp]
    lib.plenvec_debug_substeps.argtypes = [vp, vp, i32, vp, vp]
    lib.plenvec_set_params.argtypes = [vp, vp, vp, vp]
    lib.plenvec_get_nonfinite_count.argtypes = [vp, C.POINTER(C.c_int64), vp]
    lib.plenvec_timing_begin.argtypes = [vp, vp]
    lib.plenvec_timing_end.argtypes = [vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.plenvec_last_error.restype = C.c_char_p
    lib.plenvec_version.restype = C.c_char_p
    for name in EXPORTS:
        getattr(lib, name)          # every symbol include/plenvec.h declares must be exported
    _lib = lib
    return lib


def check(code):
    if code != 0:
        raise PlenvecError("libplenvec error %d: %s" % (code, load().plenvec_last_error().decode()))


def default_model():
    m = PlenModel()
    check(load().plenvec_default_model(C.byref(m)))
    return m


def default_cfg(joint_act=False):
    cfg = PlenCfg()
    check(load().plenvec_default_cfg(C.byref(cfg), int(bool(joint_act))))
    return cfg
