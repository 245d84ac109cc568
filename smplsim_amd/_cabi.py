"""ctypes declarations of the C ABI in include/smplsim_hip.h and include/smplsim_motion.h (structs + prototypes).

Pure declarations: `bind(cdll)` attaches argtypes/restypes to an already-loaded library.
The product loads libsmplsim_hip.so through smplsim_amd/_lib.py; tests bind the same
declarations onto the wavefront-emulator build of the kernel source.
"""
import ctypes as C

import numpy as np

TASK_BASE, TASK_SPEED, TASK_GETUP, TASK_REACH = 0, 1, 2, 3
INIT_DEFAULT, INIT_FALL, INIT_EXTERNAL = 0, 1, 2
CTRL_UHC_PD, CTRL_PD, CTRL_TORQUE, CTRL_SIMPLE_PID, CTRL_DEFAULT = 0, 1, 2, 3, 4
TASKS = {"HumanoidEnv": TASK_BASE, "HumanoidSpeed": TASK_SPEED, "HumanoidGetup": TASK_GETUP, "HumanoidReach": TASK_REACH}
CONTROL_MODES = {"uhc_pd": CTRL_UHC_PD, "pd": CTRL_PD, "torque": CTRL_TORQUE, "simple_pid": CTRL_SIMPLE_PID,
                 "default": CTRL_DEFAULT}
STATE_INITS = {"Default": INIT_DEFAULT, "Fall": INIT_FALL, "External": INIT_EXTERNAL}
SS_MAX_SELF_CONTACTS = 64        # include/smplsim_hip.h: one body-body contact per lane of the env's wavefront


class ModelDesc(C.Structure):
    _fields_ = [
        ("nbody", C.c_int32), ("body_parent", C.c_void_p), ("body_pos", C.c_void_p), ("body_mass", C.c_void_p),
        ("body_ipos", C.c_void_p), ("body_iquat", C.c_void_p), ("body_inertia", C.c_void_p),
        ("geom_type", C.c_void_p), ("geom_size", C.c_void_p), ("geom_pos", C.c_void_p), ("geom_quat", C.c_void_p),
        ("dof_armature", C.c_void_p), ("jnt_range", C.c_void_p), ("jnt_limited", C.c_void_p),
        ("body_invweight0", C.c_void_p), ("dof_invweight0", C.c_void_p), ("qpos0", C.c_void_p),
        ("nu", C.c_int32), ("actuator_dof", C.c_void_p), ("kp", C.c_void_p), ("kd", C.c_void_p),
        ("torque_lim", C.c_void_p), ("act_scale", C.c_void_p), ("act_offset", C.c_void_p),
        ("legal_contact", C.c_void_p),
        ("timestep", C.c_double), ("gravity", C.c_double), ("solref", C.c_double * 2), ("solimp", C.c_double * 5),
        ("margin", C.c_double), ("friction", C.c_double), ("impratio", C.c_double),
        ("geom_contype", C.c_void_p), ("geom_conaffinity", C.c_void_p), ("nexclude", C.c_int32), ("exclude", C.c_void_p),
        ("meaninertia", C.c_double),
    ]


class EnvCfg(C.Structure):
    _fields_ = [
        ("task", C.c_int32), ("state_init", C.c_int32), ("self_obs_v", C.c_int32), ("control_mode", C.c_int32),
        ("episode_length", C.c_int32), ("control_freq_inv", C.c_int32), ("root_height_obs", C.c_int32),
        ("power_scale", C.c_float),
        ("tar_speed_min", C.c_float), ("tar_speed_max", C.c_float), ("speed_change_min", C.c_int32),
        ("speed_change_max", C.c_int32),
        ("tar_height_min", C.c_float), ("tar_height_max", C.c_float), ("height_change_min", C.c_int32),
        ("height_change_max", C.c_int32), ("recovery_steps", C.c_int32), ("newton_iters", C.c_int32),
        ("tar_dist_max", C.c_float), ("reach_body", C.c_int32), ("self_collision", C.c_int32),
        ("solver_tolerance", C.c_float),
    ]


class State(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int32), ("qpos", C.c_void_p), ("qvel", C.c_void_p), ("qpos_prev", C.c_void_p),
        ("qvel_prev", C.c_void_p), ("qacc_warm", C.c_void_p), ("body_vel", C.c_void_p), ("touch", C.c_void_p),
        ("cur_t", C.c_void_p), ("task", C.c_void_p), ("nwarn", C.c_void_p), ("solver_iters", C.c_void_p),
        ("pid_integral", C.c_void_p), ("pid_last_error", C.c_void_p), ("pid_started", C.c_void_p), ("shape_id", C.c_void_p),
        ("self_contacts", C.c_void_p),
    ]


class Skeleton(C.Structure):
    _fields_ = [("nbody", C.c_int32), ("parent", C.c_void_p), ("smpl_2_mujoco", C.c_void_p)]


MOTION_DATA_ARRAYS = ("length_starts", "motion_num_frames", "motion_dt", "motion_lengths", "frame_motion", "pose_aa", "trans",
                      "offsets", "gts", "grs", "lrs", "gvs", "gavs", "dof_pos", "dvs", "qpos", "qvel")


class MotionData(C.Structure):
    _fields_ = [("num_motions", C.c_int32), ("num_frames", C.c_int32), ("nbody", C.c_int32)] + \
               [(n, C.c_void_p) for n in MOTION_DATA_ARRAYS]


MOTION_STATE_FIELDS = ("root_pos", "root_rot", "dof_pos", "root_vel", "root_ang_vel", "dof_vel", "rg_pos", "rb_rot", "body_vel",
                       "body_ang_vel", "qpos", "qvel")


class MotionState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in MOTION_STATE_FIELDS]


class ImitationCfg(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("k_pos", "k_rot", "k_vel", "k_ang_vel", "w_pos", "w_rot", "w_vel", "w_ang_vel",
                                         "termination_distance", "obs_dt")]


class ImitationIO(C.Structure):
    _fields_ = [("data", C.POINTER(MotionData)), ("cfg", ImitationCfg), ("motion_ids", C.c_void_p), ("start_times", C.c_void_p),
                ("offset", C.c_void_p), ("sampling_cdf", C.c_void_p), ("truncate_time", C.c_float), ("random_start", C.c_int32),
                ("obs_final", C.c_void_p), ("obs_next", C.c_void_p), ("obs_stride", C.c_int32), ("reward", C.c_void_p),
                ("reward_parts", C.c_void_p), ("terminated", C.c_void_p), ("truncated", C.c_void_p)]


class MjcfOptions(C.Structure):
    _fields_ = [("control_mode", C.c_int32), ("clip_actions", C.c_int32), ("pdp_scale", C.c_double), ("pdd_scale", C.c_double),
                ("timestep", C.c_double), ("num_contact_bodies", C.c_int32), ("contact_bodies", C.POINTER(C.c_char_p))]


def bind(lib):
    vp = C.c_void_p
    lib.ss_model_create_from_mjcf.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(MjcfOptions), C.c_int, C.POINTER(vp)]
    lib.ss_set_fall_actions.argtypes = [vp, vp]
    lib.ss_get_state.argtypes = [vp, C.c_int32, vp, vp]
    lib.ss_set_state.argtypes = [vp, C.c_int32, vp, vp]
    lib.ss_imitation_bind.argtypes = [vp, C.POINTER(ImitationIO)]
    lib.ss_imitation_step_fused.argtypes = [vp, vp, vp, vp]
    lib.ss_model_last_error.argtypes = [vp]; lib.ss_model_last_error.restype = C.c_char_p
    lib.ss_batch_last_error.argtypes = [vp]; lib.ss_batch_last_error.restype = C.c_char_p
    lib.ss_model_create.argtypes = [C.POINTER(ModelDesc), C.c_int, C.POINTER(vp)]
    lib.ss_model_create_shapes.argtypes = [C.POINTER(ModelDesc), C.c_int32, C.c_int, C.POINTER(vp)]
    lib.ss_model_destroy.argtypes = [vp]; lib.ss_model_destroy.restype = None
    lib.ss_model_dims.argtypes = [vp] + [C.POINTER(C.c_int32)] * 4
    lib.ss_obs_size.argtypes = [vp, C.POINTER(EnvCfg)]
    lib.ss_batch_create.argtypes = [vp, C.POINTER(EnvCfg), C.POINTER(State), C.POINTER(vp)]
    lib.ss_batch_destroy.argtypes = [vp]; lib.ss_batch_destroy.restype = None
    lib.ss_reset.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.ss_step.argtypes = [vp] * 8
    lib.ss_substep.argtypes = [vp, vp, C.c_int, vp]
    lib.ss_kinematics.argtypes = [vp, vp, vp, vp]
    lib.ss_debug_forward.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.ss_debug_self_contacts.argtypes = [vp, vp]
    lib.ss_debug_self_truncation.argtypes = [vp, vp]
    lib.ss_model_elimination_tree.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.ss_step_autoreset.argtypes = [vp] * 10
    lib.ss_schedule_longest_first.argtypes = [vp, vp]
    lib.ss_gae.argtypes = [vp, vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_float, C.c_float, vp, vp, vp]
    lib.ss_debug_prof.argtypes = [vp, vp, C.c_int]
    lib.ss_set_order.argtypes = [vp, vp]
    lib.ss_set_body_outputs.argtypes = [vp, vp, vp]
    lib.ss_set_power_output.argtypes = [vp, vp]
    lib.ss_set_launch_geometry.argtypes = [vp, C.c_int32, C.c_int32]
    lib.ss_launch_info.argtypes = [vp] + [C.POINTER(C.c_int32)] * 3
    lib.ss_last_error.argtypes = []; lib.ss_last_error.restype = C.c_char_p
    lib.ss_motion_cook.argtypes = [C.POINTER(Skeleton), C.POINTER(MotionData), C.c_int32, vp]
    lib.ss_motion_state_at.argtypes = [C.POINTER(MotionData), vp, vp, vp, vp, C.c_int32, C.c_int32, C.POINTER(MotionState), vp]
    lib.ss_motion_resample.argtypes = [C.POINTER(MotionData), vp, vp, vp, C.c_float, C.c_int32, vp, vp, vp]
    lib.ss_imitation_step.argtypes = [C.POINTER(MotionData), C.POINTER(ImitationCfg)] + [vp] * 5 + [C.c_int32] + [vp] * 4 + \
                                     [C.c_int32] + [vp] * 5
    return lib


FIELDS = {"qpos": 0, "qvel": 1, "xpos": 2, "xmat": 3, "body_vel": 4, "touch": 5, "qacc_warm": 6, "cur_t": 7}   # SS_FIELD_*
ACTIVATIONS = {"none": 0, "silu": 1, "tanh": 2, "relu": 3}
MLP_EXPORTS = ["ss_linear_bf16", "ss_linear_bf16_train", "ss_linear_bf16_dx", "ss_wgrad_bf16", "ss_obs_to_bf16", "ss_gaussian_sample"]            # include/smplsim_mlp.h (product library only: the matrix-core kernels)


def bind_mlp(lib):
    vp = C.c_void_p
    lib.ss_linear_bf16.argtypes = [vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp]
    lib.ss_linear_bf16_train.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp]
    lib.ss_linear_bf16_dx.argtypes = [vp, vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp]
    lib.ss_wgrad_bf16.argtypes = [vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp]
    lib.ss_obs_to_bf16.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, C.c_float, C.c_float, C.c_float, vp, C.c_int32, vp]
    lib.ss_gaussian_sample.argtypes = [vp, vp, vp, C.c_int32, C.c_int32, vp, C.c_int32, vp, C.c_int32, C.c_float, C.c_float, vp, vp]
    return lib


EXPORTS = ["ss_model_create", "ss_model_create_shapes", "ss_model_destroy", "ss_model_dims", "ss_model_elimination_tree", "ss_obs_size", "ss_batch_create",
           "ss_batch_destroy", "ss_reset", "ss_step", "ss_step_autoreset", "ss_substep", "ss_kinematics", "ss_debug_forward", "ss_debug_self_contacts", "ss_debug_self_truncation",
           "ss_gae", "ss_debug_prof", "ss_set_order", "ss_set_body_outputs", "ss_set_power_output", "ss_set_launch_geometry", "ss_schedule_longest_first", "ss_launch_info", "ss_last_error",
           "ss_model_create_from_mjcf", "ss_model_last_error", "ss_batch_last_error", "ss_imitation_bind", "ss_imitation_step_fused", "ss_get_state", "ss_set_state", "ss_set_fall_actions",
           "ss_motion_cook", "ss_motion_state_at", "ss_motion_resample", "ss_imitation_step"]


def make_model_desc(mc, kp, kd, torque_lim, act_scale, act_offset, legal_bodies=(), timestep=1.0 / 450):
    """ModelConst (smplsim_amd.mjcf) + actuator tables -> (ModelDesc, keepalive list)."""
    keep = []

    def arr(x, dt):
        a = np.ascontiguousarray(x, dtype=dt)
        keep.append(a)
        return a.ctypes.data_as(C.c_void_p)

    d = ModelDesc()
    d.nbody = mc.nbody
    d.body_parent = arr(mc.body_parent, np.int32)
    for name in ("body_pos", "body_mass", "body_ipos", "body_iquat", "body_inertia", "geom_size", "geom_pos",
                 "geom_quat", "dof_armature", "jnt_range", "body_invweight0", "dof_invweight0", "qpos0"):
        setattr(d, name, arr(getattr(mc, name), np.float64))
    d.geom_type = arr(mc.geom_type, np.int32)
    d.jnt_limited = arr(mc.jnt_limited, np.uint8)
    d.nu = mc.nu
    d.actuator_dof = arr(mc.actuator_dof, np.int32)
    d.kp, d.kd, d.torque_lim = arr(kp, np.float64), arr(kd, np.float64), arr(torque_lim, np.float64)
    d.act_scale, d.act_offset = arr(act_scale, np.float64), arr(act_offset, np.float64)
    d.legal_contact = arr([int(n in legal_bodies) for n in mc.body_names], np.uint8)
    d.timestep, d.gravity = timestep, mc.gravity
    d.solref = (C.c_double * 2)(*mc.solref)
    d.solimp = (C.c_double * 5)(*mc.solimp)
    d.margin, d.friction, d.impratio = mc.geom_margin, mc.friction, mc.impratio
    d.geom_contype = arr(mc.geom_contype, np.int32)
    d.geom_conaffinity = arr(mc.geom_conaffinity, np.int32)
    ex = [(mc.body_names.index(a), mc.body_names.index(b)) for a, b in mc.excludes]
    d.nexclude = len(ex)
    d.exclude = arr(np.array(ex, dtype=np.int32).reshape(-1, 2), np.int32) if ex else None
    d.meaninertia = mc.meaninertia
    return d, keep


def make_env_cfg(task=TASK_BASE, state_init=INIT_DEFAULT, self_obs_v=1, control_mode=CTRL_UHC_PD,
                 episode_length=300, control_freq_inv=15, root_height_obs=True, power_scale=1.0,
                 tar_speed=(0.0, 5.0), speed_change=(100, 200), tar_height=(0.5, 1.2), height_change=(100, 200),
                 recovery_steps=60, newton_iters=0, tar_dist_max=1.0, reach_body=0, self_collision=False, solver_tolerance=0.0):
    """newton_iters / solver_tolerance: mjOption.iterations / tolerance; 0 = MuJoCo's defaults (100, 1e-8)."""
    return EnvCfg(task, state_init, self_obs_v, control_mode, episode_length, control_freq_inv, int(root_height_obs),
                  power_scale, tar_speed[0], tar_speed[1], speed_change[0], speed_change[1], tar_height[0],
                  tar_height[1], height_change[0], height_change[1], recovery_steps, newton_iters, tar_dist_max, reach_body,
                  int(bool(self_collision)), solver_tolerance)
