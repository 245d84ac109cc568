/* smplsim_hip.h — C ABI of libsmplsim_hip.so, the MI355X (gfx950) batched SMPL-humanoid env stepper.
 *
 * The reference (ZhengyiLuo/SMPLSim) has no FFI of its own on this path: `HumanoidEnv.step()`
 * calls the MuJoCo C API through its Python bindings and SciPy/LAPACK
 * (reference smpl_sim/envs/humanoid_env.py:439-469, smpl_sim/envs/controllers.py:116-190).
 * Each entry point below names the reference call(s) it replaces.  Conventions follow
 * plain pointers and sizes, no torch types; every function returns 0 on
 * success or a negative ss_status and records a message retrievable with ss_last_error();
 * all device work is enqueued on the caller's hipStream_t (passed as void*) and is
 * asynchronous w.r.t. the host; the caller owns every state/IO buffer (device pointers
 * taken from PyTorch-ROCm tensors), the library owns only the model tables.
 *
 * State layout (struct-of-arrays by field, env-major rows, float32): one wavefront steps one
 * env, so each row `field[env, :]` is a contiguous segment read/written with coalesced loads.
 */
#ifndef SMPLSIM_HIP_H
#define SMPLSIM_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  SS_OK = 0,
  SS_ERR_INVALID = -1,      /* bad argument / unsupported model */
  SS_ERR_HIP = -2,          /* a HIP runtime call failed */
  SS_ERR_NOMEM = -3,
  SS_ERR_LDS = -4           /* model does not fit the 160 KiB LDS budget */
} ss_status;

enum { SS_GEOM_BOX = 0, SS_GEOM_CAPSULE = 1,
       SS_GEOM_SPHERE = 2 };   /* geom_size = (radius, 0, 0); the reference's Skeleton writes spheres with box_body False / freeze_hand (skeleton_local.py:668-677) */
enum { SS_TASK_BASE = 0, SS_TASK_SPEED = 1, SS_TASK_GETUP = 2, SS_TASK_REACH = 3 };   /* reference tasks/humanoid_{speed,getup,reach}.py */
enum { SS_INIT_DEFAULT = 0, SS_INIT_FALL = 1,                          /* HumanoidEnv.StateInit, humanoid_env.py:141-146 */
       SS_INIT_EXTERNAL = 2 };  /* reference-state init (imitation): ss_reset keeps the qpos/qvel the caller wrote into ss_state
                                   (e.g. ss_motion_state_at's qpos/qvel) and runs the reset's mj_forward + observation on them */
enum { SS_CTRL_UHC_PD = 0, SS_CTRL_PD = 1, SS_CTRL_TORQUE = 2, SS_CTRL_SIMPLE_PID = 3, SS_CTRL_DEFAULT = 4 };   /* control_mode, humanoid_env.py:312-323 */

/* Compiled model constants (host pointers, float64; produced by smplsim_amd.mjcf.compile_mjcf).
 * Replaces the mjModel built by mujoco.MjModel.from_xml_string (reference base_env.py:139-142)
 * plus the per-actuator tables of HumanoidEnv.build_pd_action_scale (humanoid_env.py:325-370). */
typedef struct {
  int32_t nbody;                 /* bodies without the world body; nv = 6 + 3*(nbody-1), nq = nv+1 */
  const int32_t *body_parent;    /* [nbody] (-1 root), parents precede children */
  const double *body_pos;        /* [nbody,3] */
  const double *body_mass;       /* [nbody] */
  const double *body_ipos;       /* [nbody,3] */
  const double *body_iquat;      /* [nbody,4] wxyz */
  const double *body_inertia;    /* [nbody,3] principal */
  const int32_t *geom_type;      /* [nbody] SS_GEOM_* */
  const double *geom_size;       /* [nbody,3] box: half sizes; capsule: radius, half length, 0; sphere: radius, 0, 0 */
  const double *geom_pos;        /* [nbody,3] */
  const double *geom_quat;       /* [nbody,4] */
  const double *dof_armature;    /* [nv] */
  const double *jnt_range;       /* [nv,2] radians */
  const uint8_t *jnt_limited;    /* [nv] */
  const double *body_invweight0; /* [nbody,2] */
  const double *dof_invweight0;  /* [nv] */
  const double *qpos0;           /* [nq] */
  int32_t nu;
  const int32_t *actuator_dof;   /* [nu] */
  const double *kp, *kd, *torque_lim, *act_scale, *act_offset;   /* [nu] */
  const uint8_t *legal_contact;  /* [nbody] bodies allowed to touch the floor (cfg.env.contact_bodies) */
  double timestep, gravity, solref[2], solimp[5], margin, friction, impratio;
  /* body-body contacts (reference smpl_humanoid.xml:5,24 contype / conaffinity, :231-242 <exclude> pairs; parent-child pairs
   * are filtered like MuJoCo's filterparent): the candidate pair table is built from these; ss_env_cfg.self_collision turns
   * the pair functions on.  contype / conaffinity may be NULL (= 1) */
  const int32_t *geom_contype;   /* [nbody] */
  const int32_t *geom_conaffinity; /* [nbody] */
  int32_t nexclude;
  const int32_t *exclude;        /* [nexclude,2] body indices */
  /* mjModel.stat.meaninertia: mean diagonal of the joint-space inertia matrix (armature included) at qpos0.  It scales the
   * solver's termination test like in mj_step (below: ss_env_cfg.solver_tolerance).  <= 0: the library computes it from this
   * description.  (Only callers compiled against THIS header may leave it at 0: a caller built against the shorter ABI-2 struct
   * hands over an object that ends before this field, and reading it is undefined behaviour, not zero — INTEGRATION.md "ABI history".) */
  double meaninertia;
} ss_model_desc;

/* Environment configuration (the keys of the reference's smpl_sim/data/cfg/env yaml files). */
typedef struct {
  int32_t task, state_init, self_obs_v, control_mode;
  int32_t episode_length, control_freq_inv, root_height_obs;
  float power_scale;
  float tar_speed_min, tar_speed_max; int32_t speed_change_min, speed_change_max;
  float tar_height_min, tar_height_max; int32_t height_change_min, height_change_max, recovery_steps;
  int32_t newton_iters;          /* mjOption.iterations: max Newton iterations of the constraint solve per mj_step; 0 = MuJoCo's
                                    default, 100 (the reference MJCF does not override it) */
  /* reach task (tasks/humanoid_reach.py): target x,y in +-tar_dist_max, z in [tar_height_min, tar_height_max], resampled
   * every [height_change_min, height_change_max) steps; reward on the world position of body `reach_body` */
  float tar_dist_max; int32_t reach_body;
  /* 1: contacts between the humanoid's own bodies (capsule-capsule, capsule-box, box-box) as mj_step makes
   * them for the reference MJCF — every contact of the narrow phase is kept, like MuJoCo (one per lane of the env's wavefront:
   * SS_MAX_SELF_CONTACTS = 64; rounds 2-3 kept the deepest 8); 0: floor contacts and joint limits only */
  int32_t self_collision;
  /* mjOption.tolerance: the Newton iteration of an mj_step ends like MuJoCo's (engine_solver.c, mj_solPrimal) when
   *   improvement * scale < tolerance   or   gradient * scale < tolerance,   scale = 1 / (meaninertia * max(1, nv)),
   * improvement = the cost decrease of the iteration, gradient = the norm of the cost gradient after it — or, float32 only, when
   * the Newton decrement is below the rounding error of its own evaluation (DESIGN.md "solver termination").  0 = MuJoCo's default
   * 1e-8 (the reference MJCF does not override it). */
  float solver_tolerance;
} ss_env_cfg;
#ifndef SS_MAX_SELF_CONTACTS_N
#define SS_MAX_SELF_CONTACTS_N 64         /* one body-body contact per lane of the wavefront */
#endif
enum { SS_MAX_SELF_CONTACTS = SS_MAX_SELF_CONTACTS_N };

/* Device buffers of one shard of environments (all caller-owned, float32 unless noted). */
typedef struct {
  int32_t num_envs;
  float *qpos;        /* [N,nq]  mjData.qpos */
  float *qvel;        /* [N,nv]  mjData.qvel */
  float *qpos_prev;   /* [N,nq]  state at which the last mj_forward ran (source of the stale qM, qfrc_bias */
  float *qvel_prev;   /* [N,nv]   that StablePDController reads; SURVEY.md §3.2 "staleness subtlety") */
  float *qacc_warm;   /* [N,nv]  mjData.qacc_warmstart */
  float *body_vel;    /* [N,nbody,6] world lin(3)+ang(3) velocity of every body frame at the last forward
                                   (the framelinvel/frameangvel sensors, humanoid_env.py:539-544) */
  int32_t *touch;     /* [N,2]   bit b of the 64-bit mask set: body b touched the floor at the last forward (mjData.contact) */
  int32_t *cur_t;     /* [N]     BaseEnv.cur_t */
  float *task;        /* [N,4]   speed/getup: tar_speed|tar_height, change_steps, recovery_counter, 0 ; reach: tar xyz, change_steps */
  int32_t *nwarn;     /* [N]     count of MuJoCo-style autoresets (mj_checkPos/Vel/Acc) */
  int32_t *solver_iters; /* [N]  Newton iterations spent in the last step (diagnostic) */
  /* control_mode simple_pid only (may be NULL otherwise): the state of the reference's SimplePID object
   * (controllers.py:217-221), which lives as long as the env and is NOT cleared by reset() */
  float *pid_integral;   /* [N,nu] */
  float *pid_last_error; /* [N,nu] */
  int32_t *pid_started;  /* [N]    0 until the controller ran once (its first derivative term is zero) */
  /* models made by ss_model_create_shapes only (NULL otherwise): body shape of every env, 0 .. num_shapes-1; read by every
   * launch, so it may be rewritten between launches (e.g. a new body at reset) */
  const int32_t *shape_id; /* [N] */
  /* optional (may be NULL): body-body contacts of every env at its last forward pass [N] (self_collision envs) */
  int32_t *self_contacts;
} ss_state;

typedef struct ss_model ss_model;
typedef struct ss_batch ss_batch;

/* mujoco.MjModel.from_xml_string + setup_humanoid_properties + setup_controller */
int ss_model_create(const ss_model_desc *desc, int device_id, ss_model **out);
/* Per-env body shapes (reference cfg.robot.has_shape_variation, humanoid_env.py:205: every env process builds its own MJCF from
 * its betas): num_shapes descriptions of the SAME humanoid — tree, joints, limits, gains, actuators, geom types and options
 * must agree; body offsets, inertias, geom sizes / positions and the inverse weights may differ.  The geometry tables are
 * stored per shape and every env reads those of ss_state.shape_id[env]; one launch steps all shapes together. */
int ss_model_create_shapes(const ss_model_desc *descs, int32_t num_shapes, int device_id, ss_model **out);

/* The same from MJCF text, for hosts without the Python compiler: mujoco.MjModel.from_xml_string (reference
 * smpl_sim/envs/base_env.py:139-142) for the MJCF subset the reference's humanoids use (smplsim_amd/csrc/ss_mjcf.h lists it;
 * anything outside is SS_ERR_INVALID with the offending element in the message), plus the per-actuator tables of
 * HumanoidEnv.setup_controller / build_pd_action_scale (humanoid_env.py:312-370) from the reference's gain table by body name
 * (humanoid_env.py:62-84).  opts NULL = the reference defaults below.  len 0 = strlen(xml). */
typedef struct {
  int32_t control_mode;          /* SS_CTRL_*; default SS_CTRL_UHC_PD (cfg.robot.control_mode) */
  int32_t clip_actions;          /* default 1 (cfg.env.clip_actions): action scale = min(1.2 * max|range|, pi) */
  double pdp_scale, pdd_scale;   /* <= 0 = 1 (cfg.env.pdp_scale / pdd_scale) */
  double timestep;               /* <= 0 = 1/450 (cfg.env.sim_timestep_inv) */
  int32_t num_contact_bodies;    /* with contact_bodies: names of the bodies allowed to touch the floor */
  const char *const *contact_bodies; /* NULL = R_Ankle, L_Ankle, R_Toe, L_Toe (cfg.env.contact_bodies) */
} ss_mjcf_options;
int ss_model_create_from_mjcf(const char *xml, size_t len, const ss_mjcf_options *opts, int device_id, ss_model **out);
void ss_model_destroy(ss_model *m);
/* nq, nv, nu, nbody, obs size for (self_obs_v, task, root_height_obs) — humanoid_env.py:293-299 */
int ss_model_dims(const ss_model *m, int32_t *nq, int32_t *nv, int32_t *nu, int32_t *nbody);
/* Diagnostics: the elimination tree of the articulated-body solves ("the elimination tree is rooted at the centre of the
 * body tree"): root = the body the sweeps run towards (SMPL: 10, Spine; SMPL-X: Chest), levels = tree levels below it (6 / 7; rooted
 * at the pelvis the trees are 8 / 10 deep), widest = nodes in its widest level (<= 16), widths[0] = the level of body 0 and
 * widths[1 .. levels] = nodes per level, most_children[1 .. levels] = the most children any node of the level has (room for 33
 * values each), most_children[0] = two bit masks over the levels: bits 0-15, the 1:1 levels (bit L - 1: every node of level L has
 * at most one child, in its own slot of level L + 1: the sweep towards the root keeps such rows in registers); bits 16-30, the levels
 * with a node reached against the kinematic direction (bit 16 + L - 1; its joint's motion subspace enters negated).  Any pointer may be NULL.  The fixed-layout kernel instantiations
 * (ss_env_kernel.h: HdrSmpl, HdrSmplx) carry exactly these numbers as compile-time constants. */
int ss_model_elimination_tree(const ss_model *m, int32_t *root, int32_t *levels, int32_t *widest, int32_t *widths, int32_t *most_children);
int ss_obs_size(const ss_model *m, const ss_env_cfg *cfg);

/* mujoco.MjData(model) for N envs: binds caller-owned state buffers */
int ss_batch_create(const ss_model *m, const ss_env_cfg *cfg, const ss_state *state, ss_batch **out);
void ss_batch_destroy(ss_batch *b);

/* HumanoidEnv.reset()/HumanoidTask.reset() (humanoid_env.py:471-512, humanoid_task.py:6-9) for the envs
 * whose mask byte is non-zero (mask NULL = all).  fall_actions [N,3,nu] uniform(0,1) draws consumed by
 * StateInit.Fall (may be NULL for Default); task_rand [N,4] uniform(0,1) draws for target resampling
 * (speed/getup use [0]=target, [1]=change steps; reach uses [0..2]=target xyz, [3]=change steps)
 * (may be NULL for the base task).  obs [N,obs_size] is written for the reset envs only. */
int ss_reset(ss_batch *b, const uint8_t *mask, const float *fall_actions, const float *task_rand,
             float *obs, void *stream);

/* BaseEnv.step(): control_freq_inv x (controller + mj_step) + observation + reward + reset flags, one
 * launch (humanoid_env.py:439-469; controllers.py:116-190; tasks compute_reward/compute_reset).
 * actions [N,nu]; obs [N,obs_size]; reward [N]; terminated, truncated [N] bytes. */
int ss_step(ss_batch *b, const float *actions, const float *task_rand, float *obs, float *reward,
            uint8_t *terminated, uint8_t *truncated, void *stream);

/* Scheduling hint (no effect on results): a permutation [N] of env ids (device pointer, caller-owned, NULL = natural
 * order).  The persistent wavefronts pull env ids in this order; passing the envs sorted by their previous step's
 * `solver_iters` (descending) starts the expensive ones first (longest-processing-time-first). */
int ss_set_order(ss_batch *b, const int32_t *order);
/* Same hint computed on the device from what the previous step left in ss_state: the envs are handed out by decreasing
 *   key = solver_iters + 6 (bodies touching the floor) + 8 ln(1 + max |qacc_warm|) + 8 ln(1 + max |qvel|),
 * a predictor of the coming step's Newton-iteration count (profiles/r03_lpt_features.txt) — two small launches on `stream`
 * (key per env, counting sort), library-owned buffers; stays in force until ss_set_order(b, NULL/other). */
int ss_schedule_longest_first(ss_batch *b, void *stream);

/* ss_step followed, in the same launch, by the reset of every env whose episode just ended (terminated | truncated) —
 * GymVectEnv's autoreset (reference nv/gymwrapper.py:53-60) without a second launch.  StateInit.Default, or StateInit.Fall after
 * ss_set_fall_actions (the Fall reset's 45 warm-up mj_steps then run in the wave that stepped the env; a separate masked
 * ss_reset launch lasts as long as a whole step however few envs it resets).  obs = observation of the step for every env (the
 * "final_observation" of the envs that ended); obs_next = the observation the policy acts on next: the reset one for
 * envs that ended, the same as obs otherwise.  reset_task_rand [N,4] feeds reset_task of the envs that ended. */
int ss_step_autoreset(ss_batch *b, const float *actions, const float *task_rand, const float *reset_task_rand, float *obs,
                      float *obs_next, float *reward, uint8_t *terminated, uint8_t *truncated, void *stream);
/* fall_actions [N,3,nu] uniform(0,1) draws (device pointer, caller-owned, read by every following ss_step_autoreset of a
 * StateInit.Fall batch: refill it with fresh draws between steps; NULL = none) */
int ss_set_fall_actions(ss_batch *b, const float *fall_actions);

/* n x (controller + mj_step) without the env epilogue — substep-granular parity/debugging */
int ss_substep(ss_batch *b, const float *actions, int n_substeps, void *stream);

/* mj_kinematics readback: xpos [N,nbody,3], xquat-equivalent rotation matrices xmat [N,nbody,9] */
int ss_kinematics(ss_batch *b, float *xpos, float *xmat, void *stream);

/* Optional by-product of ss_step / ss_step_autoreset / ss_reset: the body frames of the launch's last forward (what a
 * following ss_kinematics would return) are also written to xpos [N,nbody,3] / xmat [N,nbody,9] (caller-owned; both NULL =
 * off).  Saves the extra launch for callers that need world body poses every step (the imitation task, smplsim_motion.h). */
int ss_set_body_outputs(ss_batch *b, float *xpos, float *xmat);

/* Optional by-product of ss_step / ss_step_autoreset: HumanoidEnv.curr_power_usage (reference humanoid_env.py:443-451) — per
 * mj_step of the control step, |qfrc_actuator * qvel| of every hinge dof with the torque applied in that mj_step and the velocity
 * after it: power [N, control_freq_inv, nv - 6] (caller-owned; NULL = off, the default).  Like ss_set_body_outputs it selects the
 * kernel instantiation with the optional outputs. */
int ss_set_power_output(ss_batch *b, float *power);

/* Diagnostics for parity triage: one mj_forward at (qpos, qvel) with raw joint torques [N,nu] (NULL = 0);
 * writes the dense joint-space mass matrix [N,nv,nv] (what mj_fullM returns; the stepping path itself never forms
 * it), qfrc_bias [N,nv] and the constrained qacc [N,nv]. */
int ss_debug_forward(ss_batch *b, const float *torques, float *M, float *bias, float *qacc, void *stream);
/* Diagnostics (self_collision batches): a caller-owned buffer [N, SS_MAX_SELF_CONTACTS, 24] that every launch fills with the
 * body-body contact records of each env's last forward pass (mjData.contact of the body pairs): body1 body2 | world position 3 |
 * normal body1->body2 3 | first tangent 3 | 1/R of the pyramid rows | aref 4 | jar 4 | jd 4 ; NULL turns it off. */
int ss_debug_self_contacts(ss_batch *b, float *records);
/* Diagnostics (self_collision batches): counts [N] (int32, caller-owned, NULL = off) — every mj_step of env n whose narrow phase
 * found more than SS_MAX_SELF_CONTACTS = 64 body-body contacts (one per lane of the wavefront; the 65th and later ones are dropped;
 * MuJoCo keeps all: the reference MJCF asks for nconmax 700, smpl_humanoid.xml:293) adds 1 to counts[n].  bench.py reports the rate
 * as reference_contact_set.truncated_mj_step_frac (3e-5 on the benchmark workload: states about to blow up). */
int ss_debug_self_truncation(ss_batch *b, int32_t *counts);
/* ---- caller side of the path: device-side generalised advantage estimation for the PPO sampler that
 * feeds ss_step.  The rollout is stored time-major [T,N]; one recursion per env column, exactly the loop of the
 * reference's estimate_advantages (smpl_sim/learning/learning_utils.py:198-218):
 *   delta_t = r_t + gamma * V_{t+1} * not_dead_t - V_t ;  A_t = delta_t + gamma * tau * A_{t+1} * not_done_t ;  R_t = V_t + A_t
 * with V_T = bootstrap[n] (NULL = 0, which is what the reference's episode-complete batches amount to) and A_T = 0.
 * Advantage normalisation (mean / std over the batch) is left to the caller.  Stateless; runs on the current device. */
int ss_gae(const float *rewards, const float *not_done, const float *not_dead, const float *values, const float *bootstrap,
           int32_t T, int32_t N, float gamma, float tau, float *advantages, float *returns, void *stream);

/* -DSS_PROFILE builds only: accumulated shader-clock ticks per kernel stage (tools/stage_profile.py) */
int ss_debug_prof(ss_batch *b, unsigned long long *out, int n);

/* launch geometry: resident envs per workgroup = per CU (LDS-capacity bound) and the LDS bytes of such a workgroup; batches
 * smaller than (CUs x envs_per_wg) are launched with ceil(N / CUs) envs per workgroup so that they cover every CU;
 * kernel_regs = VGPRs of the kernel instantiation THIS batch launched last (0 before its first launch) */
int ss_launch_info(const ss_batch *b, int32_t *envs_per_wg, int32_t *lds_bytes, int32_t *kernel_regs);

/* Launch-geometry override for callers that step several batches concurrently on different streams (body-shape groups):
 * envs_per_wg = run every launch of this batch with exactly that many envs per workgroup (1 .. ss_launch_info's value; 0 =
 * automatic, which spreads a small batch over all CUs and thereby claims every CU's LDS); max_workgroups = cap on the
 * persistent workgroups of a launch (0 = one per CU), so that K concurrent batches can each own 1/K of the CUs. */
int ss_set_launch_geometry(ss_batch *b, int32_t envs_per_wg, int32_t max_workgroups);

/* State access by field for hosts that do not keep the ss_state pointers around (teacher forcing, checkpoints; what callers of
 * the reference read from / write to mjData: data.qpos, data.qvel, data.xpos, data.xmat, the velocity sensors, the floor
 * contacts).  Device-to-device copies on `stream` between the batch's bound buffers and `buf` (a device pointer, float32 unless
 * noted); XPOS / XMAT run mj_kinematics on the current qpos (ss_kinematics) into the batch's own scratch first.
 * ss_set_state accepts QPOS, QVEL, QACC_WARM and CUR_T; QPOS / QVEL also overwrite the "previous" copies the Stable-PD controller
 * takes its stale M, C from (i.e. the state counts as having been forwarded, like after mj_forward). */
enum { SS_FIELD_QPOS = 0,      /* [N,nq] */
       SS_FIELD_QVEL = 1,      /* [N,nv] */
       SS_FIELD_XPOS = 2,      /* [N,nbody,3] */
       SS_FIELD_XMAT = 3,      /* [N,nbody,9] */
       SS_FIELD_BODY_VEL = 4,  /* [N,nbody,6] framelinvel ; frameangvel of the last forward */
       SS_FIELD_TOUCH = 5,     /* [N,2] int32: bit b = body b touches the floor */
       SS_FIELD_QACC_WARM = 6, /* [N,nv] */
       SS_FIELD_CUR_T = 7 };   /* [N] int32 */
int ss_get_state(ss_batch *b, int32_t field, void *buf, void *stream);
int ss_set_state(ss_batch *b, int32_t field, const void *buf, void *stream);

/* Launches of one batch must be enqueued on ONE stream (or be ordered by the caller's events): a launch zeroes the work counter of
 * the next one, so a launch on another stream must not start before the previous launch of the batch has finished. */

/* Message of the calling thread's last failed call; and of the last failed call that took this handle (valid until the next
 * failing call on the handle or its destruction) — for hosts that call from pooled threads and cannot rely on thread identity. */
const char *ss_last_error(void);
const char *ss_model_last_error(const ss_model *m);
const char *ss_batch_last_error(const ss_batch *b);

#ifdef __cplusplus
}
#endif
#endif
