/* oracle.h — CPU (float64) restatement of the SMPLSim env-step hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under smplsim_amd/ may include, link or load
 * this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY STATUS: "parity unpinned" for the physics (mj_step): MuJoCo is a third-party
 * dependency of the reference (pyproject.toml:21 `mujoco>=3`), its source is not in
 * /root/reference and no wheel exists in the build container, and the reference holds no
 * golden vectors for it.  The physics below restates MuJoCo's documented
 * pipeline; the pure-NumPy parts of the path (Stable-PD solve, observations, rewards,
 * quaternion helpers) ARE pinned against the reference's own code via tests/golden/.
 *
 * MJ-(V): what a `mujoco` wheel verifies, in the order to check it (each item rests on the ones before it).  One command writes
 * the vectors, one pytest invocation runs every item (README.md "Pinning the physics"):
 *     python tools/dump_mujoco_golden.py tests/golden/mujoco_vectors.npz && python -m pytest tests/test_oracle_vs_mujoco.py
 *   (V1)  model constants: body mass / inertia / ipos / iquat from the geoms, jnt_range          test_stage_model_constants
 *   (V2)  body_invweight0, dof_invweight0 (enter every constraint row's R)                        test_stage_model_constants
 *   (V3)  stat.meaninertia (scale of the solver's termination test)                               test_stage_stat_meaninertia
 *   (V4)  kinematics: xpos, xquat, xipos                                                          test_stage_kinematics
 *   (V5)  qM, qfrc_bias, qacc_smooth                                                              test_stage_inertia_and_bias
 *   (V6)  floor contacts: plane-box "first 4 corners with ldist <= 0", plane-capsule spheres;
 *         which body pairs collide (contype / conaffinity / parent filter / excludes)             test_stage_collision
 *   (V7)  pair functions capsule-capsule, capsule-box, box-box on random geometry — capsule-box's
 *         second contact and box-box's face manifold are restated as RULES (oracle.c "pair
 *         functions"): this is the item most likely to need a change                              test_pair_functions_against_mujoco
 *   (V8)  constraint rows: impedance, aref, R = diagApprox (1 + mu^2), Rpy = 2 mu^2 R, limit rows test_stage_constraint_rows
 *   (V9)  the solve: qacc, efc_force, qfrc_constraint (Newton, mj_solPrimal's termination)        test_stage_solve
 *   (V9b) the solve's line search: solver_niter of the V9 states and of the benchmark rollout with OM_LS_MUJOCO
 *         (PrimalSearch restated: bracketing + 1-D Newton, ls_tolerance 0.01, ls_iterations 50) against the
 *         exact search the kernel and the default oracle use                                      test_stage_solve_iterations
 *   (V10) one mj_step; one control step of the reference loop (15 x Stable PD + mj_step)          test_stage_step_and_control_step
 *   (V11) the benchmark workload's statistics: bad-state autoreset rate, Newton iterations per
 *         control step (does MuJoCo diverge as often as the oracle and the kernel do?)            test_rollout_statistics_of_the_benchmark_workload
 */
#ifndef SMPLSIM_ORACLE_H
#define SMPLSIM_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OM_MAXB 64
#define OM_MAXV (6 + 3 * (OM_MAXB - 1))
#define OM_GEOM_BOX 0
#define OM_GEOM_CAPSULE 1
#define OM_GEOM_SPHERE 2

/* Primitive (uncompiled) model description: what the MJCF text says. */
typedef struct {
  int nbody;
  const int32_t *parent;      /* [nbody], -1 root */
  const double *body_pos;     /* [nbody,3] */
  const int32_t *geom_type;   /* [nbody] */
  const double *geom_params;  /* [nbody,10]: box: pos3 half3 quat4 ; capsule: from3 to3 radius 0 0 0 ; sphere: pos3 0 0 0 radius 0 0 0 */
  const double *density;      /* [nbody] */
  const double *armature;     /* [nv] */
  const double *range_deg;    /* [nv,2] */
  const int32_t *limited;     /* [nv] */
  int nu;
  const int32_t *act_dof;     /* [nu] */
  const double *kp, *kd, *torque_lim, *act_scale, *act_offset; /* [nu] */
  const int32_t *legal_contact; /* [nbody] bodies allowed to touch the floor (contact_bodies) */
  double timestep, gravity, solref[2], solimp[5], margin, mu, impratio;
  /* body-body contacts (reference smpl_humanoid.xml:5,24,231-242): geom contype / conaffinity bit masks, <contact><exclude>
   * body pairs; parent-child pairs are filtered like MuJoCo's filterparent.  self_collision 0 = floor contacts only. */
  int self_collision;
  const int32_t *contype, *conaffinity; /* [nbody] or NULL (= 1, 1) */
  int nexclude;
  const int32_t *exclude;     /* [nexclude,2] body indices */
  int max_self_contacts;      /* 0 = all; otherwise the deepest N body-body contacts are kept (the HIP kernel's capacity) */
} om_desc;

typedef struct om_model om_model;
typedef struct om_data om_data;
typedef struct om_env om_env;

om_model *om_model_create(const om_desc *d);
void om_model_destroy(om_model *m);
/* compiled constants, for checking the product's compiler: field ids below */
enum { OM_M_MASS = 0, OM_M_IPOS, OM_M_IQUAT, OM_M_INERTIA, OM_M_GPOS, OM_M_GQUAT, OM_M_GSIZE,
       OM_M_BODY_INVW, OM_M_DOF_INVW, OM_M_RANGE, OM_M_MEANINERTIA };
int om_model_get(const om_model *m, int field, double *out);
/* Termination of the constraint solver's Newton iteration.  OM_SOLVER_MUJOCO (the default: what mj_step does, MuJoCo
 * engine_solver.c mj_solPrimal): improvement * scale < tolerance || gradient * scale < tolerance, scale = 1 / (meaninertia *
 * max(1, nv)), at most `iterations` iterations (MuJoCo's defaults 1e-8 / 100; arguments <= 0 keep the current values).
 * OM_SOLVER_CONVERGED: iterate until the gradient is at the rounding level of the forces (the parity triage's reference). */
enum { OM_SOLVER_CONVERGED = 0, OM_SOLVER_MUJOCO = 1 };
void om_model_set_solver(om_model *m, int mode, double tolerance, int iterations);
/* The Newton iteration's 1-D search.  OM_LS_EXACT (the default, and what the HIP kernel does): the minimiser of the convex piecewise-
 * quadratic cost along the Newton direction, to rounding.  OM_LS_MUJOCO: mj_solPrimal's PrimalSearch — bracketing + 1-D Newton steps,
 * stopped at |slope| < tolerance * ls_tolerance * |direction| / scale or after ls_iterations evaluations (MuJoCo's defaults 0.01 / 50;
 * arguments <= 0 keep the current values).  The converged solution is the same; iterates, iteration counts and the point at which
 * `improvement < tolerance` fires can differ: MJ-(V9b). */
enum { OM_LS_EXACT = 0, OM_LS_MUJOCO = 1 };
void om_model_set_linesearch(om_model *m, int mode, double ls_tolerance, int ls_iterations);

om_data *om_data_create(const om_model *m);
void om_data_destroy(om_data *d);
enum { OM_D_QPOS = 0, OM_D_QVEL, OM_D_QACC, OM_D_WARM, OM_D_CTRL, OM_D_M, OM_D_BIAS, OM_D_XPOS, OM_D_XQUAT,
       OM_D_LINVEL, OM_D_ANGVEL, OM_D_TOUCH, OM_D_NCON, OM_D_CON_POS, OM_D_CON_DIST, OM_D_CON_BODY,
       OM_D_QACC_SMOOTH, OM_D_NEFC, OM_D_EFC_FORCE, OM_D_SOLVER_ITER, OM_D_ENERGY, OM_D_XIPOS,
       OM_D_QFRC_CONSTRAINT, OM_D_CON_FRAME, OM_D_CON_BODY1 /* first body of every contact, -1 = floor */,
       OM_D_NSELF /* [contacts between two bodies, candidate pairs of the model, contacts dropped by max_self_contacts] */,
       OM_D_QPOS_FWD, OM_D_QVEL_FWD /* the state of the last om_forward: the M and bias the Stable-PD controller reads belong to it */,
       OM_D_LS_STATS /* [line-search evaluations, line searches, searches that ran out of ls_iterations] since creation (OM_LS_MUJOCO) */ };
int om_get(const om_model *m, const om_data *d, int field, double *out);
int om_set(const om_model *m, om_data *d, int field, const double *in);

void om_kinematics(const om_model *m, om_data *d);            /* mj_kinematics */
void om_forward(const om_model *m, om_data *d);               /* mj_forward    */
void om_step(const om_model *m, om_data *d);                  /* mj_step       */
/* StablePDController.control (reference controllers.py:116-190) on the data's (stale) M, bias */
void om_spd_torque(const om_model *m, const om_data *d, const double *action, double *tau);
/* other controllers selectable by control_mode: 1 = pd (controllers.py:335-346), 2 = torque (:45-46),
 * 3 = simple_pid (:193-262; stateful: integral / last error live in the data), 4 = default (ctrl = action) */
void om_ctrl_torque(const om_model *m, const om_data *d, int control_mode, double power_scale,
                    const double *action, double *tau);
void om_set_pid_dt(om_data *d, double dt);                  /* dt handed to SimplePID (timestep * control_freq_inv) */

/* ---- env layer (reference humanoid_env.py / humanoid_task.py / tasks) ---- */
enum { OM_TASK_BASE = 0, OM_TASK_SPEED = 1, OM_TASK_GETUP = 2, OM_TASK_REACH = 3 };
enum { OM_INIT_DEFAULT = 0, OM_INIT_FALL = 1, OM_INIT_EXTERNAL = 2 /* keep the caller's qpos/qvel (reference-state init) */ };
typedef struct {
  int task, state_init, self_obs_v, control_mode /*0 uhc_pd,1 pd,2 torque,3 simple_pid,4 default*/;
  int episode_length, control_freq_inv, root_height_obs;
  double power_scale;
  double tar_speed_min, tar_speed_max; int speed_change_min, speed_change_max;
  double tar_height_min, tar_height_max; int height_change_min, height_change_max, recovery_steps;
  double tar_dist_max; int reach_body;     /* reach task (tasks/humanoid_reach.py); target change steps reuse height_change_* */
} om_env_cfg;

om_env *om_env_create(const om_model *m, const om_env_cfg *cfg);
void om_env_destroy(om_env *e);
int om_env_obs_size(const om_env *e);
om_data *om_env_data(om_env *e);
/* fall_actions: [3,nu] uniform(0,1) draws (used when state_init == Fall), task_rand: [4] uniform(0,1)
 * (speed/getup: [0] target, [1] change steps; reach: [0..2] target xyz, [3] change steps) */
void om_env_reset(om_env *e, const double *fall_actions, const double *task_rand, float *obs);
void om_env_step(om_env *e, const double *action, const double *task_rand, float *obs, double *reward,
                 int *terminated, int *truncated);
void om_env_obs(om_env *e, float *obs);   /* compute_observations() on the current state */
void om_quat_op(int op, const double *a, const double *b, double *out);
int om_narrow_phase(int kind, const double *in, double *out);   /* pair functions on raw geometry (test hook, see oracle.c) */
/* task scalars: [cur_t, tar_speed|tar_height|tar_x, change_steps, recovery_counter, prev_root_pos xyz, tar_y, tar_z] */
void om_env_get_task(const om_env *e, double *out9);
void om_env_set_task(om_env *e, const double *in9);

/* obs functions alone (pinned against the reference's numpy code) */
void om_obs_v1(int nbody, const double *qpos, const double *qvel, const double *xpos, const double *xquat,
               int root_height_obs, float *obs);
void om_obs_v2(int nbody, const double *xpos, const double *xquat, const double *linvel, const double *angvel,
               int root_height_obs, float *obs);

/* batched rollout on `nthreads` host threads, for the cpu_baseline timing: every env gets the
 * same action stream layout actions[step][env][nu]; returns total env-steps done */
long om_batch_rollout(om_env **envs, int nenv, int nsteps, const double *actions, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
