/* smplsim_motion.h — C ABI of the motion-library / imitation side of libsmplsim_hip.so (BASELINE config 4).
 *
 * The reference keeps its motion clips as flat, concatenated per-frame arrays (MotionLibBase.load_motions,
 * smpl_sim/smpllib/motion_lib_base.py:176-197) that it fills on the CPU, one clip at a time, with torch code
 * (Humanoid_Batch.fk_batch, smpl_sim/smpllib/torch_smpl_humanoid_batch.py:118-165) and then indexes per env per control
 * step (get_motion_state / get_motion_state_intervaled, motion_lib_base.py:311-423).  Here the same arrays live in HBM,
 * are filled by three launches over ALL frames of ALL clips at once, and are sampled — and turned into the imitation task
 * observation, tracking reward and early-termination flag — by one launch per control step next to ss_step.
 *
 * Conventions as in smplsim_hip.h: plain pointers and sizes, every function returns 0 or a negative ss_status with the
 * message in ss_last_error(); device work is enqueued on the caller's hipStream_t (void*); the caller owns every
 * buffer (device pointers of PyTorch-ROCm tensors) — these functions are stateless.  Quaternions are wxyz.
 * Body order is the MuJoCo (MJCF depth-first) order of the reference's SMPL_MUJOCO_NAMES; the raw clips arrive in SMPL
 * joint order exactly as the reference's AMASS pickles hold them.
 */
#ifndef SMPLSIM_MOTION_H
#define SMPLSIM_MOTION_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Humanoid_Batch._parents / smpl_2_mujoco (torch_smpl_humanoid_batch.py:44,74).  HOST pointers (validated and packed
 * into the launch arguments): nbody <= 64, parents precede children in depth-first order, tree depth <= 16. */
typedef struct {
  int32_t nbody;
  const int32_t *parent;          /* [J] MuJoCo order, -1 = root */
  const int32_t *smpl_2_mujoco;   /* [J] SMPL joint index of MuJoCo body j */
} ss_skeleton;

/* The motion library's arrays, all DEVICE pointers, float32 unless noted; F = total frames of the M loaded clips. */
typedef struct {
  int32_t num_motions, num_frames, nbody;
  /* clip table — MotionLibBase._motion_num_frames/_motion_dt/_motion_lengths/length_starts (motion_lib_base.py:170-190) */
  const int32_t *length_starts;       /* [M] first frame of each clip */
  const int32_t *motion_num_frames;   /* [M] */
  const float *motion_dt;             /* [M] 1/fps */
  const float *motion_lengths;        /* [M] dt * (num_frames - 1) */
  const int32_t *frame_motion;        /* [F] clip of each frame (read by ss_motion_cook only) */
  /* raw clips, read by ss_motion_cook only (may be NULL afterwards) */
  const float *pose_aa;               /* [F,J,3] axis-angle, SMPL joint order (the pickles' pose_aa) */
  const float *trans;                 /* [F,3] root translation (after the caller's height fix) */
  const float *offsets;               /* [M,J,3] joint offsets of each clip's body shape, MuJoCo order (Humanoid_Batch._offsets) */
  /* cooked arrays = the attributes of the same names of MotionLibBase (motion_lib_base.py:176-187) */
  float *gts;       /* [F,J,3] global_translation */
  float *grs;       /* [F,J,4] global_rotation */
  float *lrs;       /* [F,J,4] local_rotation (SMPL joint order, like the reference) */
  float *gvs;       /* [F,J,3] global_velocity          (grvs  = gvs[:,0]) */
  float *gavs;      /* [F,J,3] global_angular_velocity  (gravs = gavs[:,0]) */
  float *dof_pos;   /* [F,J-1,3] Euler XYZ of the local rotations, MuJoCo order */
  float *dvs;       /* [F,J-1,3] dof_vels */
  float *qpos;      /* [F,7+3(J-1)] */
  float *qvel;      /* [F,6+3(J-1)] */
} ss_motion_data;

/* Humanoid_Batch.fk_batch(return_full=True, count_offset=True) for every frame of every clip
 * (torch_smpl_humanoid_batch.py:118-228: axis-angle -> matrices, forward kinematics, matrix_to_quaternion, Euler dofs with
 * fix_continous_dof, finite-difference velocities with the sigma-2 Gaussian filter when filter_vel != 0). */
int ss_motion_cook(const ss_skeleton *skel, const ss_motion_data *data, int32_t filter_vel, void *stream);

/* Outputs of a lookup, device pointers [N,...]; any pointer may be NULL (not written). */
typedef struct {
  float *root_pos;      /* [N,3] */
  float *root_rot;      /* [N,4] */
  float *dof_pos;       /* [N,3(J-1)] */
  float *root_vel;      /* [N,3] */
  float *root_ang_vel;  /* [N,3] */
  float *dof_vel;       /* [N,3(J-1)] */
  float *rg_pos;        /* [N,J,3]  ("xpos" of the intervaled lookup) */
  float *rb_rot;        /* [N,J,4]  ("xquat") */
  float *body_vel;      /* [N,J,3] */
  float *body_ang_vel;  /* [N,J,3] */
  float *qpos;          /* [N,7+3(J-1)] reference-state initialisation: feed to ss_state.qpos */
  float *qvel;          /* [N,6+3(J-1)] */
} ss_motion_state;

/* MotionLibBase.get_motion_state (intervaled = 0: frames f0, f1 blended — lerp, slerp for rotations; motion_lib_base.py:
 * 359-423 with the integer frame numbers of its PHC original) or get_motion_state_intervaled (intervaled = 1: the single
 * frame of :311-355) for N (motion id, time) pairs; offset [N,3] is added to the body positions (NULL = none); rows whose
 * mask byte is 0 are left untouched (mask NULL = all) — with out->qpos / out->qvel pointing at ss_state.qpos / qvel this is
 * the reference-state initialisation of the envs that just finished, in place. */
int ss_motion_state_at(const ss_motion_data *data, const int32_t *motion_ids, const float *times, const float *offset,
                       const uint8_t *mask, int32_t N, int32_t intervaled, const ss_motion_state *out, void *stream);

/* MotionLibBase.sample_motions + sample_time (motion_lib_base.py:277-292) on the device for the envs whose mask byte is set
 * (NULL = all): rand [N,2] uniform(0,1) draws, cdf [M] the inclusive cumulative sum of the batch sampling probabilities
 * (_sampling_batch_prob); motion_ids[n] = the clip whose CDF interval holds rand[n,0], start_times[n] = rand[n,1] *
 * max(length - truncate_time, 0). */
int ss_motion_resample(const ss_motion_data *data, const uint8_t *mask, const float *rand, const float *cdf, float truncate_time,
                       int32_t N, int32_t *motion_ids, float *start_times, void *stream);

/* Imitation task.  NOT in the reference (SURVEY.md 8f-2 asks to define it): PHC's tracking reward
 *   r = w_pos exp(-k_pos mean|dp|^2) + w_rot exp(-k_rot mean angle^2) + w_vel exp(-k_vel mean|dv|^2) + w_ang exp(-k_ang mean|dw|^2)
 * against the clip at time t = start_times[n] + cur_t[n] * obs_dt, early termination when the mean body distance exceeds
 * termination_distance, truncation when t + obs_dt reaches the end of the clip, and PHC's v6 task observation against the
 * clip at t + obs_dt, per body, in the root-heading frame:
 *   [dpos 3J | drot (tan-norm) 6J | dvel 3J | dangvel 3J | ref pos rel. root 3J | ref rot (tan-norm) 6J]   -> 24 J floats. */
typedef struct {
  float k_pos, k_rot, k_vel, k_ang_vel;
  float w_pos, w_rot, w_vel, w_ang_vel;
  float termination_distance;
  float obs_dt;                 /* control step (s): clip time advances by obs_dt per env step; the observation looks obs_dt ahead */
} ss_imitation_cfg;

/* One launch per control step: samples the clips for every env and body and reduces over bodies inside the wavefront.
 * xpos [N,J,3], xmat [N,J,9] (ss_set_body_outputs / ss_kinematics), body_vel [N,J,6] (ss_state.body_vel) describe the
 * simulated humanoid; cur_t [N] = ss_state.cur_t (NULL = 0).  Outputs: task_obs rows of 24J floats at row stride obs_stride
 * floats (so they can land behind the self observation in one policy input buffer); reward [N], reward_parts [N,4],
 * terminated [N], truncated [N] bytes — each may be NULL.  Envs whose mask byte is 0 are skipped (mask NULL = all). */
int ss_imitation_step(const ss_motion_data *data, const ss_imitation_cfg *cfg, const int32_t *motion_ids, const float *start_times,
                      const int32_t *cur_t, const float *offset, const uint8_t *mask, int32_t N, const float *xpos, const float *xmat,
                      const float *body_vel, float *task_obs, int32_t obs_stride, float *reward, float *reward_parts,
                      uint8_t *terminated, uint8_t *truncated, void *stream);

/* The whole imitation control step in ONE launch: ss_step (15 x (Stable-PD + mj_step), self observation, body frames) and, in
 * the wavefront that stepped the env, on what it just wrote: the work of ss_imitation_step (reward, flags, task observation
 * behind the self observation) and — for the envs that terminated or ran out of clip, when `rand` is given — the reference-state
 * re-initialisation PHC's HumanoidIm does: ss_motion_resample (clip by inverse CDF from rand[n,0], start time from rand[n,1]),
 * ss_motion_state_at into the simulator's qpos / qvel, the reset's mj_forward + self observation (ss_reset, StateInit External)
 * and the task observation of the new state.  Same element functions, same results as that sequence of six launches.
 *
 * ss_imitation_bind stores the buffers (device pointers the caller keeps alive and in place) with the batch; the batch must
 * have task base, StateInit External and ss_set_body_outputs buffers (per-env body shapes are fine; with self_collision the
 * one-launch step exists for SMPL-sized single-shape models — otherwise SS_ERR_INVALID, use the separate launches).  Rows of obs_final / obs_next are
 * [self observation | task observation (24 J)] at a row stride of obs_stride floats: obs_final = after the step (what the
 * learner stores), obs_next = what the policy acts on next (= obs_final for envs that go on). */
struct ss_batch;
typedef struct {
  const ss_motion_data *data;    /* copied */
  ss_imitation_cfg cfg;
  int32_t *motion_ids;           /* [N] clip of every env, rewritten on re-initialisation */
  float *start_times;            /* [N] clip time at cur_t = 0 */
  const float *offset;           /* [N,3] world offset of the clips, or NULL */
  const float *sampling_cdf;     /* [num_motions] inclusive CDF of the clip sampling probabilities (NULL: no re-initialisation) */
  float truncate_time;           /* sample_time(truncate_time): start times are drawn from [0, length - truncate_time) */
  int32_t random_start;          /* 0: re-initialised envs start at t = 0 */
  float *obs_final, *obs_next;   /* [N, obs_stride] */
  int32_t obs_stride;
  float *reward;                 /* [N] */
  float *reward_parts;           /* [N,4] or NULL */
  uint8_t *terminated, *truncated; /* [N] */
} ss_imitation_io;
int ss_imitation_bind(struct ss_batch *b, const ss_imitation_io *io);
/* actions [N,nu]; rand [N,2] uniform [0,1) draws or NULL (finished envs are then left as they are: evaluation runs) */
int ss_imitation_step_fused(struct ss_batch *b, const float *actions, const float *rand, void *stream);

#ifdef __cplusplus
}
#endif
#endif
