// ss_imfused.h — the imitation task folded into the step launch (ss_imitation_step_fused, include/smplsim_motion.h).
//
// What SMPLSimImitationVecEnv.step otherwise does in six launches — ss_step, ss_imitation_step, and for the envs whose clip ended
// or that drifted off it: ss_motion_resample, ss_motion_state_at, a masked ss_reset, a masked ss_imitation_step — runs here inside
// the wavefront that stepped the env, on what it just wrote (body frames, velocities, cur_t), with the very same element / wave
// functions (ss_motion.h): lane = body for the clip lookups, lane 0 for the resampling.  The reset's mj_forward is the stepper's
// own fused reset pass (run_env, MODE_RESET with StateInit External) on the qpos / qvel this code wrote.
#pragma once
#include "ss_motion.h"

namespace ss {
namespace mo {

// host-side validation of a cooked clip library
inline const char *check_data(const ss_motion_data *d, bool cooking) {
  if (!d) return "null motion data";
  if (d->num_motions < 1 || d->num_frames < 1 || d->nbody < 1 || d->nbody > kMaxBodies) return "motion data: bad sizes";
  if (!d->length_starts || !d->motion_num_frames || !d->motion_dt || !d->motion_lengths) return "motion data: null clip table";
  if (!d->gts || !d->grs || !d->gvs || !d->gavs || !d->dof_pos || !d->dvs || !d->qpos || !d->qvel) return "motion data: null cooked array";
  if (cooking && (!d->frame_motion || !d->pose_aa || !d->trans || !d->offsets || !d->lrs)) return "motion data: null raw clip array";
  return nullptr;
}

struct ImFused {
  ImArgs im;                  // mask null; obs = task part of the obs_final rows; reward / parts / terminated / truncated of the step
  int32_t *ids;               // = im.ids, writable (re-initialisation)
  float *start_times;         // = im.start_times, writable
  const float *cdf;           // [num_motions] sampling CDF of the clips
  float truncate;             // sample_time's truncate_time
  int random_start;           // 0: re-initialised envs start their clip at t = 0
  float *obs_final, *obs_next;   // [N, im.obs_stride] rows = [self obs | task obs]: after the step / what the policy acts on next
  int obs_floats;             // floats per row that carry data (self + task)
  float *qpos, *qvel;         // the simulator's state rows (ss_state.qpos / qvel)
  int nq, nv;
};

// after the step pass of env `env` (its wave): imitation reward / flags / task observation; then either copy the row to obs_next
// or re-initialise the env from the clip library.  Returns true when the caller has to run the reset pass next.
template <class W>
SS_DEV bool fused_after_step(W *w, const ImFused *f, const float *rand, int env) {
  const int lane = w->lane();
  w->mem_fence();                                            // the step pass's body frames / velocities / cur_t / self observation
  const int flags = imitation_wave<W, 64>(w, f->im, env);
  const size_t row = (size_t)env * f->im.obs_stride;
  if (flags == 0 || !rand) {
    w->mem_fence();
    for (int i = lane; i < f->obs_floats; i += 64) f->obs_next[row + i] = f->obs_final[row + i];
    return false;
  }
  if (lane == 0) {                                           // MotionLibBase.sample_motions + sample_time for this env
    ResampleArgs r{f->im.d, nullptr, rand, f->cdf, f->truncate, f->im.N, f->ids, f->start_times};
    resample_elem(r, env);
    if (!f->random_start) f->start_times[env] = 0.f;
  }
  w->mem_fence();
  if (lane < f->im.d.nbody) {                                // reference-state init: the clip's qpos / qvel at the start time
    StateArgs s{};
    s.d = f->im.d; s.ids = f->ids; s.times = f->start_times; s.offset = f->im.offset; s.N = f->im.N;
    s.out.qpos = f->qpos; s.out.qvel = f->qvel;
    state_elem(s, env, lane);
  }
  w->mem_fence();
  return true;
}

// after the reset pass: the task observation of the re-initialised env (observation only, like the masked launch it replaces)
template <class W>
SS_DEV void fused_after_reset(W *w, const ImFused *f, int env) {
  w->mem_fence();
  imitation_wave<W, 64>(w, f->im, env, f->obs_next + (f->im.obs - f->obs_final), true);
}

}  // namespace mo
}  // namespace ss
