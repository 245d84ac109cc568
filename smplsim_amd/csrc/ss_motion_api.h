// ss_motion_api.h — host side of include/smplsim_motion.h, instantiated like ss_api.h with the HIP backend (product)
// or the emulator backend (unit tests).  A backend provides, each returning nullptr or an error string:
//   static const char *motion_cook(const ss::mo::CookArgs &, void *stream);      // three launches: fk, dof fix, velocities
//   static const char *motion_state(const ss::mo::StateArgs &, void *stream);
//   static const char *imitation(const ss::mo::ImArgs &, void *stream);
//   static const char *motion_resample(const ss::mo::ResampleArgs &, void *stream);
#pragma once
#include <cmath>
#include <string>

#include "ss_api.h"
#include "ss_motion.h"

namespace ss {
namespace mo {

// packs and validates the skeleton (the reference's MuJoCo body order is depth-first; the level of each body drives fk_wave)
inline bool pack_skeleton(const ss_skeleton *s, Skel *out, std::string *err) {
  if (!s || !s->parent || !s->smpl_2_mujoco) { *err = "null skeleton"; return false; }
  const int J = s->nbody;
  if (J < 1 || J > kMaxBodies) { *err = "skeleton: nbody must be in [1, 64]"; return false; }
  int depth[kMaxBodies];
  bool seen[kMaxBodies] = {};
  for (int j = 0; j < J; j++) {
    const int p = s->parent[j], m = s->smpl_2_mujoco[j];
    if (m < 0 || m >= J || seen[m]) { *err = "skeleton: smpl_2_mujoco is not a permutation"; return false; }
    seen[m] = true;
    if (j == 0) {
      if (p != -1) { *err = "skeleton: body 0 must be the root (parent -1)"; return false; }
      depth[0] = 0;
    } else {
      if (p < 0 || p >= j) { *err = "skeleton: parents must precede their children"; return false; }
      int anc = j - 1;                                       // depth-first order: the parent is body j-1 or one of its ancestors
      while (anc != p && anc > 0) anc = s->parent[anc];
      if (anc != p) { *err = "skeleton: bodies are not in depth-first order"; return false; }
      depth[j] = depth[p] + 1;
      if (depth[j] >= kMaxDepth) { *err = "skeleton: tree deeper than 16 levels"; return false; }
    }
    out->parent[j] = (int8_t)p;
    out->s2m[j] = (uint8_t)m;
    out->depth[j] = (uint8_t)depth[j];
    if (depth[j] > out->maxdepth) out->maxdepth = depth[j];
  }
  out->nb = J;
  return true;
}

}  // namespace mo
}  // namespace ss

#define SS_DEFINE_MOTION_API(BE)                                                                                     \
  extern "C" {                                                                                                       \
  int ss_motion_cook(const ss_skeleton *skel, const ss_motion_data *data, int32_t filter_vel, void *stream) {          \
    ss::mo::CookArgs a{};                                                                                            \
    std::string err;                                                                                                 \
    if (!ss::mo::pack_skeleton(skel, &a.sk, &err)) return ss_api<BE>::fail(SS_ERR_INVALID, err);                       \
    if (const char *e = ss::mo::check_data(data, true)) return ss_api<BE>::fail(SS_ERR_INVALID, e);                    \
    if (data->nbody != skel->nbody) return ss_api<BE>::fail(SS_ERR_INVALID, "motion data and skeleton disagree on nbody"); \
    a.d = *data; a.filter = filter_vel ? 1 : 0;                                                                        \
    double wsum = 0.0, wk[2 * ss::mo::kGaussRadius + 1];                                                               \
    for (int k = -ss::mo::kGaussRadius; k <= ss::mo::kGaussRadius; k++) { wk[k + ss::mo::kGaussRadius] = std::exp(-0.5 * k * k / 4.0); wsum += wk[k + ss::mo::kGaussRadius]; } \
    for (int k = 0; k < 2 * ss::mo::kGaussRadius + 1; k++) a.gw[k] = (float)(wk[k] / wsum);                            \
    const char *e = BE::motion_cook(a, stream);                                                                       \
    return e ? ss_api<BE>::fail(SS_ERR_HIP, e) : SS_OK;                                                               \
  }                                                                                                                  \
  int ss_motion_state_at(const ss_motion_data *data, const int32_t *ids, const float *times, const float *offset,      \
                         const uint8_t *mask, int32_t N, int32_t intervaled, const ss_motion_state *out, void *stream) { \
    if (const char *e = ss::mo::check_data(data, false)) return ss_api<BE>::fail(SS_ERR_INVALID, e);                   \
    if (!ids || !times || !out) return ss_api<BE>::fail(SS_ERR_INVALID, "null argument");                              \
    if (N < 1) return ss_api<BE>::fail(SS_ERR_INVALID, "N must be positive");                                          \
    ss::mo::StateArgs a{*data, ids, times, offset, mask, N, intervaled ? 1 : 0, *out};                                       \
    const char *e = BE::motion_state(a, stream);                                                                      \
    return e ? ss_api<BE>::fail(SS_ERR_HIP, e) : SS_OK;                                                               \
  }                                                                                                                  \
  int ss_motion_resample(const ss_motion_data *data, const uint8_t *mask, const float *rand, const float *cdf,        \
                         float truncate_time, int32_t N, int32_t *ids, float *start_times, void *stream) {            \
    if (const char *e = ss::mo::check_data(data, false)) return ss_api<BE>::fail(SS_ERR_INVALID, e);                   \
    if (!rand || !cdf || !ids || !start_times) return ss_api<BE>::fail(SS_ERR_INVALID, "null argument");               \
    if (N < 1 || truncate_time < 0.f) return ss_api<BE>::fail(SS_ERR_INVALID, "N must be positive, truncate_time >= 0"); \
    ss::mo::ResampleArgs a{*data, mask, rand, cdf, truncate_time, N, ids, start_times};                               \
    const char *e = BE::motion_resample(a, stream);                                                                  \
    return e ? ss_api<BE>::fail(SS_ERR_HIP, e) : SS_OK;                                                               \
  }                                                                                                                  \
  int ss_imitation_step(const ss_motion_data *data, const ss_imitation_cfg *cfg, const int32_t *ids, const float *start_times, \
                        const int32_t *cur_t, const float *offset, const uint8_t *mask, int32_t N, const float *xpos,  \
                        const float *xmat, const float *body_vel, float *task_obs, int32_t obs_stride, float *reward,  \
                        float *parts, uint8_t *terminated, uint8_t *truncated, void *stream) {                        \
    if (const char *e = ss::mo::check_data(data, false)) return ss_api<BE>::fail(SS_ERR_INVALID, e);                   \
    if (!cfg || !ids || !start_times || !xpos || !xmat || !body_vel || !task_obs)                                      \
      return ss_api<BE>::fail(SS_ERR_INVALID, "null argument");                                                      \
    if (N < 1) return ss_api<BE>::fail(SS_ERR_INVALID, "N must be positive");                                          \
    if (obs_stride < 24 * data->nbody) return ss_api<BE>::fail(SS_ERR_INVALID, "obs_stride smaller than the task observation"); \
    ss::mo::ImArgs a{*data, *cfg, ids, start_times, cur_t, offset, mask, N, xpos, xmat, body_vel, task_obs, obs_stride, reward, parts, \
                     terminated, truncated};                                                                         \
    const char *e = BE::imitation(a, stream);                                                                         \
    return e ? ss_api<BE>::fail(SS_ERR_HIP, e) : SS_OK;                                                               \
  }                                                                                                                  \
  }
