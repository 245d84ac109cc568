// ss_motion.h — motion-library kernels: cooking the clips, per-step lookup, imitation obs/reward.
//
// Written, like ss_kernel.h, so that the identical code is compiled by hipcc for gfx950 and by g++ for the CPU
// emulator under tests/wave_emu (unit-test infrastructure only).  Element functions (one (frame, body) / one (env, body))
// have no cross-lane traffic; the wave functions use the W policy for sync(), any() and shfl_xor().
//
// All of this is HBM-bound gather/scatter over flat frame arrays (no GEMM shape anywhere); what matters is that
// consecutive lanes touch consecutive bodies of one frame so that a wave reads whole 288..1248-byte rows.
#pragma once
#include <math.h>
#include <stdint.h>

#include "../../include/smplsim_hip.h"
#include "../../include/smplsim_motion.h"

#ifndef SS_DEV
#if defined(__HIPCC__)
#define SS_DEV __device__ __forceinline__
#else
#define SS_DEV inline
#endif
#endif

namespace ss {
namespace mo {

constexpr int kMaxBodies = 64;
constexpr int kMaxDepth = 16;
constexpr int kXformStride = 13;         // world rotation (9) + position (3) of one body in LDS, padded to an odd stride
constexpr int kGaussRadius = 8;          // scipy gaussian_filter1d(sigma=2): int(4 * 2 + 0.5)
constexpr float kPi = 3.14159265358979323846f;

struct Skel { int nb, maxdepth; int8_t parent[kMaxBodies]; uint8_t s2m[kMaxBodies]; uint8_t depth[kMaxBodies]; };
struct CookArgs { Skel sk; ss_motion_data d; int filter; float gw[2 * kGaussRadius + 1]; };
struct StateArgs { ss_motion_data d; const int32_t *ids; const float *times; const float *offset; const uint8_t *mask; int N; int intervaled; ss_motion_state out; };
struct ImArgs {
  ss_motion_data d; ss_imitation_cfg c; const int32_t *ids; const float *start_times; const int32_t *cur_t; const float *offset;
  const uint8_t *mask; int N;
  const float *xpos, *xmat, *body_vel; float *obs; int obs_stride; float *reward, *parts; uint8_t *terminated, *truncated;
};
struct ResampleArgs { ss_motion_data d; const uint8_t *mask; const float *rand; const float *cdf; float truncate; int N; int32_t *ids; float *start_times; };

struct Q { float w, x, y, z; };

SS_DEV float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// pytorch3d_transforms.axis_angle_to_quaternion (:546-569)
SS_DEV Q q_from_aa(float ax, float ay, float az) {
  const float ang = sqrtf(ax * ax + ay * ay + az * az);
  const float k = ang < 1e-6f ? 0.5f - ang * ang / 48.f : sinf(0.5f * ang) / ang;
  return Q{cosf(0.5f * ang), ax * k, ay * k, az * k};
}
// quaternion_to_matrix (:48-76), row-major
SS_DEV void q_to_mat(const Q &q, float *m) {
  const float s = 2.f / (q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  m[0] = 1.f - s * (q.y * q.y + q.z * q.z); m[1] = s * (q.x * q.y - q.z * q.w); m[2] = s * (q.x * q.z + q.y * q.w);
  m[3] = s * (q.x * q.y + q.z * q.w); m[4] = 1.f - s * (q.x * q.x + q.z * q.z); m[5] = s * (q.y * q.z - q.x * q.w);
  m[6] = s * (q.x * q.z - q.y * q.w); m[7] = s * (q.y * q.z + q.x * q.w); m[8] = 1.f - s * (q.x * q.x + q.y * q.y);
}
// matrix_to_quaternion (:139-185): best-conditioned of the four candidates, its own component positive
SS_DEV Q mat_to_q(const float *m) {
  const float t0 = 1.f + m[0] + m[4] + m[8], t1 = 1.f + m[0] - m[4] - m[8], t2 = 1.f - m[0] + m[4] - m[8], t3 = 1.f - m[0] - m[4] + m[8];
  const float a0 = t0 > 0.f ? sqrtf(t0) : 0.f, a1 = t1 > 0.f ? sqrtf(t1) : 0.f, a2 = t2 > 0.f ? sqrtf(t2) : 0.f, a3 = t3 > 0.f ? sqrtf(t3) : 0.f;
  int pick = 0; float best = a0;
  if (a1 > best) { pick = 1; best = a1; }
  if (a2 > best) { pick = 2; best = a2; }
  if (a3 > best) { pick = 3; best = a3; }
  const float den = 2.f * fmaxf(best, 0.1f);
  Q q;
  if (pick == 0) q = Q{a0 * a0, m[7] - m[5], m[2] - m[6], m[3] - m[1]};
  else if (pick == 1) q = Q{m[7] - m[5], a1 * a1, m[3] + m[1], m[2] + m[6]};
  else if (pick == 2) q = Q{m[2] - m[6], m[3] + m[1], a2 * a2, m[5] + m[7]};
  else q = Q{m[3] - m[1], m[6] + m[2], m[7] + m[5], a3 * a3};
  return Q{q.w / den, q.x / den, q.y / den, q.z / den};
}
SS_DEV Q q_mul(const Q &a, const Q &b) {
  return Q{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
           a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
SS_DEV Q q_conj(const Q &a) { return Q{a.w, -a.x, -a.y, -a.z}; }
// np_transform_utils.quat_rotate (:23-32)
SS_DEV void q_rot(const Q &q, const float *v, float *o) {
  const float a = 2.f * q.w * q.w - 1.f, d = 2.f * (q.x * v[0] + q.y * v[1] + q.z * v[2]);
  const float cx = q.y * v[2] - q.z * v[1], cy = q.z * v[0] - q.x * v[2], cz = q.x * v[1] - q.y * v[0];
  o[0] = v[0] * a + cx * q.w * 2.f + q.x * d; o[1] = v[1] * a + cy * q.w * 2.f + q.y * d; o[2] = v[2] * a + cz * q.w * 2.f + q.z * d;
}
// quat_to_tan_norm (:86-96): the rotated x and z axes
SS_DEV void q_tan_norm(const Q &q, float *o) {
  const float ex[3] = {1.f, 0.f, 0.f}, ez[3] = {0.f, 0.f, 1.f};
  q_rot(q, ex, o); q_rot(q, ez, o + 3);
}
SS_DEV Q ldq(const float *p) { return Q{p[0], p[1], p[2], p[3]}; }
SS_DEV void stq(float *p, const Q &q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }

// ------------------------------------------------------------------------------------------------ cooking
// Forward kinematics (forward_kinematics_batch, torch_smpl_humanoid_batch.py:166-196): lane j of an LPE-lane group holds
// body j of one frame, so every global load and store of a group is one contiguous row of the frame.  The tree is walked
// level by level; a body reads its parent's world transform from the wave's LDS tile (xf[(group * LPE + body) * 13]).
template <class W, int LPE>
SS_DEV void fk_wave(W *w, const CookArgs &a, int wave_id, float *xf) {
  const ss_motion_data &d = a.d;
  const int J = a.sk.nb, lane = w->lane(), j = lane % LPE, g = lane / LPE, f = wave_id * (64 / LPE) + g, nq = 7 + 3 * (J - 1);
  const bool act = f < d.num_frames && j < J;
  float lm[9], R[9], P[3] = {0.f, 0.f, 0.f};
  int p = -1, lev_j = -1;
  const float *off = nullptr;
  if (act) {
    const int s = a.sk.s2m[j];
    p = a.sk.parent[j]; lev_j = a.sk.depth[j];
    off = d.offsets + ((size_t)d.frame_motion[f] * J + j) * 3;
    const float *aa = d.pose_aa + ((size_t)f * J + s) * 3;
    const Q q = q_from_aa(aa[0], aa[1], aa[2]);
    stq(d.lrs + ((size_t)f * J + s) * 4, q);
    q_to_mat(q, lm);
    if (j >= 1) {                                            // matrix_to_euler_angles(.., "XYZ") (:331-364)
      float *dp = d.dof_pos + ((size_t)f * (J - 1) + (j - 1)) * 3;
      dp[0] = atan2f(-lm[5], lm[8]); dp[1] = asinf(clampf(lm[2], -1.f, 1.f)); dp[2] = atan2f(-lm[1], lm[0]);
    }
    if (s == 0) stq(d.qpos + (size_t)f * nq + 3, q);         // qpos[3:7] = pose_quat[..., 0, :] (SMPL joint 0)
  }
  float *mine = xf + (g * LPE + j) * kXformStride;
  for (int lev = 0; lev <= a.sk.maxdepth; lev++) {
    if (lev_j == lev) {
      if (p < 0) {
        const float *tr = d.trans + (size_t)f * 3;
        for (int c = 0; c < 9; c++) R[c] = lm[c];
        for (int c = 0; c < 3; c++) P[c] = tr[c] + off[c];
      } else {
        const float *pp = xf + (g * LPE + p) * kXformStride;
        for (int r = 0; r < 3; r++) {
          P[r] = pp[3 * r] * off[0] + pp[3 * r + 1] * off[1] + pp[3 * r + 2] * off[2] + pp[9 + r];
          for (int c = 0; c < 3; c++) R[3 * r + c] = pp[3 * r] * lm[c] + pp[3 * r + 1] * lm[3 + c] + pp[3 * r + 2] * lm[6 + c];
        }
      }
      for (int c = 0; c < 9; c++) mine[c] = R[c];
      for (int c = 0; c < 3; c++) mine[9 + c] = P[c];
    }
    w->sync();
  }
  if (act) {
    float *gp = d.gts + ((size_t)f * J + j) * 3;
    gp[0] = P[0]; gp[1] = P[1]; gp[2] = P[2];
    stq(d.grs + ((size_t)f * J + j) * 4, mat_to_q(R));
    if (j == 0) { float *qp = d.qpos + (size_t)f * nq; qp[0] = P[0]; qp[1] = P[1]; qp[2] = P[2]; }
  }
}

SS_DEV float wrap_pi(float x) { if (x > kPi) x -= 2.f * kPi; if (x < -kPi) x += 2.f * kPi; return x; }

// fix_continous_dof (pytorch3d_transforms.py:749-778) for one clip: one lane per joint, frames in sequence; also copies
// the (fixed) Euler angles into qpos[7:].
template <class W>
SS_DEV void dof_fix_clip(W *w, const CookArgs &a, int m) {
  const ss_motion_data &d = a.d;
  const int J1 = a.sk.nb - 1, nq = 7 + 3 * J1, s0 = d.length_starts[m], T = d.motion_num_frames[m];
  const int lane = w->lane();
  const bool act = lane < J1;
  float p0 = 0.f, p1 = 0.f, p2 = 0.f;
  for (int t = 0; t < T; t++) {
    float *dp = d.dof_pos + ((size_t)(s0 + t) * J1 + lane) * 3;
    float c0 = act ? dp[0] : 0.f, c1 = act ? dp[1] : 0.f, c2 = act ? dp[2] : 0.f;
    if (t >= 1 && t < T - 1) {
      for (int tries = 0; tries < 2; tries++) {
        const float d0 = fabsf(c0 - p0), d1 = fabsf(c1 - p1), d2 = fabsf(c2 - p2);
        if (!w->any(act && fmaxf(d0, fmaxf(d1, d2)) >= 3.f)) break;
        if (act && d0 + d1 + d2 >= 3.f) {
          c0 = kPi + c0; c1 = kPi - c1; c2 = kPi + c2;
          c0 = wrap_pi(c0); c1 = wrap_pi(c1); c2 = wrap_pi(c2);
        }
      }
    }
    if (act) {
      dp[0] = c0; dp[1] = c1; dp[2] = c2;
      float *qp = d.qpos + (size_t)(s0 + t) * nq + 7 + 3 * lane;
      qp[0] = c0; qp[1] = c1; qp[2] = c2;
    }
    p0 = c0; p1 = c1; p2 = c2;
  }
}

// _compute_velocity / _compute_angular_velocity (torch_smpl_humanoid_batch.py:198-221) + dof_vels, qvel (:148-161): finite
// differences inside the clip, last difference repeated (linear) or zero (angular, whose last quaternion difference is the
// identity), then the 17-tap Gaussian with the edge samples repeated.  One wave owns kVelTile consecutive frames: it first
// puts the raw velocities of those frames and of the 8 frames on either side into LDS (raw[(slot * J + body) * 6]), then
// every output is 17 LDS taps — the acos / normalisation work is done ~2x per element instead of 17x.
constexpr int kVelTile = 16;
SS_DEV void raw_velocity(const CookArgs &a, int g, int j, float *o) {
  const ss_motion_data &d = a.d;
  const int J = a.sk.nb, m = d.frame_motion[g], s0 = d.length_starts[m], T = d.motion_num_frames[m], t = g - s0;
  const float dt = d.motion_dt[m];
  for (int c = 0; c < 6; c++) o[c] = 0.f;
  if (T < 2) return;
  const int b = t < T - 2 ? t : T - 2;
  const float *p0 = d.gts + ((size_t)(s0 + b) * J + j) * 3, *p1 = p0 + (size_t)J * 3;
  for (int c = 0; c < 3; c++) o[c] = (p1[c] - p0[c]) / dt;
  if (t <= T - 2) {
    const float *g0 = d.grs + ((size_t)g * J + j) * 4;
    Q dq = q_mul(ldq(g0 + (size_t)J * 4), q_conj(ldq(g0)));
    if (dq.w < 0.f) dq = Q{-dq.w, -dq.x, -dq.y, -dq.z};                         // quat_normalize = unit(pos(.))
    const float n = fmaxf(sqrtf(dq.w * dq.w + dq.x * dq.x + dq.y * dq.y + dq.z * dq.z), 1e-9f);
    dq = Q{dq.w / n, dq.x / n, dq.y / n, dq.z / n};
    const float ang = acosf(clampf(2.f * dq.w * dq.w - 1.f, -1.f, 1.f));        // quat_angle_axis
    const float an = fmaxf(sqrtf(dq.x * dq.x + dq.y * dq.y + dq.z * dq.z), 1e-10f);
    o[3] = dq.x / an * ang / dt; o[4] = dq.y / an * ang / dt; o[5] = dq.z / an * ang / dt;
  }
}

template <class W>
SS_DEV void vel_wave(W *w, const CookArgs &a, int wave_id, float *raw) {
  const ss_motion_data &d = a.d;
  const int J = a.sk.nb, J1 = J - 1, F = d.num_frames, lane = w->lane();
  const int f_lo = wave_id * kVelTile, g_lo = f_lo - kGaussRadius, nslot = kVelTile + 2 * kGaussRadius;
  for (int idx = lane; idx < nslot * J; idx += 64) {
    const int g = g_lo + idx / J;
    if (g >= 0 && g < F) raw_velocity(a, g, idx % J, raw + (size_t)idx * 6);
  }
  w->sync();
  const int r = a.filter ? kGaussRadius : 0;
  for (int idx = lane; idx < kVelTile * J; idx += 64) {
    const int f = f_lo + idx / J, j = idx % J;
    if (f >= F) break;
    const int m = d.frame_motion[f], s0 = d.length_starts[m], T = d.motion_num_frames[m], t = f - s0;
    const float dt = d.motion_dt[m];
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = -r; k <= r; k++) {
      int tt = t + k;
      tt = tt < 0 ? 0 : (tt > T - 1 ? T - 1 : tt);
      const float wgt = a.filter ? a.gw[k + kGaussRadius] : 1.f;
      const float *x = raw + ((size_t)(s0 + tt - g_lo) * J + j) * 6;
      for (int c = 0; c < 6; c++) acc[c] += wgt * x[c];
    }
    float *gv = d.gvs + ((size_t)f * J + j) * 3, *ga = d.gavs + ((size_t)f * J + j) * 3;
    for (int c = 0; c < 3; c++) { gv[c] = acc[c]; ga[c] = acc[3 + c]; }
    float *qv = d.qvel + (size_t)f * (6 + 3 * J1);
    if (j >= 1) {
      const int b = t < T - 2 ? t : T - 2;
      float *dv = d.dvs + ((size_t)f * J1 + (j - 1)) * 3;
      for (int c = 0; c < 3; c++) {
        float v = 0.f;
        if (T >= 2) { const float *x0 = d.dof_pos + ((size_t)(s0 + b) * J1 + (j - 1)) * 3; v = (x0[(size_t)J1 * 3 + c] - x0[c]) / dt; }
        dv[c] = v; qv[6 + 3 * (j - 1) + c] = v;
      }
    } else {                                                 // qvel[0:6]: world linear velocity, BODY-frame angular velocity
      float Rm[9];
      q_to_mat(ldq(d.lrs + ((size_t)f * J + a.sk.s2m[0]) * 4), Rm);
      for (int c = 0; c < 3; c++) { qv[c] = acc[c]; qv[3 + c] = Rm[c] * acc[3] + Rm[3 + c] * acc[4] + Rm[6 + c] * acc[5]; }
    }
  }
}

// ------------------------------------------------------------------------------------------------ lookup
// MotionLibBase._calc_frame_blend (motion_lib_base.py:442-453), integer frames
SS_DEV void frame_blend(const ss_motion_data &d, int id, float time, int *f0, int *f1, float *blend) {
  const int nf = d.motion_num_frames[id], s0 = d.length_starts[id];
  if (nf < 2) { *f0 = *f1 = s0; *blend = 0.f; return; }
  const float len = d.motion_lengths[id], dt = d.motion_dt[id];
  const float phase = clampf(time / len, 0.f, 1.f), tm = time < 0.f ? 0.f : time;
  const int i0 = (int)(phase * (float)(nf - 1)), i1 = i0 + 1 < nf - 1 ? i0 + 1 : nf - 1;
  *blend = clampf((tm - (float)i0 * dt) / dt, 0.f, 1.f);
  *f0 = s0 + i0; *f1 = s0 + i1;
}
// frame of get_motion_state_intervaled (:317-319): the NumPy _calc_frame_blend keeps FLOAT frame numbers
SS_DEV int frame_intervaled(const ss_motion_data &d, int id, float time) {
  const int nf = d.motion_num_frames[id], s0 = d.length_starts[id];
  if (nf < 2) return s0;
  const float len = d.motion_lengths[id], dt = d.motion_dt[id];
  const float phase = clampf(time / len, 0.f, 1.f), tm = time < 0.f ? 0.f : time;
  const float i0 = phase * (float)(nf - 1), i1 = fminf(i0 + 1.f, (float)(nf - 1));
  const float b = clampf((tm - i0 * dt) / dt, 0.f, 1.f);
  return s0 + (int)((1.f - b) * i0 + b * i1);
}
// torch_utils.slerp (smpl_sim/utils/torch_utils.py:405-426)
SS_DEV Q slerp(const Q &q0, Q q1, float t) {
  float c = q0.w * q1.w + q0.x * q1.x + q0.y * q1.y + q0.z * q1.z;
  if (c < 0.f) { q1 = Q{-q1.w, -q1.x, -q1.y, -q1.z}; c = -c; }
  if (c >= 1.f) return q0;
  const float sn = sqrtf(1.f - c * c);
  if (sn < 0.001f) return Q{0.5f * q0.w + 0.5f * q1.w, 0.5f * q0.x + 0.5f * q1.x, 0.5f * q0.y + 0.5f * q1.y, 0.5f * q0.z + 0.5f * q1.z};
  const float half = acosf(c), ra = sinf((1.f - t) * half) / sn, rb = sinf(t * half) / sn;
  return Q{ra * q0.w + rb * q1.w, ra * q0.x + rb * q1.x, ra * q0.y + rb * q1.y, ra * q0.z + rb * q1.z};
}

struct BodyRef { float p[3]; Q q; float v[3]; float w[3]; };
SS_DEV void sample_body(const ss_motion_data &d, int f0, int f1, float b, int j, const float *off, BodyRef *o) {
  const int J = d.nbody;
  const size_t i0 = (size_t)f0 * J + j, i1 = (size_t)f1 * J + j;
  const float a = 1.f - b;
  for (int c = 0; c < 3; c++) {
    o->p[c] = a * d.gts[i0 * 3 + c] + b * d.gts[i1 * 3 + c] + (off ? off[c] : 0.f);
    o->v[c] = a * d.gvs[i0 * 3 + c] + b * d.gvs[i1 * 3 + c];
    o->w[c] = a * d.gavs[i0 * 3 + c] + b * d.gavs[i1 * 3 + c];
  }
  o->q = slerp(ldq(d.grs + i0 * 4), ldq(d.grs + i1 * 4), b);
}

// get_motion_state / get_motion_state_intervaled for (env n, body j)
SS_DEV void state_elem(const StateArgs &a, int n, int j) {
  if (a.mask && !a.mask[n]) return;
  const ss_motion_data &d = a.d;
  const ss_motion_state &o = a.out;
  const int J = d.nbody, J1 = J - 1, nq = 7 + 3 * J1, nv = 6 + 3 * J1, id = a.ids[n];
  const float time = a.times[n];
  const float *off = a.offset ? a.offset + (size_t)n * 3 : nullptr;
  int f0, f1; float b;
  if (a.intervaled) { f0 = f1 = frame_intervaled(d, id, time); b = 0.f; }
  else frame_blend(d, id, time, &f0, &f1, &b);
  BodyRef r;
  if (a.intervaled) {                                        // plain copy of one frame (no slerp renormalisation effects)
    const size_t i0 = (size_t)f0 * J + j;
    for (int c = 0; c < 3; c++) { r.p[c] = d.gts[i0 * 3 + c] + (off ? off[c] : 0.f); r.v[c] = d.gvs[i0 * 3 + c]; r.w[c] = d.gavs[i0 * 3 + c]; }
    r.q = ldq(d.grs + i0 * 4);
  } else sample_body(d, f0, f1, b, j, off, &r);
  const size_t nj = (size_t)n * J + j;
  if (o.rg_pos) for (int c = 0; c < 3; c++) o.rg_pos[nj * 3 + c] = r.p[c];
  if (o.rb_rot) stq(o.rb_rot + nj * 4, r.q);
  if (o.body_vel) for (int c = 0; c < 3; c++) o.body_vel[nj * 3 + c] = r.v[c];
  if (o.body_ang_vel) for (int c = 0; c < 3; c++) o.body_ang_vel[nj * 3 + c] = r.w[c];
  if (j == 0) {
    if (o.root_pos) for (int c = 0; c < 3; c++) o.root_pos[(size_t)n * 3 + c] = r.p[c];
    if (o.root_rot) stq(o.root_rot + (size_t)n * 4, r.q);
    if (o.root_vel) for (int c = 0; c < 3; c++) o.root_vel[(size_t)n * 3 + c] = r.v[c];
    if (o.root_ang_vel) for (int c = 0; c < 3; c++) o.root_ang_vel[(size_t)n * 3 + c] = r.w[c];
    if (a.intervaled) {
      if (o.qpos) for (int c = 0; c < 7; c++) o.qpos[(size_t)n * nq + c] = d.qpos[(size_t)f0 * nq + c];   // as in the reference: no offset
      if (o.qvel) for (int c = 0; c < 6; c++) o.qvel[(size_t)n * nv + c] = d.qvel[(size_t)f0 * nv + c];
    } else {
      if (o.qpos) { for (int c = 0; c < 3; c++) o.qpos[(size_t)n * nq + c] = r.p[c]; stq(o.qpos + (size_t)n * nq + 3, r.q); }
      if (o.qvel) {
        float Rm[9];
        q_to_mat(r.q, Rm);
        for (int c = 0; c < 3; c++) {
          o.qvel[(size_t)n * nv + c] = r.v[c];
          o.qvel[(size_t)n * nv + 3 + c] = Rm[c] * r.w[0] + Rm[3 + c] * r.w[1] + Rm[6 + c] * r.w[2];
        }
      }
    }
  } else {
    const float a0 = 1.f - b;
    for (int c = 0; c < 3; c++) {
      const size_t k0 = ((size_t)f0 * J1 + (j - 1)) * 3 + c, k1 = ((size_t)f1 * J1 + (j - 1)) * 3 + c;
      const float dp = a0 * d.dof_pos[k0] + b * d.dof_pos[k1], dv = a0 * d.dvs[k0] + b * d.dvs[k1];
      const size_t on = (size_t)n * 3 * J1 + 3 * (j - 1) + c;
      if (o.dof_pos) o.dof_pos[on] = dp;
      if (o.dof_vel) o.dof_vel[on] = dv;
      if (o.qpos) o.qpos[(size_t)n * nq + 7 + 3 * (j - 1) + c] = dp;
      if (o.qvel) o.qvel[(size_t)n * nv + 6 + 3 * (j - 1) + c] = dv;
    }
  }
}

// MotionLibBase.sample_motions + sample_time (motion_lib_base.py:277-292) for the envs whose mask byte is set: clip by
// inverse CDF of the batch sampling probabilities from rand[n,0], start time = rand[n,1] * (length - truncate)
SS_DEV void resample_elem(const ResampleArgs &a, int n) {
  if (a.mask && !a.mask[n]) return;
  const int M = a.d.num_motions;
  const float u = a.rand[2 * n];
  int lo = 0, hi = M - 1;                                    // smallest m with cdf[m] > u
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.cdf[mid] > u) hi = mid; else lo = mid + 1; }
  a.ids[n] = lo;
  a.start_times[n] = a.rand[2 * n + 1] * fmaxf(a.d.motion_lengths[lo] - a.truncate, 0.f);
}

// ------------------------------------------------------------------------------------------------ imitation task
// heading of the simulated root, the convention of the self observation (compute_humanoid_self_obs_v2,
// humanoid_env.py:644-647: remove_base_rot, then calc_heading_quat_inv)
SS_DEV Q heading_inv(const Q &root) {
  const Q q = q_mul(root, Q{0.5f, -0.5f, -0.5f, -0.5f});
  const float ex[3] = {1.f, 0.f, 0.f};
  float x[3];
  q_rot(q, ex, x);
  const float h = atan2f(x[1], x[0]);
  return Q{cosf(-0.5f * h), 0.f, 0.f, sinf(-0.5f * h)};
}

// LPE lanes per env (32 for J <= 32, else 64): lane j of an env's group handles body j; the five per-body error terms
// are summed over the group with a shfl_xor ladder, lane 0 of the group writes reward / termination.
// Returns, in every lane of an env's group, bit 0 = terminated, bit 1 = truncated (0 for skipped envs).
template <class W, int LPE>
SS_DEV int imitation_wave(W *w, const ImArgs &a, int wave_id, float *obs_rows = nullptr, bool obs_only = false) {
  const ss_motion_data &d = a.d;
  const int J = d.nbody, lane = w->lane(), j = lane % LPE, n = wave_id * (64 / LPE) + lane / LPE;
  const bool act = n < a.N && j < J && !(a.mask && !a.mask[n]);
  float e_pos = 0.f, e_rot = 0.f, e_vel = 0.f, e_ang = 0.f, dist = 0.f, time = 0.f;
  if (act) {
    const int id = a.ids[n];
    time = a.start_times[n] + (a.cur_t ? (float)a.cur_t[n] * a.c.obs_dt : 0.f);
    const float *off = a.offset ? a.offset + (size_t)n * 3 : nullptr;
    const size_t nj = (size_t)n * J + j;
    const float *sp = a.xpos + nj * 3, *sv = a.body_vel + nj * 6, *rp = a.xpos + (size_t)n * J * 3;
    const Q sq = mat_to_q(a.xmat + nj * 9), rq = mat_to_q(a.xmat + (size_t)n * J * 9);
    int f0, f1; float b;
    BodyRef r;
    // tracking error against the clip at `time`
    frame_blend(d, id, time, &f0, &f1, &b);
    sample_body(d, f0, f1, b, j, off, &r);
    float dp[3], dv[3], dw[3];
    for (int c = 0; c < 3; c++) { dp[c] = r.p[c] - sp[c]; dv[c] = r.v[c] - sv[c]; dw[c] = r.w[c] - sv[3 + c]; }
    const float d2 = dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2];
    e_pos = d2; dist = sqrtf(d2);
    e_vel = dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2];
    e_ang = dw[0] * dw[0] + dw[1] * dw[1] + dw[2] * dw[2];
    Q dq = q_mul(r.q, q_conj(sq));
    // angle = 2 atan2(|vec|, |w|): the same value as 2 acos(|w|/|q|), without its loss of precision near zero
    const float ang = 2.f * atan2f(sqrtf(dq.x * dq.x + dq.y * dq.y + dq.z * dq.z), fabsf(dq.w));
    e_rot = ang * ang;
    // task observation against the clip at `time + obs_dt`
    frame_blend(d, id, time + a.c.obs_dt, &f0, &f1, &b);
    sample_body(d, f0, f1, b, j, off, &r);
    const Q hi = heading_inv(rq), hq = q_conj(hi);
    float *ob = (obs_rows ? obs_rows : a.obs) + (size_t)n * a.obs_stride;   // obs_rows: another set of rows than the bound one
    float t3[3], o3[3], o6[6];
    for (int c = 0; c < 3; c++) t3[c] = r.p[c] - sp[c];
    q_rot(hi, t3, o3); for (int c = 0; c < 3; c++) ob[3 * j + c] = o3[c];
    dq = q_mul(q_mul(hi, q_mul(r.q, q_conj(sq))), hq);
    q_tan_norm(dq, o6); for (int c = 0; c < 6; c++) ob[3 * J + 6 * j + c] = o6[c];
    for (int c = 0; c < 3; c++) t3[c] = r.v[c] - sv[c];
    q_rot(hi, t3, o3); for (int c = 0; c < 3; c++) ob[9 * J + 3 * j + c] = o3[c];
    for (int c = 0; c < 3; c++) t3[c] = r.w[c] - sv[3 + c];
    q_rot(hi, t3, o3); for (int c = 0; c < 3; c++) ob[12 * J + 3 * j + c] = o3[c];
    for (int c = 0; c < 3; c++) t3[c] = r.p[c] - rp[c];
    q_rot(hi, t3, o3); for (int c = 0; c < 3; c++) ob[15 * J + 3 * j + c] = o3[c];
    q_tan_norm(q_mul(hi, r.q), o6); for (int c = 0; c < 6; c++) ob[18 * J + 6 * j + c] = o6[c];
  }
  for (int mask = 1; mask < LPE; mask <<= 1) {
    e_pos += w->shfl_xor(e_pos, mask); e_rot += w->shfl_xor(e_rot, mask); e_vel += w->shfl_xor(e_vel, mask);
    e_ang += w->shfl_xor(e_ang, mask); dist += w->shfl_xor(dist, mask);
  }
  int flags = 0;
  if (obs_only) return 0;
  if (act && j == 0) {
    const float ij = 1.f / (float)J;
    const float r0 = expf(-a.c.k_pos * e_pos * ij * (1.f / 3.f)), r1 = expf(-a.c.k_rot * e_rot * ij);
    const float r2 = expf(-a.c.k_vel * e_vel * ij * (1.f / 3.f)), r3 = expf(-a.c.k_ang_vel * e_ang * ij * (1.f / 3.f));
    if (a.reward) a.reward[n] = a.c.w_pos * r0 + a.c.w_rot * r1 + a.c.w_vel * r2 + a.c.w_ang_vel * r3;
    if (a.parts) { a.parts[4 * n] = r0; a.parts[4 * n + 1] = r1; a.parts[4 * n + 2] = r2; a.parts[4 * n + 3] = r3; }
    const int term = dist * ij > a.c.termination_distance ? 1 : 0;
    const int trunc = time + a.c.obs_dt >= d.motion_lengths[a.ids[n]] ? 1 : 0;                  // no later frame to look ahead to
    if (a.terminated) a.terminated[n] = (uint8_t)term;
    if (a.truncated) a.truncated[n] = (uint8_t)trunc;
    flags = term | (trunc << 1);
  }
  for (int mask = 1; mask < LPE; mask <<= 1) flags |= w->shfl_xor_i(flags, mask);
  return flags;
}

}  // namespace mo
}  // namespace ss
