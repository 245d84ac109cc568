// ss_hdr.h — POD header shared by host code and the device kernels (no STL).
#pragma once
#include <stdint.h>

#include "../../include/smplsim_hip.h"

namespace ss {

constexpr int kWave = 64;
constexpr int kBodyC = 16;   // floats per body in the body-constant table
constexpr int kDofC = 12;    // floats per dof in the dof-constant table
constexpr int kCandC = 8;    // floats per contact candidate

// Header passed to the kernels by value: dims, table offsets (32-bit words into the shared blob),
// per-env LDS layout (float offsets) and physics scalars.
struct Hdr {
  int nb, nn, nv, nq, nu, ncand, nlev, nblev, nbox, nslot, maxlev;
  unsigned long long nkpack[2];      // (nodes in level L) - 1, 4 bits per level: level bounds by SALU shifts, no table/kernarg loads
  // shared-blob word offsets
  int o_dofc, o_boff, o_chainnode, o_ndepth, o_lev, o_bparent, o_sumsmall, o_sumbig, o_sumcover, n_sumsmall, n_sumbig, shared_words;
  // per-env LDS float offsets.  Z = solver region: Aown | IA (2 level buffers) | Ubuf | Wst ; aliases: contact records
  // at Z, R/r inside Wst, Gb and the body_accel scratch inside IA, Ad = Ubuf = An, V = Pb
  int l_q, l_v, l_a, l_tau, l_C, l_Pb, l_delta, l_diag, l_S, l_Ab, l_An, l_Aown, l_IA, l_Ubuf, l_Wst,
      l_R, l_r, l_Gb, l_tmp, l_V, l_Iown, ia_stride, env_floats;
  float dt, grav, margin, mu, solimp[5], K, B;   // K, B of aref (from solref, dmax)
  float qpos0_root[3];
};

// compiled kernel variants: 0 = SMPL-sized (<= 128 dofs / candidates, <= 64 contact slots, <= 8 nodes per tree level),
// 1 = SMPL-X/H-sized (<= 192 dofs / candidates, <= 128 slots, <= 16 nodes per level); -1 = none fits
inline int kernel_variant(const Hdr &h) {
  const int dofp = (h.nv + 63) / 64, candp = (h.ncand + 63) / 64, slotp = (h.nslot + 63) / 64, npass = (h.maxlev + 7) / 8;
  if (dofp <= 2 && candp <= 2 && slotp <= 1 && npass <= 1) return 0;
  if (dofp <= 3 && candp <= 3 && slotp <= 2 && npass <= 2) return 1;
  return -1;
}

constexpr int shape_stride(const Hdr &h) { return h.nb * kBodyC + ((h.nv + 3) & ~3); }   // floats per shape in the shaped bodyc array

enum { MODE_STEP = 0, MODE_SUBSTEP = 1, MODE_RESET = 2, MODE_KINEMATICS = 3, MODE_DEBUG_FORWARD = 4 };

// Everything a launch needs, passed to the kernel by value.
struct KArgs {
  Hdr h;
  ss_env_cfg cfg;
  ss_state st;
  const uint32_t *shared_g;   // shared tables (global copy)
  const float *bodyc;         // [nb][kBodyC]
  const float *candc;         // [ncand][kCandC]
  const int32_t *candb;       // [ncand]
  uint64_t illegal_mask;
  // per call
  int mode, nsub, obs_size;
  const float *actions;       // [N,nu]
  const float *task_rand;     // [N,2] or null
  const float *fall_actions;  // [N,3,nu] or null
  const uint8_t *mask;        // [N] or null
  unsigned long long *prof;   // optional stage-cycle accumulators (SS_PROFILE builds), else null
  const int32_t *order;       // optional [N] processing order of the envs (heavy first), or null
  int32_t *work_counter;      // device word, zeroed before each launch: persistent waves pull env ids from it
  float *obs, *reward;
  // fused autoreset (ss_step_autoreset): envs whose step ends an episode run the Default reset in the same launch;
  // obs2 receives the observation AFTER the (possible) reset for every env, task_rand2 feeds the reset's reset_task
  int fused_reset;
  float *obs2;
  const float *task_rand2;
  uint8_t *terminated, *truncated;
  float *out0, *out1, *out2;  // kinematics: xpos, xmat ; debug forward: M [N,nv,nv], bias [N,nv], qacc [N,nv]
  // per-env body shapes (ss_model_create_shapes): bodyc holds num_shapes consecutive blocks of [nb][kBodyC] body constants
  // followed by [nv] dof inverse weights (block stride shape_stride(h) floats), candc num_shapes consecutive tables;
  // st.shape_id [N] selects per env (null = single-shape model)
};

}  // namespace ss
