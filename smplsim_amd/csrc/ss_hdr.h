// ss_hdr.h — POD header shared by host code and the device kernels (no STL).
#pragma once
#include <stdint.h>

#include "../../include/smplsim_hip.h"

// Scalar type of the kernel.  The product is float32.  -DSS_F64 (tests/wave_emu only) instantiates the same source in float64
// with the Newton solve run to convergence: the triage build that separates float32 rounding from formulation differences
// against the float64 oracle (tests/test_parity_f64.py).
#ifdef SS_F64
namespace ss { typedef double real; }
#define SS_M(fn) fn
#define SS_LS_TOL_EFF 1e-13                  /* "exact" line search */
#define SS_ROUND_REL 2.2e-15                 /* rounding level of the line-search terms (10 eps) */
#define SS_MOVE_REL 4e-16
#define SS_MOVE_ABS 1e-24
#define SS_LS_MAXIT 60
#else
namespace ss { typedef float real; }
#define SS_M(fn) fn##f
#define SS_LS_TOL_EFF SS_LS_TOL
#define SS_ROUND_REL 2e-6f
#define SS_MOVE_REL 4e-7f
#define SS_MOVE_ABS 1e-12f
#define SS_LS_MAXIT 16
#endif

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
  // (o_dofc, o_boff: real-valued tables, offsets in reals from the start of the blob; the layout of this struct is part of
  // the kernel's register allocation — one more field here cost 850 SGPR reloads in the step kernel)
  int o_dofc, o_boff, o_chainnode, o_ndepth, o_lev, o_bparent, o_sumsmall, o_sumbig, o_sumcover, n_sumsmall, n_sumbig, shared_words;
  // per-env LDS float offsets.  Z = solver region: Aown | IA (2 level buffers) | Ubuf | Wst ; aliases: contact records
  // at Z, R/r inside Wst, Gb and the body_accel scratch inside IA, Ad = Ubuf = An, V = Pb
  int l_q, l_v, l_a, l_tau, l_C, l_Pb, l_delta, l_diag, l_S, l_Ab, l_An, l_Aown, l_IA, l_Ubuf, l_Wst,
      l_R, l_r, l_Gb, l_tmp, l_V, l_Iown, ia_stride, env_floats;
  real dt, grav, margin, mu, solimp[5], K, B;   // K, B of aref (from solref, dmax)
  real qpos0_root[3];
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
  const real *bodyc;          // [nb][kBodyC]
  const real *candc;          // [ncand][kCandC]
  const int32_t *candb;       // [ncand]
  uint64_t illegal_mask;
  // per call
  int mode, nsub, obs_size;
  const real *actions;        // [N,nu]   (the C ABI's float* arrays, seen as `real`: see gptr() in ss_kernel.h)
  const real *task_rand;      // [N,4] or null
  const real *fall_actions;   // [N,3,nu] or null
  const uint8_t *mask;        // [N] or null
  unsigned long long *prof;   // optional stage-cycle accumulators (SS_PROFILE builds), else null
  const int32_t *order;       // optional [N] processing order of the envs (heavy first), or null
  int32_t *work_counter;      // device word, zeroed before each launch: persistent waves pull env ids from it
  real *obs, *reward;
  // fused autoreset (ss_step_autoreset): envs whose step ends an episode run the Default reset in the same launch;
  // obs2 receives the observation AFTER the (possible) reset for every env, task_rand2 feeds the reset's reset_task
  int fused_reset;
  real *obs2;
  const real *task_rand2;
  uint8_t *terminated, *truncated;
  real *out0, *out1, *out2;  // kinematics: xpos, xmat ; debug forward: M [N,nv,nv], bias [N,nv], qacc [N,nv]
  // per-env body shapes (ss_model_create_shapes): bodyc holds num_shapes consecutive blocks of [nb][kBodyC] body constants
  // followed by [nv] dof inverse weights (block stride shape_stride(h) floats), candc num_shapes consecutive tables;
  // st.shape_id [N] selects per env (null = single-shape model)
};

}  // namespace ss
