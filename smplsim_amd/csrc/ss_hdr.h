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
  int nb, nn, nv, nq, nu, ne, ncand, nlev, nblev, maxD, nblk, nbox, nslot, maxU;
  int levstart[20], blevstart[20];   // node / body level offsets (kernel arguments -> scalar loads)
  int itemA[20], itemB[20];          // per-level offsets into the packed work-item tables
  int accp[20], bsol[20];            // per-level offsets: tree-accumulation parents, backward-solve targets
  // shared-blob word offsets
  int o_dofc, o_chainnode, o_nbase, o_ndepth, o_levnodes, o_bparent, o_blevbodies, o_blk, o_itemA, o_itemB, o_fsrc, o_accp, o_children, o_bsol, o_bsrc, o_subsize, shared_words;   // o_bsrc: (kk*3*D + 3J) | n_k<<16 into the U buffer
  // per-env LDS float offsets
  int l_H, l_S, l_G, l_Dinv, l_R, l_r, l_Ic, l_K, l_V, l_Ab, l_Ad, l_Gb, l_q, l_v, l_a, l_tau, l_grad,
      l_delta, l_C, l_diag, l_misc, env_floats;
  float dt, grav, margin, mu, solimp[5], K, B;   // K, B of aref (from solref, dmax)
  float qpos0_root[3];
};

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
  uint8_t *terminated, *truncated;
  float *out0, *out1, *out2;  // kinematics: xpos, xmat ; debug forward: M entries [N,ne], bias [N,nv], qacc [N,nv]
};

}  // namespace ss
