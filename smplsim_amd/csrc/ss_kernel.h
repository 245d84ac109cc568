// ss_kernel.h — one wavefront steps one SMPL-humanoid environment.
//
// This replaces, for the whole env batch, what the reference does per process with
//   15 x ( StablePDController.control   reference smpl_sim/envs/controllers.py:116-190
//        + mujoco.mj_step               call site smpl_sim/envs/humanoid_env.py:450 )
//   + compute_proprioception / reward / reset flags   humanoid_env.py:388-403,455-469, tasks/*.py
// in ONE kernel launch.  It is a new design, not a translation of MuJoCo:
//   * all rigid-body quantities are 6-D spatial vectors in a world-aligned frame whose origin is
//     the root body (keeps |r| < 2 m so float32 cancellation in m r^2 terms stays ~1e-6);
//   * the joint-space matrices are never formed densely: H(i,j) = S_j . (Hc_body(i) S_i) on the
//     tree sparsity pattern, factored by a level-parallel 3x3-block L^T D L in LDS;
//   * MuJoCo's soft-constraint problem  min_a 1/2 (a-a_s)^T M (a-a_s) + sum_i s_i(J_i a - aref_i)
//     is solved by Newton's method with the contact Jacobian folded into per-body 6x6 matrices:
//     J^T D J = sum_b X_b^T K_b X_b, i.e. the Hessian M + J^T D J is assembled by the SAME composite
//     "inertia" pass as M (floor contacts touch one ancestor chain, so the sparsity is unchanged);
//   * lanes are bodies / dofs / matrix entries / contact candidates depending on the stage; data
//     crosses lanes through the env's LDS block only, separated by wave-level syncs.
//
// The source is written against a tiny "wave context" W (lane id, sync, reductions, LDS atomic add)
// so that the identical code is compiled by hipcc for gfx950 and by g++ against the 64-fiber
// wavefront emulator under tests/wave_emu (unit-test infrastructure; never a product path).
#pragma once
#include <math.h>
#include <stdint.h>

#include "ss_hdr.h"

#if defined(__HIPCC__)
#define SS_DEV __device__ __forceinline__
#else
#define SS_DEV inline
#endif

namespace ss {

struct Contact {
  float rx, ry, rz;      // contact point relative to the root origin
  float t1x, t1y;        // first tangent (unit, in the floor plane); second = (-t1y, t1x)
  float D;               // 1/R of the 4 pyramid rows
  float aref[4], jar[4], jd[4];
  int body, active;
};
struct Limit { float sign, D, aref, jar, jd; };

SS_DEV float bits2f(uint32_t u) { union { uint32_t u; float f; } c; c.u = u; return c.f; }
SS_DEV bool is_bad(float x) { return !(x <= 1e10f && x >= -1e10f); }

template <class W, int DOFP, int CANDP>
struct Sim {
  W *w;
  const KArgs *k;
  const uint32_t *T;      // shared tables in LDS
  int lane, env;
  // per-env LDS arrays
  float *H, *S, *G, *Dinv, *R, *r, *Ic, *Kc, *V, *Ab, *Ad, *Gb, *q, *v, *a, *tau, *grad, *delta, *C, *diag, *misc;
  // per-lane constants
  float bc[kBodyC];
  int bpar, bdep;
  float cc[CANDP][kCandC];
  int cb[CANDP];
  // per-lane state
  float Ib[10], fb[6], sv[6];
  Contact con[CANDP];
  Limit lim[DOFP];
  int iters, nwarn_add;
  unsigned long long touchmask;

  SS_DEV int ti(int off, int i) const { return (int)T[off + i]; }
  SS_DEV float tf(int off, int i) const { return bits2f(T[off + i]); }
  SS_DEV float dc(int dof, int f) const { return tf(k->h.o_dofc, dof * kDofC + f); }

  SS_DEV void init(W *w_, const KArgs *k_, const uint32_t *T_, float *L, int env_) {
    w = w_; k = k_; T = T_; lane = w->lane(); env = env_;
    const Hdr &h = k->h;
    H = L + h.l_H; S = L + h.l_S; G = L + h.l_G; Dinv = L + h.l_Dinv; R = L + h.l_R; r = L + h.l_r;
    Ic = L + h.l_Ic; Kc = L + h.l_K; V = L + h.l_V; Ab = L + h.l_Ab; Ad = L + h.l_Ad; Gb = L + h.l_Gb;
    q = L + h.l_q; v = L + h.l_v; a = L + h.l_a; tau = L + h.l_tau; grad = L + h.l_grad;
    delta = L + h.l_delta; C = L + h.l_C; diag = L + h.l_diag; misc = L + h.l_misc;
    bpar = -1; bdep = -1;
    for (int i = 0; i < kBodyC; i++) bc[i] = 0.f;
    if (lane < h.nb) {
      for (int i = 0; i < kBodyC; i++) bc[i] = k->bodyc[lane * kBodyC + i];
      bpar = ti(h.o_bparent, lane);
      bdep = ti(h.o_ndepth, lane + 1) - 1;
    }
    for (int p = 0; p < CANDP; p++) {
      int c = p * 64 + lane;
      cb[p] = -1;
      for (int i = 0; i < kCandC; i++) cc[p][i] = 0.f;
      if (c < h.ncand) {
        cb[p] = k->candb[c];
        for (int i = 0; i < kCandC; i++) cc[p][i] = k->candc[c * kCandC + i];
      }
      con[p].active = 0;
    }
    for (int p = 0; p < DOFP; p++) lim[p].sign = 0.f;
    for (int i = 0; i < 6; i++) sv[i] = 0.f;
    iters = 0; nwarn_add = 0; touchmask = 0ull;
  }

  // ------------------------------------------------------------------ HBM <-> LDS
  SS_DEV void load(float *dst, const float *src, int n) { for (int i = lane; i < n; i += 64) dst[i] = src[i]; }
  SS_DEV void store(float *dst, const float *src, int n) { for (int i = lane; i < n; i += 64) dst[i] = src[i]; }

  // ------------------------------------------------------------------ tree helpers
  template <int NC>
  SS_DEV void tree_accumulate(float *arr) {               // arr[b] += sum over descendants, in place
    const Hdr &h = k->h;
    for (int L = h.nblev - 1; L >= 1; --L) {
      int s = ti(h.o_blevstart, L), n = (ti(h.o_blevstart, L + 1) - s) * NC;
      for (int idx = lane; idx < n; idx += 64) {
        int bi = idx / NC, c = idx - bi * NC;
        int b = ti(h.o_blevbodies, s + bi), p = ti(h.o_bparent, b);
        w->atomic_add(&arr[p * NC + c], arr[b * NC + c]);
      }
      w->sync();
    }
  }

  // A[b] = sum over the dofs d on the chain of body b of S[d] * x[d]   (spatial accel without bias)
  SS_DEV void body_accel(const float *x, float *A) {
    const Hdr &h = k->h;
    for (int idx = lane; idx < 6 * h.nb; idx += 64) {
      int b = idx / 6, c = idx - 6 * b, n = b + 1;
      int dn = ti(h.o_ndepth, n);
      float s = 0.f;
      for (int kk = 0; kk <= dn; kk++) {
        int d = 3 * ti(h.o_chainnode, n * h.nlev + kk);
        s += S[6 * d + c] * x[d] + S[6 * d + 6 + c] * x[d + 1] + S[6 * d + 12 + c] * x[d + 2];
      }
      A[idx] = s;
    }
  }

  // ------------------------------------------------------------------ kinematics + velocities + inertia + bias
  // with_dyn = false: positions/orientations only (observation FK)
  SS_DEV void forward_kin(bool with_dyn) {
    const Hdr &h = k->h;
    float vb[6] = {0, 0, 0, 0, 0, 0}, ab[6] = {0, 0, 0, 0, 0, 0};
    float Rb[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, rb[3] = {0, 0, 0};
    if (lane == 0) {
      float qw = q[3], qx = q[4], qy = q[5], qz = q[6];
      float n = sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
      if (n < 1e-15f) { qw = 1; qx = qy = qz = 0; } else { float in = 1.f / n; qw *= in; qx *= in; qy *= in; qz *= in; }
      Rb[0] = 1 - 2 * (qy * qy + qz * qz); Rb[1] = 2 * (qx * qy - qw * qz); Rb[2] = 2 * (qx * qz + qw * qy);
      Rb[3] = 2 * (qx * qy + qw * qz); Rb[4] = 1 - 2 * (qx * qx + qz * qz); Rb[5] = 2 * (qy * qz - qw * qx);
      Rb[6] = 2 * (qx * qz - qw * qy); Rb[7] = 2 * (qy * qz + qw * qx); Rb[8] = 1 - 2 * (qx * qx + qy * qy);
      for (int i = 0; i < 9; i++) R[i] = Rb[i];
      r[0] = r[1] = r[2] = 0.f;
      for (int d = 0; d < 6; d++) for (int c = 0; c < 6; c++) S[6 * d + c] = 0.f;
      S[0 * 6 + 3] = 1.f; S[1 * 6 + 4] = 1.f; S[2 * 6 + 5] = 1.f;
      for (int d = 0; d < 3; d++) { S[6 * (3 + d) + 0] = Rb[d]; S[6 * (3 + d) + 1] = Rb[3 + d]; S[6 * (3 + d) + 2] = Rb[6 + d]; }
      if (with_dyn) {
        float wl0 = v[3], wl1 = v[4], wl2 = v[5];
        vb[0] = Rb[0] * wl0 + Rb[1] * wl1 + Rb[2] * wl2;
        vb[1] = Rb[3] * wl0 + Rb[4] * wl1 + Rb[5] * wl2;
        vb[2] = Rb[6] * wl0 + Rb[7] * wl1 + Rb[8] * wl2;
        vb[3] = v[0]; vb[4] = v[1]; vb[5] = v[2];
        // free joint: spatial bias acceleration (0 ; u x w)
        ab[3] = vb[4] * vb[2] - vb[5] * vb[1];
        ab[4] = vb[5] * vb[0] - vb[3] * vb[2];
        ab[5] = vb[3] * vb[1] - vb[4] * vb[0];
        for (int c = 0; c < 6; c++) { V[c] = vb[c]; Ad[c] = ab[c]; }
      }
    }
    w->sync();
    for (int L = 1; L < h.nblev; L++) {
      if (bdep == L) {
        const int b = lane, n = b + 1;
        float Rp[9], rp[3];
        for (int i = 0; i < 9; i++) Rp[i] = R[9 * bpar + i];
        for (int i = 0; i < 3; i++) rp[i] = r[3 * bpar + i];
        for (int i = 0; i < 3; i++) rb[i] = rp[i] + Rp[3 * i] * bc[0] + Rp[3 * i + 1] * bc[1] + Rp[3 * i + 2] * bc[2];
        float sx, cx, sy, cy, sz, cz;
        sincosf(q[3 * b + 4], &sx, &cx); sincosf(q[3 * b + 5], &sy, &cy); sincosf(q[3 * b + 6], &sz, &cz);
        float ax[3], ay[3], az[3], c0[3], c1[3], c2[3];
        for (int i = 0; i < 3; i++) {
          float p0 = Rp[3 * i], p1 = Rp[3 * i + 1], p2 = Rp[3 * i + 2];
          ax[i] = p0;
          float r11 = cx * p1 + sx * p2, r12 = -sx * p1 + cx * p2;       // R1 = Rp Rx (col0 = p0)
          ay[i] = r11;
          float r20 = cy * p0 - sy * r12, r22 = sy * p0 + cy * r12;       // R2 = R1 Ry (col1 = r11)
          az[i] = r22;
          c0[i] = cz * r20 + sz * r11; c1[i] = -sz * r20 + cz * r11; c2[i] = r22;   // R3 = R2 Rz
        }
        for (int i = 0; i < 3; i++) { Rb[3 * i] = c0[i]; Rb[3 * i + 1] = c1[i]; Rb[3 * i + 2] = c2[i]; }
        for (int i = 0; i < 9; i++) R[9 * b + i] = Rb[i];
        for (int i = 0; i < 3; i++) r[3 * b + i] = rb[i];
        float sd[3][6];
        const float *axs[3] = {ax, ay, az};
        for (int j = 0; j < 3; j++) {
          const float *A_ = axs[j];
          sd[j][0] = A_[0]; sd[j][1] = A_[1]; sd[j][2] = A_[2];
          sd[j][3] = rb[1] * A_[2] - rb[2] * A_[1];
          sd[j][4] = rb[2] * A_[0] - rb[0] * A_[2];
          sd[j][5] = rb[0] * A_[1] - rb[1] * A_[0];
          for (int c = 0; c < 6; c++) S[6 * (3 * n + j) + c] = sd[j][c];
        }
        if (with_dyn) {
          for (int c = 0; c < 6; c++) { vb[c] = V[6 * bpar + c]; ab[c] = Ad[6 * bpar + c]; }
          for (int j = 0; j < 3; j++) {
            float qd = v[3 * n + j];
            const float *s = sd[j];
            // (w;u) x_m (sw;su) = (w x sw ; w x su + u x sw)
            float c0_ = vb[1] * s[2] - vb[2] * s[1], c1_ = vb[2] * s[0] - vb[0] * s[2], c2_ = vb[0] * s[1] - vb[1] * s[0];
            float d0 = vb[1] * s[5] - vb[2] * s[4] + vb[4] * s[2] - vb[5] * s[1];
            float d1 = vb[2] * s[3] - vb[0] * s[5] + vb[5] * s[0] - vb[3] * s[2];
            float d2 = vb[0] * s[4] - vb[1] * s[3] + vb[3] * s[1] - vb[4] * s[0];
            ab[0] += c0_ * qd; ab[1] += c1_ * qd; ab[2] += c2_ * qd; ab[3] += d0 * qd; ab[4] += d1 * qd; ab[5] += d2 * qd;
            for (int c = 0; c < 6; c++) vb[c] += s[c] * qd;
          }
          for (int c = 0; c < 6; c++) { V[6 * b + c] = vb[c]; Ad[6 * b + c] = ab[c]; }
        }
      }
      w->sync();
    }
    if (!with_dyn) return;
    // ---- body spatial inertia about the root origin (world axes), bias force, sensor velocities
    if (lane < h.nb) {
      const int b = lane;
      float cx_ = rb[0] + Rb[0] * bc[3] + Rb[1] * bc[4] + Rb[2] * bc[5];
      float cy_ = rb[1] + Rb[3] * bc[3] + Rb[4] * bc[4] + Rb[5] * bc[5];
      float cz_ = rb[2] + Rb[6] * bc[3] + Rb[7] * bc[4] + Rb[8] * bc[5];
      float m = bc[6];
      // Ibar = Rb Ibody Rb^T
      float Bm[9] = {bc[7], bc[8], bc[9], bc[8], bc[10], bc[11], bc[9], bc[11], bc[12]};
      float RB[9];
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
        RB[3 * i + j] = Rb[3 * i] * Bm[j] + Rb[3 * i + 1] * Bm[3 + j] + Rb[3 * i + 2] * Bm[6 + j];
      float Ixx = RB[0] * Rb[0] + RB[1] * Rb[1] + RB[2] * Rb[2];
      float Ixy = RB[0] * Rb[3] + RB[1] * Rb[4] + RB[2] * Rb[5];
      float Ixz = RB[0] * Rb[6] + RB[1] * Rb[7] + RB[2] * Rb[8];
      float Iyy = RB[3] * Rb[3] + RB[4] * Rb[4] + RB[5] * Rb[5];
      float Iyz = RB[3] * Rb[6] + RB[4] * Rb[7] + RB[5] * Rb[8];
      float Izz = RB[6] * Rb[6] + RB[7] * Rb[7] + RB[8] * Rb[8];
      Ib[0] = m; Ib[1] = m * cx_; Ib[2] = m * cy_; Ib[3] = m * cz_;
      Ib[4] = Ixx + m * (cy_ * cy_ + cz_ * cz_); Ib[5] = Ixy - m * cx_ * cy_; Ib[6] = Ixz - m * cx_ * cz_;
      Ib[7] = Iyy + m * (cx_ * cx_ + cz_ * cz_); Ib[8] = Iyz - m * cy_ * cz_; Ib[9] = Izz + m * (cx_ * cx_ + cy_ * cy_);
      for (int i = 0; i < 10; i++) Ic[10 * b + i] = Ib[i];
      // f = I (a - a_grav) + v x* (I v)
      float ag[6] = {ab[0], ab[1], ab[2], ab[3], ab[4], ab[5] - h.grav};
      float Ia[6], Iv[6];
      imul(Ib, ag, Ia); imul(Ib, vb, Iv);
      fb[0] = Ia[0] + vb[1] * Iv[2] - vb[2] * Iv[1] + vb[4] * Iv[5] - vb[5] * Iv[4];
      fb[1] = Ia[1] + vb[2] * Iv[0] - vb[0] * Iv[2] + vb[5] * Iv[3] - vb[3] * Iv[5];
      fb[2] = Ia[2] + vb[0] * Iv[1] - vb[1] * Iv[0] + vb[3] * Iv[4] - vb[4] * Iv[3];
      fb[3] = Ia[3] + vb[1] * Iv[5] - vb[2] * Iv[4];
      fb[4] = Ia[4] + vb[2] * Iv[3] - vb[0] * Iv[5];
      fb[5] = Ia[5] + vb[0] * Iv[4] - vb[1] * Iv[3];
      for (int c = 0; c < 6; c++) Gb[6 * b + c] = fb[c];
      // framelinvel / frameangvel of the body frame origin
      sv[0] = vb[3] + vb[1] * rb[2] - vb[2] * rb[1];
      sv[1] = vb[4] + vb[2] * rb[0] - vb[0] * rb[2];
      sv[2] = vb[5] + vb[0] * rb[1] - vb[1] * rb[0];
      sv[3] = vb[0]; sv[4] = vb[1]; sv[5] = vb[2];
    }
    w->sync();
    // composite inertia and bias force C = S^T subtree(f): one fused level sweep (16 comps per body)
    for (int L = h.nblev - 1; L >= 1; --L) {
      int s = ti(h.o_blevstart, L), n = (ti(h.o_blevstart, L + 1) - s) * 16;
      for (int idx = lane; idx < n; idx += 64) {
        int bi = idx >> 4, c = idx & 15;
        int b = ti(h.o_blevbodies, s + bi), p = ti(h.o_bparent, b);
        if (c < 10) w->atomic_add(&Ic[10 * p + c], Ic[10 * b + c]);
        else w->atomic_add(&Gb[6 * p + c - 10], Gb[6 * b + c - 10]);
      }
      w->sync();
    }
    for (int p = 0; p < DOFP; p++) {
      int i = p * 64 + lane;
      if (i < h.nv) {
        int n = i / 3, b = n > 0 ? n - 1 : 0;
        float s = 0.f;
        for (int c = 0; c < 6; c++) s += S[6 * i + c] * Gb[6 * b + c];
        C[i] = s;
      }
    }
    w->sync();
  }

  // spatial inertia (10 params, about the origin) times motion vector (w;u) -> force (n;f)
  SS_DEV static void imul(const float *I, const float *x, float *y) {
    float m = I[0], cx = I[1], cy = I[2], cz = I[3];
    y[0] = I[4] * x[0] + I[5] * x[1] + I[6] * x[2] + cy * x[5] - cz * x[4];
    y[1] = I[5] * x[0] + I[7] * x[1] + I[8] * x[2] + cz * x[3] - cx * x[5];
    y[2] = I[6] * x[0] + I[8] * x[1] + I[9] * x[2] + cx * x[4] - cy * x[3];
    y[3] = m * x[3] - (cy * x[2] - cz * x[1]);
    y[4] = m * x[4] - (cz * x[0] - cx * x[2]);
    y[5] = m * x[5] - (cx * x[1] - cy * x[0]);
  }

  // ------------------------------------------------------------------ MuJoCo impedance d(r)
  SS_DEV float impedance(float pos, float margin) const {
    const float *si = k->h.solimp;
    float x = (pos - margin) / si[2];
    if (x < 0) x = -x;
    if (x >= 1.f) return si[1];
    if (x <= 0.f) return si[0];
    float y;
    if (si[4] == 1.f) y = x;
    else if (x <= si[3]) y = powf(x, si[4]) / powf(si[3], si[4] - 1.f);
    else y = 1.f - powf(1.f - x, si[4]) / powf(1.f - si[3], si[4] - 1.f);
    return si[0] + y * (si[1] - si[0]);
  }

  // ------------------------------------------------------------------ floor contacts + joint limits
  SS_DEV void make_constraints() {
    const Hdr &h = k->h;
    const float pz = q[2], mu = h.mu;
    touchmask = 0ull;
    for (int p = 0; p < CANDP; p++) {
      Contact &c = con[p];
      c.active = 0;
      int qual = 0;
      float dist = 0.f, px = 0, py = 0, pzr = 0;
      const bool valid = cb[p] >= 0;
      const int b = valid ? (cb[p] & 255) : 0;
      const bool caps = valid && (cb[p] & 256);
      if (valid) {
        const float *Rb = R + 9 * b, *rb = r + 3 * b, *cv = cc[p];
        if (!caps) {
          float ldist = Rb[6] * cv[0] + Rb[7] * cv[1] + Rb[8] * cv[2];
          float dcen = pz + rb[2] + Rb[6] * cv[3] + Rb[7] * cv[4] + Rb[8] * cv[5];
          qual = !(dcen + ldist > h.margin || ldist > 0.f);
          dist = dcen + ldist;
          float lx = cv[0] + cv[3], ly = cv[1] + cv[4], lz = cv[2] + cv[5];
          px = rb[0] + Rb[0] * lx + Rb[1] * ly + Rb[2] * lz;
          py = rb[1] + Rb[3] * lx + Rb[4] * ly + Rb[5] * lz;
          pzr = rb[2] + Rb[6] * lx + Rb[7] * ly + Rb[8] * lz - 0.5f * dist;
          c.t1x = 0.f; c.t1y = 1.f;
        } else {
          float czw = pz + rb[2] + Rb[6] * cv[0] + Rb[7] * cv[1] + Rb[8] * cv[2];
          dist = czw - cv[6];
          qual = !(dist > h.margin);
          px = rb[0] + Rb[0] * cv[0] + Rb[1] * cv[1] + Rb[2] * cv[2];
          py = rb[1] + Rb[3] * cv[0] + Rb[4] * cv[1] + Rb[5] * cv[2];
          pzr = rb[2] + Rb[6] * cv[0] + Rb[7] * cv[1] + Rb[8] * cv[2] - (cv[6] + 0.5f * dist);
          float axx = Rb[0] * cv[3] + Rb[1] * cv[4] + Rb[2] * cv[5];
          float axy = Rb[3] * cv[3] + Rb[4] * cv[4] + Rb[5] * cv[5];
          float nn = sqrtf(axx * axx + axy * axy);
          if (nn < 1e-15f) { c.t1x = 1.f; c.t1y = 0.f; } else { c.t1x = axx / nn; c.t1y = axy / nn; }
        }
      }
      unsigned long long bal = w->ballot(qual && !caps);
      int act = qual;
      if (valid && !caps) {                                   // plane-box: first 4 qualifying corners in index order
        int grp = lane & ~7;
        unsigned bits = (unsigned)((bal >> grp) & 0xFFull) & ((1u << (lane & 7)) - 1u);
        int rank = 0;
        for (unsigned t = bits; t; t &= t - 1) rank++;
        act = qual && rank < 4;
      }
      if (act) {
        c.active = 1; c.body = b; c.rx = px; c.ry = py; c.rz = pzr;
        const float *vb = V + 6 * b;
        float vx = vb[3] + vb[1] * pzr - vb[2] * py;
        float vy = vb[4] + vb[2] * px - vb[0] * pzr;
        float vz = vb[5] + vb[0] * py - vb[1] * px;
        float vt1 = c.t1x * vx + c.t1y * vy, vt2 = -c.t1y * vx + c.t1x * vy;
        float imp = impedance(dist, h.margin);
        float R0 = (1.f - imp) / imp * cc[p][7] * (1.f + mu * mu);
        if (R0 < 1e-15f) R0 = 1e-15f;
        float Rpy = 2.f * mu * mu * R0;
        c.D = 1.f / Rpy;
        float kterm = h.K * imp * (dist - h.margin);
        c.aref[0] = -h.B * (vz + mu * vt1) - kterm;
        c.aref[1] = -h.B * (vz - mu * vt1) - kterm;
        c.aref[2] = -h.B * (vz + mu * vt2) - kterm;
        c.aref[3] = -h.B * (vz - mu * vt2) - kterm;
      }
      touchmask |= w->bor(act ? (1ull << b) : 0ull);         // wave-wide OR: bodies touching the floor
    }
    for (int p = 0; p < DOFP; p++) {
      int i = p * 64 + lane;
      Limit &l = lim[p];
      l.sign = 0.f; l.D = 0.f; l.aref = 0.f; l.jar = 0.f; l.jd = 0.f;
      if (i >= 6 && i < h.nv && dc(i, 3) != 0.f) {
        float qi = q[i + 1], lo = dc(i, 1), hi = dc(i, 2), pos = 0.f;
        if (qi - lo < 0.f) { l.sign = 1.f; pos = qi - lo; }
        else if (hi - qi < 0.f) { l.sign = -1.f; pos = hi - qi; }
        if (l.sign != 0.f) {
          float imp = impedance(pos, 0.f);
          float Rr = (1.f - imp) / imp * dc(i, 4);
          if (Rr < 1e-15f) Rr = 1e-15f;
          l.D = 1.f / Rr;
          l.aref = -h.B * (l.sign * v[i]) - h.K * imp * pos;
        }
      }
    }
  }

  // rows: jar (from A = Ab, x = a) or jd (from A = Ad, x = delta)
  SS_DEV void eval_rows(const float *A, const float *x, bool is_delta) {
    const float mu = k->h.mu;
    for (int p = 0; p < CANDP; p++) {
      Contact &c = con[p];
      if (!c.active) continue;
      const float *Ab_ = A + 6 * c.body;
      float ax = Ab_[3] + Ab_[1] * c.rz - Ab_[2] * c.ry;
      float ay = Ab_[4] + Ab_[2] * c.rx - Ab_[0] * c.rz;
      float az = Ab_[5] + Ab_[0] * c.ry - Ab_[1] * c.rx;
      float t1 = mu * (c.t1x * ax + c.t1y * ay), t2 = mu * (-c.t1y * ax + c.t1x * ay);
      if (is_delta) { c.jd[0] = az + t1; c.jd[1] = az - t1; c.jd[2] = az + t2; c.jd[3] = az - t2; }
      else { c.jar[0] = az + t1 - c.aref[0]; c.jar[1] = az - t1 - c.aref[1]; c.jar[2] = az + t2 - c.aref[2]; c.jar[3] = az - t2 - c.aref[3]; }
    }
    for (int p = 0; p < DOFP; p++) {
      int i = p * 64 + lane;
      Limit &l = lim[p];
      if (l.sign == 0.f) continue;
      if (is_delta) l.jd = l.sign * x[i]; else l.jar = l.sign * x[i] - l.aref;
    }
  }

  // ------------------------------------------------------------------ H assembly: H(i,j) = S_j . (Hc_body(i) S_i) + diag
  // Hc = expand(Ic) + Kc (Kc must hold subtree sums, or zeros)
  SS_DEV void assemble_H() {
    const Hdr &h = k->h;
    for (int p = 0; p < DOFP; p++) {
      int i = p * 64 + lane;
      if (i < h.nv) {
        int n = i / 3, b = n > 0 ? n - 1 : 0;
        const float *I = Ic + 10 * b, *Km = Kc + 21 * b, *s = S + 6 * i;
        float g[6];
        imul(I, s, g);
        const int off[6] = {0, 6, 11, 15, 18, 20};
        for (int a_ = 0; a_ < 6; a_++) {
          float acc = 0.f;
          for (int c = 0; c < 6; c++) {
            int lo = a_ < c ? a_ : c, hi = a_ < c ? c : a_;
            acc += Km[off[lo] + (hi - lo)] * s[c];
          }
          g[a_] += acc;
        }
        for (int c = 0; c < 6; c++) G[6 * i + c] = g[c];
      }
    }
    w->sync();
    for (int e = lane; e < h.ne; e += 64) {
      int code = ti(h.o_decode, e), i = code >> 16, j = code & 0xFFFF;
      const float *sj = S + 6 * j, *gi = G + 6 * i;
      float acc = sj[0] * gi[0] + sj[1] * gi[1] + sj[2] * gi[2] + sj[3] * gi[3] + sj[4] * gi[4] + sj[5] * gi[5];
      if (i == j) acc += diag[i];
      H[e] = acc;
    }
    w->sync();
  }

  // ------------------------------------------------------------------ level-parallel 3x3-block L^T D L
  SS_DEV void factor_H() {
    const Hdr &h = k->h;
    float *U = G;                                            // G is free between assembly and the next one
    for (int L = h.nlev - 1; L >= 0; --L) {
      const int s = ti(h.o_levstart, L), nk = ti(h.o_levstart, L + 1) - s, D = 3 * L, Wd = D + 3;
      const int cols = D > 0 ? D : 1;
      // phase 1: Dinv_k and U = Dinv_k P
      for (int idx = lane; idx < nk * cols; idx += 64) {
        int kk = idx / cols, j = idx - kk * cols;
        int n = ti(h.o_levnodes, s + kk), base = ti(h.o_nbase, n);
        float d00 = H[base + D], d10 = H[base + Wd + D], d11 = H[base + Wd + D + 1];
        float d20 = H[base + 2 * Wd + D], d21 = H[base + 2 * Wd + D + 1], d22 = H[base + 2 * Wd + D + 2];
        float c00 = d11 * d22 - d21 * d21, c01 = d21 * d20 - d10 * d22, c02 = d10 * d21 - d11 * d20;
        float det = d00 * c00 + d10 * c01 + d20 * c02;
        float id = 1.f / det;
        float i00 = c00 * id, i01 = c01 * id, i02 = c02 * id;
        float i11 = (d00 * d22 - d20 * d20) * id, i12 = (d10 * d20 - d00 * d21) * id, i22 = (d00 * d11 - d10 * d10) * id;
        if (j == 0) { float *o = Dinv + 6 * n; o[0] = i00; o[1] = i01; o[2] = i02; o[3] = i11; o[4] = i12; o[5] = i22; }
        if (j < D) {
          float p0 = H[base + j], p1 = H[base + Wd + j], p2 = H[base + 2 * Wd + j];
          U[(kk * 3 + 0) * D + j] = i00 * p0 + i01 * p1 + i02 * p2;
          U[(kk * 3 + 1) * D + j] = i01 * p0 + i11 * p1 + i12 * p2;
          U[(kk * 3 + 2) * D + j] = i02 * p0 + i12 * p1 + i22 * p2;
        }
      }
      if (L == 0) { w->sync(); break; }
      w->sync();
      // phase 2: ancestor block -= P^T U   (lower triangle i >= j over the chain positions)
      const int Tn = D * (D + 1) / 2;
      for (int idx = lane; idx < nk * Tn; idx += 64) {
        int kk = idx / Tn, t = idx - kk * Tn;
        int i = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
        while ((i + 1) * (i + 2) / 2 <= t) i++;
        while (i * (i + 1) / 2 > t) i--;
        int j = t - i * (i + 1) / 2;
        int n = ti(h.o_levnodes, s + kk), base = ti(h.o_nbase, n);
        float val = H[base + i] * U[(kk * 3) * D + j] + H[base + Wd + i] * U[(kk * 3 + 1) * D + j] + H[base + 2 * Wd + i] * U[(kk * 3 + 2) * D + j];
        int dst = ti(h.o_chainrow, n * h.maxD + i) + j;
        w->atomic_add(&H[dst], -val);
      }
      w->sync();
      // phase 3: overwrite P by U (the L factor rows)
      for (int idx = lane; idx < nk * D; idx += 64) {
        int kk = idx / D, j = idx - kk * D;
        int n = ti(h.o_levnodes, s + kk), base = ti(h.o_nbase, n);
        H[base + j] = U[(kk * 3) * D + j]; H[base + Wd + j] = U[(kk * 3 + 1) * D + j]; H[base + 2 * Wd + j] = U[(kk * 3 + 2) * D + j];
      }
      w->sync();
    }
  }

  // solve H x = b in place (x holds b on entry)
  SS_DEV void solve_H(float *x) {
    const Hdr &h = k->h;
    for (int L = h.nlev - 1; L >= 1; --L) {                  // x <- L^-T x (leaves to root)
      const int s = ti(h.o_levstart, L), nk = ti(h.o_levstart, L + 1) - s, D = 3 * L, Wd = D + 3;
      for (int idx = lane; idx < nk * D; idx += 64) {
        int kk = idx / D, j = idx - kk * D;
        int n = ti(h.o_levnodes, s + kk), base = ti(h.o_nbase, n);
        float val = H[base + j] * x[3 * n] + H[base + Wd + j] * x[3 * n + 1] + H[base + 2 * Wd + j] * x[3 * n + 2];
        int dj = 3 * ti(h.o_chainnode, n * h.nlev + j / 3) + j % 3;
        w->atomic_add(&x[dj], -val);
      }
      w->sync();
    }
    if (lane < h.nn) {                                        // x <- D^-1 x
      const float *o = Dinv + 6 * lane;
      float x0 = x[3 * lane], x1 = x[3 * lane + 1], x2 = x[3 * lane + 2];
      x[3 * lane] = o[0] * x0 + o[1] * x1 + o[2] * x2;
      x[3 * lane + 1] = o[1] * x0 + o[3] * x1 + o[4] * x2;
      x[3 * lane + 2] = o[2] * x0 + o[4] * x1 + o[5] * x2;
    }
    w->sync();
    for (int L = 1; L < h.nlev; L++) {                        // x <- L^-1 x (root to leaves)
      const int s = ti(h.o_levstart, L), nk = ti(h.o_levstart, L + 1) - s, D = 3 * L, Wd = D + 3;
      for (int idx = lane; idx < nk * D; idx += 64) {
        int kk = idx / D, j = idx - kk * D;
        int n = ti(h.o_levnodes, s + kk), base = ti(h.o_nbase, n);
        float xj = x[3 * ti(h.o_chainnode, n * h.nlev + j / 3) + j % 3];
        w->atomic_add(&x[3 * n], -H[base + j] * xj);
        w->atomic_add(&x[3 * n + 1], -H[base + Wd + j] * xj);
        w->atomic_add(&x[3 * n + 2], -H[base + 2 * Wd + j] * xj);
      }
      w->sync();
    }
  }

  // ------------------------------------------------------------------ Newton solve of the constrained acceleration
  SS_DEV void ls_eval(float al, float c1, float c2, float &d1, float &d2) {
    float s1 = 0.f, s2 = 0.f;
    for (int p = 0; p < CANDP; p++) {
      const Contact &c = con[p];
      if (!c.active) continue;
      for (int r_ = 0; r_ < 4; r_++) {
        float x = c.jar[r_] + al * c.jd[r_];
        if (x < 0.f) { s1 += c.D * x * c.jd[r_]; s2 += c.D * c.jd[r_] * c.jd[r_]; }
      }
    }
    for (int p = 0; p < DOFP; p++) {
      const Limit &l = lim[p];
      if (l.sign == 0.f) continue;
      float x = l.jar + al * l.jd;
      if (x < 0.f) { s1 += l.D * x * l.jd; s2 += l.D * l.jd * l.jd; }
    }
    d1 = c1 + al * c2 + w->sum(s1);
    d2 = c2 + w->sum(s2);
  }

  SS_DEV void newton() {
    const Hdr &h = k->h;
    const float mu = h.mu;
    const int maxit = k->cfg.newton_iters > 0 ? k->cfg.newton_iters : 8;
    body_accel(a, Ab);
    w->sync();
    eval_rows(Ab, a, false);
    for (int it = 0; it < maxit; it++) {
      iters++;
      // ---- Gb = I_b A_b ; zero Kc
      if (lane < h.nb) {
        float Ia[6];
        imul(Ib, Ab + 6 * lane, Ia);
        for (int c = 0; c < 6; c++) Gb[6 * lane + c] = Ia[c];
        for (int c = 0; c < 21; c++) Kc[21 * lane + c] = 0.f;
      }
      w->sync();
      // ---- contact forces and K_b = sum_rows D u u^T, u = (rho x w ; w)
      int nact = 0;
      for (int p = 0; p < CANDP; p++) {
        const Contact &c = con[p];
        if (!c.active) continue;
        float f[4], fx = 0, fy = 0, fz = 0;
        float Wxx = 0, Wxy = 0, Wxz = 0, Wyy = 0, Wyz = 0, Wzz = 0;
        const float wx[4] = {mu * c.t1x, -mu * c.t1x, -mu * c.t1y, mu * c.t1y};
        const float wy[4] = {mu * c.t1y, -mu * c.t1y, mu * c.t1x, -mu * c.t1x};
        for (int r_ = 0; r_ < 4; r_++) {
          f[r_] = c.jar[r_] < 0.f ? -c.D * c.jar[r_] : 0.f;
          if (c.jar[r_] < 0.f) {
            nact++;
            fx += f[r_] * wx[r_]; fy += f[r_] * wy[r_]; fz += f[r_];
            Wxx += c.D * wx[r_] * wx[r_]; Wxy += c.D * wx[r_] * wy[r_]; Wxz += c.D * wx[r_];
            Wyy += c.D * wy[r_] * wy[r_]; Wyz += c.D * wy[r_]; Wzz += c.D;
          }
        }
        if (Wzz == 0.f) continue;
        float *g = Gb + 6 * c.body;
        w->atomic_add(&g[0], -(c.ry * fz - c.rz * fy));
        w->atomic_add(&g[1], -(c.rz * fx - c.rx * fz));
        w->atomic_add(&g[2], -(c.rx * fy - c.ry * fx));
        w->atomic_add(&g[3], -fx); w->atomic_add(&g[4], -fy); w->atomic_add(&g[5], -fz);
        // Y = P W (columns rho x W[:,j]), Z = rows rho x Y[i,:]
        const float Wm[9] = {Wxx, Wxy, Wxz, Wxy, Wyy, Wyz, Wxz, Wyz, Wzz};
        float Y[9], Z[9];
        for (int j = 0; j < 3; j++) {
          float c0 = Wm[j], c1 = Wm[3 + j], c2 = Wm[6 + j];
          Y[j] = c.ry * c2 - c.rz * c1; Y[3 + j] = c.rz * c0 - c.rx * c2; Y[6 + j] = c.rx * c1 - c.ry * c0;
        }
        for (int i = 0; i < 3; i++) {
          float y0 = Y[3 * i], y1 = Y[3 * i + 1], y2 = Y[3 * i + 2];
          Z[3 * i] = c.ry * y2 - c.rz * y1; Z[3 * i + 1] = c.rz * y0 - c.rx * y2; Z[3 * i + 2] = c.rx * y1 - c.ry * y0;
        }
        float *Kb = Kc + 21 * c.body;
        w->atomic_add(&Kb[0], Z[0]); w->atomic_add(&Kb[1], Z[1]); w->atomic_add(&Kb[2], Z[2]);
        w->atomic_add(&Kb[3], Y[0]); w->atomic_add(&Kb[4], Y[1]); w->atomic_add(&Kb[5], Y[2]);
        w->atomic_add(&Kb[6], Z[4]); w->atomic_add(&Kb[7], Z[5]);
        w->atomic_add(&Kb[8], Y[3]); w->atomic_add(&Kb[9], Y[4]); w->atomic_add(&Kb[10], Y[5]);
        w->atomic_add(&Kb[11], Z[8]);
        w->atomic_add(&Kb[12], Y[6]); w->atomic_add(&Kb[13], Y[7]); w->atomic_add(&Kb[14], Y[8]);
        w->atomic_add(&Kb[15], Wxx); w->atomic_add(&Kb[16], Wxy); w->atomic_add(&Kb[17], Wxz);
        w->atomic_add(&Kb[18], Wyy); w->atomic_add(&Kb[19], Wyz); w->atomic_add(&Kb[20], Wzz);
      }
      w->sync();
      const bool any_contact_row = w->any(nact > 0);
      // ---- subtree sums of Gb (6) and Kc (21)
      for (int L = h.nblev - 1; L >= 1; --L) {
        int s = ti(h.o_blevstart, L), nbod = ti(h.o_blevstart, L + 1) - s;
        int ncomp = any_contact_row ? 27 : 6, n = nbod * ncomp;
        for (int idx = lane; idx < n; idx += 64) {
          int bi = idx / ncomp, c = idx - bi * ncomp;
          int b = ti(h.o_blevbodies, s + bi), p = ti(h.o_bparent, b);
          if (c < 6) w->atomic_add(&Gb[6 * p + c], Gb[6 * b + c]);
          else w->atomic_add(&Kc[21 * p + c - 6], Kc[21 * b + c - 6]);
        }
        w->sync();
      }
      // ---- gradient and diagonal terms
      float gg = 0.f;
      for (int p = 0; p < DOFP; p++) {
        int i = p * 64 + lane;
        if (i < h.nv) {
          int n = i / 3, b = n > 0 ? n - 1 : 0;
          float s = C[i] + dc(i, 0) * a[i] - tau[i];
          for (int c = 0; c < 6; c++) s += S[6 * i + c] * Gb[6 * b + c];
          float dg = dc(i, 0);
          const Limit &l = lim[p];
          if (l.sign != 0.f && l.jar < 0.f) { s += l.sign * l.D * l.jar; dg += l.D; }
          grad[i] = s; diag[i] = dg; delta[i] = -s;
          gg += s * s;
        }
      }
      (void)gg;
      w->sync();
      assemble_H();
      factor_H();
      solve_H(delta);
      // ---- line search along delta
      body_accel(delta, Ad);
      w->sync();
      eval_rows(Ad, delta, true);
      float dg_ = 0.f, s_a = 0.f, s_b = 0.f;
      for (int p = 0; p < DOFP; p++) { int i = p * 64 + lane; if (i < h.nv) dg_ += delta[i] * grad[i]; }
      for (int p = 0; p < CANDP; p++) {
        const Contact &c = con[p];
        if (!c.active) continue;
        for (int r_ = 0; r_ < 4; r_++) if (c.jar[r_] < 0.f) { s_a += c.D * c.jar[r_] * c.jd[r_]; s_b += c.D * c.jd[r_] * c.jd[r_]; }
      }
      for (int p = 0; p < DOFP; p++) {
        const Limit &l = lim[p];
        if (l.sign != 0.f && l.jar < 0.f) { s_a += l.D * l.jar * l.jd; s_b += l.D * l.jd * l.jd; }
      }
      dg_ = w->sum(dg_); s_a = w->sum(s_a); s_b = w->sum(s_b);
      const float c1 = dg_ - s_a, c2 = -dg_ - s_b;          // phi'(al) = c1 + al c2 + sum_active(al) D (jar + al jd) jd
      float al = 1.f, d1, d2;
      ls_eval(1.f, c1, c2, d1, d2);
      // accept when |phi'| is 1e-3 of phi'(0) = delta.grad, or at the rounding level of its terms
      const float tol = 1e-3f * fabsf(dg_) + 2e-6f * (fabsf(dg_) + fabsf(s_a) + fabsf(s_b));
      bool exact = true;
      if (!(fabsf(d1) <= tol)) {
        exact = false;
        float lo = 0.f, hi = 1.f;
        if (d1 < 0.f) {                                     // minimum beyond 1: expand
          for (int e_ = 0; e_ < 6 && d1 < 0.f; e_++) { lo = hi; hi *= 2.f; ls_eval(hi, c1, c2, d1, d2); }
          al = hi;
        }
        for (int ls = 0; ls < 10; ls++) {
          if (fabsf(d1) <= tol) break;
          if (d1 < 0.f) lo = al; else hi = al;
          float nx = d2 > 0.f ? al - d1 / d2 : 0.5f * (lo + hi);
          if (!(nx > lo && nx < hi)) nx = 0.5f * (lo + hi);
          al = nx;
          ls_eval(al, c1, c2, d1, d2);
        }
      }
      // ---- take the step; detect active-set changes
      int changed = 0;
      for (int p = 0; p < DOFP; p++) { int i = p * 64 + lane; if (i < h.nv) a[i] += al * delta[i]; }
      for (int idx = lane; idx < 6 * h.nb; idx += 64) Ab[idx] += al * Ad[idx];
      for (int p = 0; p < CANDP; p++) {
        Contact &c = con[p];
        if (!c.active) continue;
        for (int r_ = 0; r_ < 4; r_++) {
          float nj = c.jar[r_] + al * c.jd[r_];
          changed |= (nj < 0.f) != (c.jar[r_] < 0.f);
          c.jar[r_] = nj;
        }
      }
      for (int p = 0; p < DOFP; p++) {
        Limit &l = lim[p];
        if (l.sign == 0.f) continue;
        float nj = l.jar + al * l.jd;
        changed |= (nj < 0.f) != (l.jar < 0.f);
        l.jar = nj;
      }
      w->sync();
      if (!w->any(changed) && exact) break;
    }
  }

  // ------------------------------------------------------------------ controllers (torque for the NEXT mj_step)
  // uses M, C of the forward pass that is in LDS (the "stale" qM / qfrc_bias) with the current q, v
  SS_DEV void controller(const float *action, float abias) {
    const Hdr &h = k->h;
    const int mode = k->cfg.control_mode;
    if (mode != SS_CTRL_UHC_PD) {
      for (int p = 0; p < DOFP; p++) {
        int i = p * 64 + lane;
        if (i < h.nv) {
          float t = 0.f;
          if (dc(i, 10) != 0.f) {
            float act = action[(int)dc(i, 11)] + abias, lim_ = dc(i, 7);
            if (mode == SS_CTRL_PD) t = -dc(i, 5) * (q[i + 1] - (act * dc(i, 8) + dc(i, 9))) - dc(i, 6) * v[i];
            else t = act * k->cfg.power_scale * lim_;
            t = fminf(fmaxf(t, -lim_), lim_);
          }
          tau[i] = t;
        }
      }
      w->sync();
      return;
    }
    float perr[DOFP];
    if (lane < h.nb) for (int c = 0; c < 21; c++) Kc[21 * lane + c] = 0.f;
    for (int p = 0; p < DOFP; p++) {
      int i = p * 64 + lane;
      perr[p] = 0.f;
      if (i < h.nv) {
        float kp = dc(i, 5), kd = dc(i, 6);
        if (dc(i, 10) != 0.f) perr[p] = q[i + 1] + v[i] * h.dt - ((action[(int)dc(i, 11)] + abias) * dc(i, 8) + dc(i, 9));
        diag[i] = dc(i, 0) + kd * h.dt;
        delta[i] = -C[i] - kp * perr[p] - kd * v[i];
      }
    }
    w->sync();
    assemble_H();
    factor_H();
    solve_H(delta);
    for (int p = 0; p < DOFP; p++) {
      int i = p * 64 + lane;
      if (i < h.nv) {
        float t = 0.f;
        if (dc(i, 10) != 0.f) {
          float lim_ = dc(i, 7);
          t = -dc(i, 5) * perr[p] - dc(i, 6) * (v[i] + delta[i] * h.dt);
          t = fminf(fmaxf(t, -lim_), lim_);
        }
        tau[i] = t;
      }
    }
    w->sync();
  }

  // ------------------------------------------------------------------ semi-implicit Euler
  SS_DEV void integrate() {
    const Hdr &h = k->h;
    const float dt = h.dt;
    if (lane == 0) {
      float wx = v[3] + dt * a[3], wy = v[4] + dt * a[4], wz = v[5] + dt * a[5];
      float nw = sqrtf(wx * wx + wy * wy + wz * wz);
      float ang = dt * nw, ax, ay, az;
      if (nw < 1e-15f) { ax = 1; ay = 0; az = 0; ang = 0; } else { ax = wx / nw; ay = wy / nw; az = wz / nw; }
      float sh, ch; sincosf(0.5f * ang, &sh, &ch);
      float qw = q[3], qx = q[4], qy = q[5], qz = q[6];
      float n = sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
      if (n < 1e-15f) { qw = 1; qx = qy = qz = 0; } else { float in = 1.f / n; qw *= in; qx *= in; qy *= in; qz *= in; }
      float rw = ch, rx = ax * sh, ry = ay * sh, rz = az * sh;
      q[3] = qw * rw - qx * rx - qy * ry - qz * rz;
      q[4] = qw * rx + qx * rw + qy * rz - qz * ry;
      q[5] = qw * ry - qx * rz + qy * rw + qz * rx;
      q[6] = qw * rz + qx * ry - qy * rx + qz * rw;
    }
    w->sync();                                              // lane 0 read v[3:6] before lanes 3-5 update them
    for (int p = 0; p < DOFP; p++) {
      int i = p * 64 + lane;
      if (i < h.nv) {
        float vn = v[i] + dt * a[i];
        v[i] = vn;
        if (i < 3) q[i] += dt * vn;
        else if (i >= 6) q[i + 1] += dt * vn;
      }
    }
    w->sync();
  }

  SS_DEV bool any_bad(const float *x, int n) {
    int bad = 0;
    for (int i = lane; i < n; i += 64) bad |= is_bad(x[i]);
    return w->any(bad);
  }

  // mj_resetData after a bad qpos / qvel / qacc (MuJoCo autoreset)
  SS_DEV void reset_data() {
    const Hdr &h = k->h;
    for (int i = lane; i < h.nq; i += 64) q[i] = 0.f;
    for (int i = lane; i < h.nv; i += 64) { v[i] = 0.f; a[i] = 0.f; tau[i] = 0.f; }
    w->sync();
    if (lane == 0) { q[0] = h.qpos0_root[0]; q[1] = h.qpos0_root[1]; q[2] = h.qpos0_root[2]; q[3] = 1.f; }
    nwarn_add++;
    w->sync();
  }

  // mj_forward (without integration): leaves qacc in a[], M/C pieces (S, Ic, C) in LDS
  SS_DEV void forward() {
    forward_kin(true);
    make_constraints();
    newton();
  }

  // one mj_step given tau; `next_action` != null: also compute the controller torque for the next substep
  SS_DEV void mj_step(const float *next_action, float abias) {
    const Hdr &h = k->h;
    if (any_bad(q, h.nq) || any_bad(v, h.nv)) reset_data();          // mj_checkPos / mj_checkVel
    for (int attempt = 0; attempt < 2; attempt++) {
      forward();
      if (!any_bad(a, h.nv)) break;                                  // mj_checkAcc
      reset_data();
    }
    integrate();
    if (next_action) controller(next_action, abias);
  }

  // ------------------------------------------------------------------ observations (self_obs_v 1 / 2) + task tail
  SS_DEV void write_obs(float *obs, float tar) {
    const Hdr &h = k->h;
    const ss_env_cfg &cf = k->cfg;
    // heading from remove_base_rot(root quat): rotated x axis = third column of the root rotation
    float hx = R[2], hy = R[5];
    float hn = sqrtf(hx * hx + hy * hy);
    float ch = 1.f, sh = 0.f;
    if (hn > 0.f) { ch = hx / hn; sh = hy / hn; }
    int o = 0;
    if (cf.root_height_obs) { if (lane == 0) obs[0] = q[2]; o = 1; }
    const int nb = h.nb, nd = 3 * (nb - 1);
    if (lane >= 1 && lane < nb) {
      const float *rb = r + 3 * lane;
      float *dst = obs + o + 3 * (lane - 1);
      dst[0] = ch * rb[0] + sh * rb[1]; dst[1] = -sh * rb[0] + ch * rb[1]; dst[2] = rb[2];
    }
    o += nd;
    if (lane < nb) {
      const float *Rb = R + 9 * lane;
      float *dst = obs + o + 6 * lane;
      dst[0] = ch * Rb[0] + sh * Rb[3]; dst[1] = -sh * Rb[0] + ch * Rb[3]; dst[2] = Rb[6];
      dst[3] = ch * Rb[2] + sh * Rb[5]; dst[4] = -sh * Rb[2] + ch * Rb[5]; dst[5] = Rb[8];
    }
    o += 6 * nb;
    if (cf.self_obs_v == 1) {
      if (lane < 2) {
        const float *x = v + 3 * lane;
        float *dst = obs + o + 3 * lane;
        dst[0] = ch * x[0] + sh * x[1]; dst[1] = -sh * x[0] + ch * x[1]; dst[2] = x[2];
      }
      o += 6;
      for (int i = lane; i < nd; i += 64) obs[o + i] = v[6 + i];
      o += nd;
    } else {
      if (lane < nb) {
        float *d0 = obs + o + 3 * lane, *d1 = obs + o + 3 * nb + 3 * lane;
        d0[0] = ch * sv[0] + sh * sv[1]; d0[1] = -sh * sv[0] + ch * sv[1]; d0[2] = sv[2];
        d1[0] = ch * sv[3] + sh * sv[4]; d1[1] = -sh * sv[3] + ch * sv[4]; d1[2] = sv[5];
      }
      o += 6 * nb;
    }
    if (lane == 0) {
      if (cf.task == SS_TASK_SPEED) { obs[o] = ch; obs[o + 1] = -sh; obs[o + 2] = tar; }
      else if (cf.task == SS_TASK_GETUP) obs[o] = tar;
    }
  }
};

// ---------------------------------------------------------------------- per-env driver (all modes)
template <class W, int DOFP, int CANDP>
SS_DEV void run_env(W *w, const KArgs *k, const uint32_t *T, float *L, int env) {
  const Hdr &h = k->h;
  const ss_env_cfg &cf = k->cfg;
  const ss_state &st = k->st;
  if (k->mask && !k->mask[env]) return;
  Sim<W, DOFP, CANDP> sim;
  sim.init(w, k, T, L, env);
  const int lane = sim.lane;
  float *qg = st.qpos + (size_t)env * h.nq, *vg = st.qvel + (size_t)env * h.nv;
  float *qpg = st.qpos_prev + (size_t)env * h.nq, *vpg = st.qvel_prev + (size_t)env * h.nv;
  float *wg = st.qacc_warm + (size_t)env * h.nv;
  float *tk = st.task + (size_t)env * 4;
  const float *act = k->actions ? k->actions + (size_t)env * h.nu : nullptr;
  const float *trand = k->task_rand ? k->task_rand + (size_t)env * 2 : nullptr;
  float *obs = k->obs ? k->obs + (size_t)env * k->obs_size : nullptr;

  if (k->mode == MODE_KINEMATICS) {
    sim.load(sim.q, qg, h.nq);
    w->sync();
    sim.forward_kin(false);
    if (lane < h.nb) {
      for (int c = 0; c < 3; c++) k->out0[((size_t)env * h.nb + lane) * 3 + c] = sim.r[3 * lane + c] + sim.q[c];
      for (int c = 0; c < 9; c++) k->out1[((size_t)env * h.nb + lane) * 9 + c] = sim.R[9 * lane + c];
    }
    return;
  }
  if (k->mode == MODE_DEBUG_FORWARD) {                       // mj_forward at (qpos, qvel) with tau from `actions` as raw torques
    sim.load(sim.q, qg, h.nq); sim.load(sim.v, vg, h.nv); sim.load(sim.a, wg, h.nv);
    for (int i = lane; i < h.nv; i += 64) { int ai = (int)sim.dc(i, 11); sim.tau[i] = (act && ai >= 0) ? act[ai] : 0.f; }
    w->sync();
    sim.forward_kin(true);
    if (lane < h.nb) for (int c = 0; c < 21; c++) sim.Kc[21 * lane + c] = 0.f;
    for (int i = lane; i < h.nv; i += 64) sim.diag[i] = sim.dc(i, 0);
    w->sync();
    sim.assemble_H();
    sim.store(k->out0 + (size_t)env * h.ne, sim.H, h.ne);
    sim.store(k->out1 + (size_t)env * h.nv, sim.C, h.nv);
    w->sync();
    sim.make_constraints();
    sim.newton();
    sim.store(k->out2 + (size_t)env * h.nv, sim.a, h.nv);
    if (lane == 0) { st.solver_iters[env] = sim.iters; st.touch[2 * env] = (int)(sim.touchmask & 0xFFFFFFFFull); st.touch[2 * env + 1] = (int)(sim.touchmask >> 32); }
    return;
  }

  int cur_t = st.cur_t[env];
  float tar = tk[0], change = tk[1], recov = tk[2];
  int nsub = k->nsub;
  // StateInit.Fall draws action = U[0,1) - 0.5 (humanoid_env.py:487): the -0.5 is applied in the controller
  const float abias = (k->mode == MODE_RESET) ? -0.5f : 0.f;

  if (k->mode == MODE_RESET) {
    // HumanoidGetup.reset / HumanoidTask.reset: task targets are resampled with the OLD cur_t
    if (cf.task == SS_TASK_GETUP) recov = (float)cf.recovery_steps;
    if (cf.task != SS_TASK_BASE) {
      float u0 = trand ? trand[0] : 0.f, u1 = trand ? trand[1] : 0.f;
      if (cf.task == SS_TASK_SPEED) {
        tar = (cf.tar_speed_max - cf.tar_speed_min) * u0 + cf.tar_speed_min;
        change = (float)(cur_t + cf.speed_change_min + (int)floorf(u1 * (float)(cf.speed_change_max - cf.speed_change_min)));
      } else {
        tar = (cf.tar_height_max - cf.tar_height_min) * u0 + cf.tar_height_min;
        change = (float)(cur_t + cf.height_change_min + (int)floorf(u1 * (float)(cf.height_change_max - cf.height_change_min)));
      }
    }
    for (int i = lane; i < h.nq; i += 64) sim.q[i] = 0.f;
    for (int i = lane; i < h.nv; i += 64) sim.v[i] = 0.f;
    sim.load(sim.a, wg, h.nv);
    w->sync();
    if (lane == 0) {
      if (cf.state_init == SS_INIT_DEFAULT) { sim.q[2] = 0.94f; sim.q[3] = sim.q[4] = sim.q[5] = sim.q[6] = 0.5f; }
      else { sim.q[2] = 0.3f; sim.q[3] = 1.f; }
    }
    w->sync();
    nsub = 0;
    if (cf.state_init == SS_INIT_FALL) nsub = 3 * cf.control_freq_inv;
  } else {
    sim.load(sim.q, qpg, h.nq); sim.load(sim.v, vpg, h.nv); sim.load(sim.a, wg, h.nv);
    w->sync();
  }

  const float *fa = k->fall_actions ? k->fall_actions + (size_t)env * 3 * h.nu : nullptr;

  if (k->mode == MODE_STEP && cf.task != SS_TASK_BASE) {     // pre_physics_step: update_task
    if ((float)cur_t >= change) {
      float u0 = trand ? trand[0] : 0.f, u1 = trand ? trand[1] : 0.f;
      if (cf.task == SS_TASK_SPEED) {
        tar = (cf.tar_speed_max - cf.tar_speed_min) * u0 + cf.tar_speed_min;
        change = (float)(cur_t + cf.speed_change_min + (int)floorf(u1 * (float)(cf.speed_change_max - cf.speed_change_min)));
      } else {
        tar = (cf.tar_height_max - cf.tar_height_min) * u0 + cf.tar_height_min;
        change = (float)(cur_t + cf.height_change_min + (int)floorf(u1 * (float)(cf.height_change_max - cf.height_change_min)));
      }
    }
  }

  float prev_x = 0.f, prev_y = 0.f;
  if (nsub > 0) {
    // ---- prologue: rebuild the stale M, C at the previous forward state, then the first torque
    if (k->mode != MODE_RESET) {
      sim.forward_kin(true);
      sim.load(sim.q, qg, h.nq); sim.load(sim.v, vg, h.nv);
      w->sync();
    } else {
      sim.forward_kin(true);                                 // mj_forward at the Fall state (humanoid_env.py:484)
    }
    prev_x = sim.q[0]; prev_y = sim.q[1];
    const float *a0 = (k->mode == MODE_RESET) ? fa : act;
    sim.controller(a0, abias);
    for (int s = 0; s < nsub; s++) {
      const float *next = nullptr;
      if (s + 1 < nsub) next = (k->mode == MODE_RESET) ? fa + (size_t)((s + 1) / cf.control_freq_inv) * h.nu : act;
      if (s == nsub - 1) {                                   // state of the last forward = next launch's stale source
        sim.store(qpg, sim.q, h.nq); sim.store(vpg, sim.v, h.nv);
      }
      sim.mj_step(next, abias);
    }
  }

  if (k->mode == MODE_RESET) {
    // reset_sim: mj_forward at the reset state -> stale source == current state; sensors/contacts refreshed
    sim.forward_kin(true);
    sim.make_constraints();
    sim.store(qpg, sim.q, h.nq); sim.store(vpg, sim.v, h.nv);
    cur_t = 0;
  }
  sim.store(qg, sim.q, h.nq); sim.store(vg, sim.v, h.nv); sim.store(wg, sim.a, h.nv);
  if (lane < h.nb) for (int c = 0; c < 6; c++) st.body_vel[((size_t)env * h.nb + lane) * 6 + c] = sim.sv[c];
  const unsigned long long touch = sim.touchmask;
  if (lane == 0) {
    st.touch[2 * env] = (int)(touch & 0xFFFFFFFFull); st.touch[2 * env + 1] = (int)(touch >> 32);
    st.solver_iters[env] = sim.iters;
    if (sim.nwarn_add) st.nwarn[env] += sim.nwarn_add;
  }
  if (k->mode == MODE_SUBSTEP) return;

  // ---- post_physics_step: cur_t, observation (mj_kinematics on the new qpos), reward, reset flags
  if (k->mode == MODE_STEP) {
    cur_t += 1;
    w->sync();
    sim.forward_kin(false);
  }
  if (obs) sim.write_obs(obs, tar);
  if (lane == 0) {
    if (k->mode == MODE_STEP) {
      float rew = 0.f;
      int term = 0, trunc = cur_t > cf.episode_length;
      const int illegal = (touch & k->illegal_mask) != 0ull;
      if (cf.task == SS_TASK_SPEED) {
        float dtc = (float)cf.control_freq_inv * h.dt;
        float vx = (sim.q[0] - prev_x) / dtc, vy = (sim.q[1] - prev_y) / dtc;
        float err = tar - vx;
        rew = expf(-0.25f * (err * err + 0.1f * vy * vy));
        term = illegal;
      } else if (cf.task == SS_TASK_GETUP) {
        float diff = tar - sim.q[2];
        rew = expf(-4.f * diff * diff);
        if (recov > 0.f) { recov -= 1.f; term = 0; trunc = 0; }
        else term = illegal;
      }
      if (k->reward) k->reward[env] = rew;
      if (k->terminated) k->terminated[env] = (uint8_t)term;
      if (k->truncated) k->truncated[env] = (uint8_t)trunc;
    }
    st.cur_t[env] = cur_t;
    tk[0] = tar; tk[1] = change; tk[2] = recov;
  }
}

}  // namespace ss
