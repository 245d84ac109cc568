// ss_kernel.h — one wavefront steps one SMPL-humanoid environment.
//
// This replaces, for the whole env batch, what the reference does per process with
//   15 x ( StablePDController.control   reference smpl_sim/envs/controllers.py:116-190
//        + mujoco.mj_step               call site smpl_sim/envs/humanoid_env.py:450 )
//   + compute_proprioception / reward / reset flags   humanoid_env.py:388-403,455-469, tasks/*.py
// in ONE kernel launch.  It is a new design, not a translation of MuJoCo:
//   * all rigid-body quantities are 6-D spatial vectors in a world-aligned frame whose origin is
//     the root body (keeps |r| < 2 m so float32 cancellation in m r^2 terms stays ~1e-6);
//   * the joint-space matrices are never formed: every system H x = b (Newton Hessian, Stable-PD matrix)
//     is solved by an articulated-body recursion over the joint tree in LDS (aba_solve);
//   * MuJoCo's soft-constraint problem  min_a 1/2 (a-a_s)^T M (a-a_s) + sum_i s_i(J_i a - aref_i)
//     is solved by Newton's method with the contact Jacobian folded into per-body 6x6 matrices:
//     J^T D J = sum_b X_b^T K_b X_b, i.e. the Hessian M + J^T D J is the "mass matrix" of bodies with
//     generalized inertias I_b + K_b, which is exactly what the recursion takes;
//   * lanes are bodies / dofs / matrix entries / contact candidates depending on the stage; data
//     crosses lanes through the env's LDS block only, separated by wave-level syncs.
//
// The source is written against a tiny "wave context" W (lane id, sync, reductions, LDS atomic add)
// so that the identical code is compiled by hipcc for gfx950 and by g++ against the 64-fiber
// wavefront emulator under tests/wave_emu (unit-test infrastructure; never a product path).
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>

#include "ss_hdr.h"
#include "ss_selfcol.h"

#if defined(__HIPCC__)
#define SS_DEV __device__ __forceinline__
#else
#define SS_DEV inline
#endif

// Optional in-kernel stage timing (build with -DSS_PROFILE): lane 0 accumulates shader clock ticks per
// stage and adds them to k->prof[stage] (device array of 64-bit counters) at the end of each env.
#ifdef SS_PROFILE
#define SS_T0() unsigned long long t__ = w->clock()
#define SS_TICK(id) do { unsigned long long n__ = w->clock(); sim.prof[id] += n__ - t__; t__ = n__; } while (0)
#else
#define SS_T0() do {} while (0)
#define SS_TICK(id) do {} while (0)
#endif

// relative tolerance of the exact line search on |phi'(alpha)| (MuJoCo's default opt.ls_tolerance = 0.01)
#ifndef SS_LS_TOL
#define SS_LS_TOL 1e-2f
#endif


namespace ss {

enum { PF_FWD = 0, PF_CONS, PF_NBEGIN, PF_NPREP, PF_ASM, PF_FACTOR, PF_SOLVE, PF_NFIN, PF_SPDPREP, PF_SPDFIN, PF_INTEG, PF_MISC,
       PF_F_P13, PF_F_SYNC1, PF_F_P2, PF_F_BSOL, PF_K_CHAIN,
       PF_K_PRO, PF_K_LEV, PF_K_INERTIA, PF_K_SUMS, PF_K_VPROD, PF_P_CONTACT, PF_P_SUMS, PF_P_GRAD,
       PF_SC_NARROW, PF_SC_BASE, PF_SC_COLS, PF_SC_DENSE, PF_SC_FINAL,
       PF_N_DENSE, PF_N_POOLED, PF_N_T1, PF_N_T2, PF_N_T3, PF_N_T4, PF_N_T5, PF_N_CONTACTS, PF_N_UNKNOWNS, PF_SC_LOCK, PF_COUNT };   // (PF_N_*: counts, not ticks)
#ifdef SS_PROFILE
#define SS_FT0() unsigned long long ft__ = w->clock()
#define SS_FTICK(id) do { unsigned long long n__ = w->clock(); prof[id] += n__ - ft__; ft__ = n__; } while (0)
#define SS_FCOUNT(id, v) do { prof[id] += (unsigned long long)(v); } while (0)
#else
#define SS_FT0() do {} while (0)
#define SS_FTICK(id) do {} while (0)
#define SS_FCOUNT(id, v) do {} while (0)
#endif

struct alignas(16) float4_t { real x, y, z, w; };   // 4 reals: one 16-byte LDS access in the product (float) build

struct Contact {
  real rx, ry, rz;      // contact point relative to the root origin
  real t1x, t1y;        // first tangent (unit, in the floor plane); second = (-t1y, t1x)
  real D;               // 1/R of the 4 pyramid rows
  real aref[4], jar[4], jd[4];
  int body, active;
};
struct Limit { real sign, D, aref, jar, jd; };

// caller-owned HBM arrays are declared float* in the C ABI (include/smplsim_hip.h); the kernel reads them as `real`
// (identity in the product build; the float64 triage build of tests/wave_emu is handed double arrays)
SS_DEV real *gptr(float *p) { return reinterpret_cast<real *>(p); }
SS_DEV const real *gptr(const float *p) { return reinterpret_cast<const real *>(p); }
SS_DEV float4_t ld4(const real *p) { return *reinterpret_cast<const float4_t *>(p); }      // 16-byte aligned LDS row
// sin and cos of an angle of at most a few hundred radians (joint angles, half rotation angles): Cody-Waite reduction
// by pi/2 and the single-precision minimax polynomials on [-pi/4, pi/4] (abs. error ~1e-7).  libm's sincosf carries a
// Payne-Hanek path for huge arguments (private scratch array, ~600 instructions per inlined call site).
SS_DEV void sincos_small(real x, real *sn, real *cs) {
#ifdef SS_F64
  *sn = sin(x); *cs = cos(x);
  return;
#endif
  const real kf = SS_M(rint)(x * 0.636619772367581343f);
  const int ki = (int)kf;
  real r = SS_M(fma)(-kf, 1.5707962512969971f, x);
  r = SS_M(fma)(-kf, 7.5497894158615964e-08f, r);
  const real z = r * r;
  const real sp = r + r * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
  const real cp = 1.0f - 0.5f * z + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
  const real a = (ki & 1) ? cp : sp, b = (ki & 1) ? sp : cp;
  *sn = (ki & 2) ? -a : a;
  *cs = ((ki + 1) & 2) ? -b : b;
}
SS_DEV bool is_bad(real x) { return !(x <= real(1e10) && x >= -real(1e10)); }

// SHAPED: per-env body shapes — the geometry-dependent constants (body table, contact candidates, dof inverse weights) are
// indexed by the env's shape id, and the body offsets of the kinematic chain walk come from the env's own body table (staged
// through the env's LDS slice) instead of the workgroup's shared table.  A separate instantiation: the single-shape code
// is textually what it was.
template <bool SHAPED> struct ShapeTables {};
template <> struct ShapeTables<true> { const real *bodyc_s, *candc_s, *dinvw_s, *geomc_s; const int32_t *pairs_s; };   // this env's tables

// SELFCOL: contacts between the humanoid's own bodies (ss_env_cfg.self_collision).  Their rows couple two bodies, which the
// per-body generalized inertias of the articulated-body solve cannot express.  The Newton system is solved in two parts (aba_solve):
// the bodies of the contacts with an active row and their neighbours up to the root of the elimination tree (the COUPLED SET: a
// subtree that holds the root) keep their joints as unknowns of a dense system — CRBA on the articulated inertias that the
// eliminated subtrees hand up, plus sum_c E_c^T K_c E_c of the two-body rows (E_c: relative spatial acceleration of the contact's
// two bodies, non-zero on the joints between them only) — factorized by the wave in LDS (3x3 blocks, L D L^T); everything outside
// the coupled set is eliminated by the recursion exactly as in the plain solve.  One contact per lane, in registers like the floor
// contacts; no capacity other than the wavefront's 64 lanes (MuJoCo keeps every contact, and so does this).
struct SelfCon {
  int b1, b2;                  // normal points from b1 to b2
  real px, py, pz;             // contact point relative to the root origin
  real nx, ny, nz, t1x, t1y, t1z;
  real D;                      // 1/R of the 4 pyramid rows
  real aref[4], jar[4], jd[4];
};
template <bool SELFCOL> struct SelfColState {};
template <> struct SelfColState<true> {
  real *H, *g, *zb, *gc, *stage, *cand;                    // per-env LDS arrays (HdrSC)
  real *pool;                                              // the workgroup's shared dense block (lock word, then the blocks) or null
  int32_t *list;
  const int32_t *tab;                                      // [nb][3] elimination-tree neighbour | joint << 8 | sign << 16 ; path mask (2 words)
  SelfCon sc;                                              // this lane's contact (lane < nself)
  int nself;                                               // wave-uniform count of body-body contacts of this pass
  unsigned long long amask;                                // contacts with an active pyramid row at the current Newton iterate (newton_prepare)
};

// lean models (ss_tables.h) read the tracked action where it lies in global memory: the pointer exists only in the instantiations that
// can run such a model (a member the SMPL headline kernel does not have: its register allocation is sensitive to every live value)
template <bool MAYBE_LEAN> struct LeanState { const real *act_g = nullptr; SS_DEV const real *action_ptr() const { return act_g; } SS_DEV void set_action_ptr(const real *p) { act_g = p; } };
template <> struct LeanState<false> { SS_DEV const real *action_ptr() const { return nullptr; } SS_DEV void set_action_ptr(const real *) {} };

template <class W, int DOFP, int CANDP, int SLOTP, int NPASS, bool SHAPED = false, class HT = HdrRuntime, bool SELFCOL = false>
struct Sim : ShapeTables<SHAPED>, SelfColState<SELFCOL>, LeanState<HT::maybe_lean> {
  W *w;
  const KArgs *k;
  const uint32_t *T;      // shared tables in LDS (integer tables, then the real-valued ones)
  int lane, env;
  // per-env LDS arrays
  real *S, *R, *r, *V, *Ab, *An, *Ad, *Gb, *tmpb, *Aown, *IA, *Ubuf, *Wst, *Rlocp, *w2p, *q, *v, *a, *tau, *Pb, *delta, *Fb, *actl, *diag, *Iown;
  // per-lane constants
  int bpar, bdep;
  // per-lane state
  Contact con[SLOTP];
  Limit lim[DOFP];
  real perr[DOFP];
  int iters, nwarn_add, pid_on;
  int slot_body[SLOTP];    // body of this lane's contact slot(s): constant for the launch, read once from HBM
#ifdef SS_PROFILE
  unsigned long long prof[PF_COUNT], rt0;
#endif
  unsigned long long touchmask;

  SS_DEV int ti(int off, int i) const { return (int)T[off + i]; }
  // real-valued table entry; offsets (Hdr::o_dofc, o_boff) count reals from the start of the blob.  The float build reads the
  // word and bit-casts it: an integer-typed LDS load cannot alias the float arrays of the env slice, so the compiler keeps
  // table values across the kernel's stores (a float-typed read here cost 16% more instructions)
#ifdef SS_F64
  SS_DEV real tf(int off, int i) const { return reinterpret_cast<const real *>(T)[off + i]; }
#else
  SS_DEV real tf(int off, int i) const { union { uint32_t u; float f; } c; c.u = T[off + i]; return c.f; }
#endif
  // dof constants: in the workgroup's LDS tables, or — lean models (ss_tables.h: the SMPL-X size class) — in the global copy of the
  // blob (L1-resident: read once or twice per mj_step and lane), with the armature, which the Newton iteration reads, in an LDS
  // column of its own
  SS_DEV real tfg(int off, int i) const { return reinterpret_cast<const real *>(k->shared_g)[off + i]; }
  SS_DEV real dc(int dof, int f) const { return HT::lean(k->h) ? tfg(k->h.o_dofc, dof * kDofC + f) : tf(k->h.o_dofc, dof * kDofC + f); }
  SS_DEV float4_t dcq(int dof, int qd) const { return ld4(reinterpret_cast<const real *>(k->shared_g) + k->h.o_dofc + dof * kDofC + 4 * qd); }   // lean: columns 4 qd .. 4 qd + 3 of a dof's row
  SS_DEV real armature(int dof) const { return HT::lean(k->h) ? tf(hdr_o_arm(k->h), dof) : tf(k->h.o_dofc, dof * kDofC); }
  SS_DEV real action_at(int ai) const { return HT::lean(k->h) ? this->action_ptr()[ai] : actl[ai]; }   // the tracked action: LDS copy, or (lean) global
  SS_DEV typename HT::type hdr() const { return HT::view(k->h); }
  SS_DEV const real *bodyc() const { if constexpr (SHAPED) return this->bodyc_s; else return k->bodyc; }
  SS_DEV const real *candc() const { if constexpr (SHAPED) return this->candc_s; else return k->candc; }
  SS_DEV real dof_invweight(int dof) const { if constexpr (SHAPED) return this->dinvw_s[dof]; else return dc(dof, 4); }
  SS_DEV const real *geomc() const { if constexpr (SHAPED) return this->geomc_s; else return k->geomc; }   // geoms in their body frames (pair functions)
  SS_DEV const int32_t *pairs() const { if constexpr (SHAPED) return this->pairs_s; else return k->pairs; }
  // body of a contact slot: box b owns slots 4b'..4b'+3 (b' = box order), capsule ends follow
  SS_DEV int h_box_body(int sl) const { return k->candb[8 * (sl >> 2)] & 255; }
  SS_DEV int h_caps_body(int sl) const { int e = sl - 4 * k->h.nbox; int ci = 8 * k->h.nbox + e; return ci < k->h.ncand ? (k->candb[ci] & 255) : 0; }

  SS_DEV void init(W *w_, const KArgs *k_, const uint32_t *T_, real *L, int env_, real *pool_ = nullptr) {
    w = w_; k = k_; T = T_; lane = w->lane(); env = env_;
    typename HT::type h = HT::view(k->h);
    if constexpr (SHAPED) {
      const size_t sid = (size_t)k->st.shape_id[env];
      this->bodyc_s = k->bodyc + sid * shape_stride(h); this->candc_s = k->candc + sid * h.ncand * kCandC;
      this->dinvw_s = this->bodyc_s + h.nb * kBodyC;
      this->geomc_s = k->geomc ? k->geomc + sid * h.nb * kGeomC : nullptr;
      this->pairs_s = k->pairs ? k->pairs + sid * 2 * k->sc.npair : nullptr;
    }
    S = L + h.l_S; R = L + h.l_R; r = L + h.l_r; V = L + h.l_V; Ab = L + h.l_Ab; An = L + h.l_An; Ad = An;
    Gb = L + h.l_Gb; tmpb = L + h.l_tmp; Aown = L + h.l_Aown; IA = L + h.l_IA; Ubuf = L + h.l_Ubuf; Wst = L + h.l_Wst;
    Rlocp = L + h.l_Rloc; w2p = L + h.l_w2;
    q = L + h.l_q; v = L + h.l_v; a = L + h.l_a; tau = L + h.l_tau; Pb = L + h.l_Pb;
    delta = L + h.l_delta; Fb = L + h.l_Fb; actl = L + h.l_act; diag = L + h.l_diag; Iown = L + h.l_Iown;
    bpar = -1; bdep = -1;
    if (lane < h.nb) { bpar = ti(h.o_bparent, lane); bdep = ti(h.o_ndepth, lane + 1) - 1; }
    if constexpr (SELFCOL) {
      const HdrSC &y = k->sc;
      this->H = L + y.l_H; this->g = L + y.l_g; this->zb = L + y.l_zb; this->gc = L + y.l_gc; this->stage = L + y.l_Wst2; this->cand = L + y.l_cand;
      this->pool = pool_;
      this->list = reinterpret_cast<int32_t *>(L + y.l_list);
      int32_t *tb = reinterpret_cast<int32_t *>(L + y.l_tab);
      for (int i = lane; i < 3 * h.nb; i += 64) tb[i] = k->pairs[y.o_sctab + i];   // (visible after the sync that follows the state load)
      this->tab = tb;
      this->nself = 0; this->amask = 0ull; this->sc = SelfCon{};
      Wst = L + y.l_Wst2;                                    // the base layout's (W, y) slot is part of H (R, r keep living there between solves)
    }
#pragma unroll
    for (int p = 0; p < SLOTP; p++) con[p].active = 0;
#pragma unroll
    for (int p = 0; p < DOFP; p++) lim[p].sign = 0.f;
    iters = 0; nwarn_add = 0; touchmask = 0ull;
#pragma unroll
    for (int p = 0; p < SLOTP; p++) {
      const int sl = p * 64 + lane;
      slot_body[p] = sl < h.nslot ? (sl < 4 * h.nbox ? h_box_body(sl) : h_caps_body(sl)) : 0;
    }
    pid_on = (k->cfg.control_mode == SS_CTRL_SIMPLE_PID && k->st.pid_started) ? k->st.pid_started[env] : 0;
#ifdef SS_PROFILE
    for (int i = 0; i < PF_COUNT; i++) prof[i] = 0ull;
#if defined(SS_PROF_REALTIME) && defined(__HIPCC__)
    rt0 = __builtin_amdgcn_s_memrealtime();
#endif
#endif
  }

  // lane id re-read through an optimizer-opaque move at the top of every stage: lane-derived LDS/table addresses
  // are then recomputed per stage (a few VALU ops) instead of being hoisted out of the state machine and
  // spilled to scratch for the whole launch
  SS_DEV void fresh() { lane = w->opaque_v(lane); }
  // constraint registers are rebuilt by make_constraints() of the same pass; clearing them at the top of a pass ends
  // their live ranges there, so they do not occupy VGPRs (or scratch) across forward_kin
  SS_DEV void drop_constraints() {
#pragma unroll
    for (int p = 0; p < SLOTP; p++) con[p] = Contact{};
#pragma unroll
    for (int p = 0; p < DOFP; p++) { lim[p] = Limit{}; perr[p] = 0.f; }
    if constexpr (SELFCOL) { this->sc = SelfCon{}; this->nself = 0; this->amask = 0ull; }
  }

  // ------------------------------------------------------------------ HBM <-> LDS
  // (SELFCOL instantiations, built -O3: the lane index goes through an optimizer-opaque move, or the row addresses of every HBM array are
  // computed once per env at the top of run_env and sit in scratch until their one use: 512 -> 416 bytes of scratch per lane)
  SS_DEV int lane0() { if constexpr (SELFCOL) return w->opaque_v(lane); else return lane; }
  SS_DEV void load(real *dst, const real *src, int n) { for (int i = lane0(); i < n; i += 64) dst[i] = src[i]; }
  SS_DEV void store(real *dst, const real *src, int n) { for (int i = lane0(); i < n; i += 64) dst[i] = src[i]; }

  // ------------------------------------------------------------------ tree helpers
  // out[b][c] = sum of in[d][c] over the subtree of b = the contiguous index range [b, b + size_b) (depth-first
  // body order): one phase of independent LDS reads instead of a level-by-level sweep.
  // Two phases: (1) every body whose subtree has <= 8 bodies sums its range with one trip of 8 independent reads;
  // (2) the few large subtrees add their "cover" — their own input, the finished sums of their small children and the
  // cover of their large children (tables in ss_tables.h).  A lane-serial read-add chain over a 24-body range costs a
  // full LDS round trip per element and sat on every Newton iteration's critical path (profiles/r01r_*).
  // Must be called between hand-offs; it contains one.
  template <int NC>
  SS_DEV void subtree_sum(const real *in, real *out) {
    typename HT::type h = HT::view(k->h);
    for (int idx = lane; idx < NC * h.n_sumsmall; idx += 64) {
      const int o = idx / NC, c = idx - o * NC, e = ti(h.o_sumsmall, o), b = e & 255, n = e >> 8;
      const real *p = in + b * NC + c;
      real part[8];
#pragma unroll
      for (int u = 0; u < 8; u++) part[u] = p[(u < n ? u : 0) * NC];
#pragma unroll
      for (int u = 1; u < 8; u++) part[u] = u < n ? part[u] : 0.f;
      out[b * NC + c] = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
    }
    if (h.n_sumbig == 0) return;
    w->sync();
    for (int idx = lane; idx < NC * h.n_sumbig; idx += 64) {
      const int o = idx / NC, c = idx - o * NC, e = ti(h.o_sumbig, o), b = e & 255, start = (e >> 8) & 4095, cnt = e >> 20;
      real acc = 0.f;
      for (int j = 0; j < cnt; j += 4) {
        real part[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int t = ti(h.o_sumcover, start + (j + u < cnt ? j + u : j));
          part[u] = (t < 64 ? in : out)[(t & 63) * NC + c];
        }
#pragma unroll
        for (int u = 1; u < 4; u++) part[u] = j + u < cnt ? part[u] : 0.f;
        acc += (part[0] + part[1]) + (part[2] + part[3]);
      }
      out[b * NC + c] = acc;
    }
  }

  // A[b] = sum over the dofs d on the chain of body b of S[d] * x[d]   (spatial accel without bias).
  // Two stages: per-node contributions c_n = S[3n..3n+2] x[3n..3n+2] (into `tmp`, 6 floats per node),
  // then each (body, component) adds the <= depth+1 node contributions along its chain.
  SS_DEV void body_accel(const real *x, real *A, real *tmp) {
    typename HT::type h = HT::view(k->h);
    for (int idx = lane; idx < 6 * h.nn; idx += 64) {
      int n = idx / 6, c = idx - 6 * n, d = 3 * n;
      tmp[idx] = S[6 * d + c] * x[d] + S[6 * d + 6 + c] * x[d + 1] + S[6 * d + 12 + c] * x[d + 2];
    }
    w->sync();
    chain_sum(tmp, A);
  }
  // A[b] = sum of the node contributions tmp[n] (6 floats each) over the chain root .. node(b); optionally a second
  // (tmp2 -> A2) pair in the same pass over the chain table
  SS_DEV void chain_sum(const real *tmp, real *A, const real *tmp2 = nullptr, real *A2 = nullptr) {
    typename HT::type h = HT::view(k->h);
    for (int idx = lane; idx < 6 * h.nb; idx += 64) {
      int b = idx / 6, c = idx - 6 * b, n = b + 1;
      const int dn = ti(h.o_ndepth, n), row = h.o_chainnode + n * h.nlev;
      real s = 0.f, s2 = 0.f;
      for (int kk = 0; kk <= dn; kk += 4) {                  // 4 independent (table, data) read pairs per trip
        const int k1 = kk + 1 <= dn ? kk + 1 : kk, k2 = kk + 2 <= dn ? kk + 2 : kk, k3 = kk + 3 <= dn ? kk + 3 : kk;
        const int n0 = ti(row, kk), n1 = ti(row, k1), n2 = ti(row, k2), n3 = ti(row, k3);
        const real a0 = tmp[6 * n0 + c], a1 = tmp[6 * n1 + c], a2 = tmp[6 * n2 + c], a3 = tmp[6 * n3 + c];
        s += (a0 + (kk + 1 <= dn ? a1 : 0.f)) + ((kk + 2 <= dn ? a2 : 0.f) + (kk + 3 <= dn ? a3 : 0.f));
        if (tmp2) {
          const real e0 = tmp2[6 * n0 + c], e1 = tmp2[6 * n1 + c], e2 = tmp2[6 * n2 + c], e3 = tmp2[6 * n3 + c];
          s2 += (e0 + (kk + 1 <= dn ? e1 : 0.f)) + ((kk + 2 <= dn ? e2 : 0.f) + (kk + 3 <= dn ? e3 : 0.f));
        }
      }
      A[idx] = s;
      if (tmp2) A2[idx] = s2;
    }
  }


  // ------------------------------------------------------------------ kinematics + velocities + inertia + bias
  // with_dyn = false: positions/orientations only (observation FK)
  SS_DEV void forward_kin(bool with_dyn, bool write_sensors) {
    fresh();
    typename HT::type h = HT::view(k->h);
    SS_FT0();
    real vb[6] = {0, 0, 0, 0, 0, 0}, ab[6] = {0, 0, 0, 0, 0, 0};
    real Rb[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, rb[3] = {0, 0, 0};
    real bc[kBodyC];                                       // this lane's body constants (L1/L2-resident table)
    {
      const float4_t *src = reinterpret_cast<const float4_t *>(bodyc() + (lane < h.nb ? lane : 0) * kBodyC);
      const float4_t b0 = src[0], b1 = src[1], b2 = src[2], b3 = src[3];
      bc[0] = b0.x; bc[1] = b0.y; bc[2] = b0.z; bc[3] = b0.w; bc[4] = b1.x; bc[5] = b1.y; bc[6] = b1.z; bc[7] = b1.w;
      bc[8] = b2.x; bc[9] = b2.y; bc[10] = b2.z; bc[11] = b2.w; bc[12] = b3.x; bc[13] = b3.y; bc[14] = b3.z; bc[15] = b3.w;
    }
    if (lane == 0) {
      real qw = q[3], qx = q[4], qy = q[5], qz = q[6];
      real n = SS_M(sqrt)(qw * qw + qx * qx + qy * qy + qz * qz);
      if (n < real(1e-15)) { qw = 1; qx = qy = qz = 0; } else { real in = 1.f / n; qw *= in; qx *= in; qy *= in; qz *= in; }
      Rb[0] = 1 - 2 * (qy * qy + qz * qz); Rb[1] = 2 * (qx * qy - qw * qz); Rb[2] = 2 * (qx * qz + qw * qy);
      Rb[3] = 2 * (qx * qy + qw * qz); Rb[4] = 1 - 2 * (qx * qx + qz * qz); Rb[5] = 2 * (qy * qz - qw * qx);
      Rb[6] = 2 * (qx * qz - qw * qy); Rb[7] = 2 * (qy * qz + qw * qx); Rb[8] = 1 - 2 * (qx * qx + qy * qy);
#pragma unroll
      for (int i = 0; i < 9; i++) R[i] = Rb[i];
      r[0] = r[1] = r[2] = 0.f;
#pragma unroll
      for (int d = 0; d < 6; d++) for (int c = 0; c < 6; c++) S[6 * d + c] = 0.f;
      S[0 * 6 + 3] = 1.f; S[1 * 6 + 4] = 1.f; S[2 * 6 + 5] = 1.f;
#pragma unroll
      for (int d = 0; d < 3; d++) { S[6 * (3 + d) + 0] = Rb[d]; S[6 * (3 + d) + 1] = Rb[3 + d]; S[6 * (3 + d) + 2] = Rb[6 + d]; }
    }
    // local rotation Rl = Rx Ry Rz and the hinge axes in the parent frame, once per body (not per level)
    real Rl[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, ayl[3] = {0, 1, 0}, azl[3] = {0, 0, 1};
    if (lane >= 1 && lane < h.nb) {
      real sx, cx, sy, cy, sz, cz;
      sincos_small(q[3 * lane + 4], &sx, &cx); sincos_small(q[3 * lane + 5], &sy, &cy); sincos_small(q[3 * lane + 6], &sz, &cz);
      // Rx Ry = [[cy,0,sy],[sx sy,cx,-sx cy],[-cx sy,sx,cx cy]] ; times Rz
      Rl[0] = cy * cz;                 Rl[1] = -cy * sz;                Rl[2] = sy;
      Rl[3] = sx * sy * cz + cx * sz;  Rl[4] = -sx * sy * sz + cx * cz; Rl[5] = -sx * cy;
      Rl[6] = -cx * sy * cz + sx * sz; Rl[7] = cx * sy * sz + sx * cz;  Rl[8] = cx * cy;
      ayl[0] = 0.f; ayl[1] = cx; ayl[2] = sx;                // Rx e_y
      azl[0] = sy; azl[1] = -sx * cy; azl[2] = cx * cy;      // Rx Ry e_z
    }
    // The local rotations go through LDS (level-buffer region, free here) and every body walks its own chain of
    // ancestors from the root: redundant arithmetic across lanes is free, a level-by-level sweep costs one LDS
    // hand-off per tree level on every pass.
    real *Rloc = Rlocp;                                      // (the level buffer; the aliased layout: the head of the idle Aown region)
    if (lane >= 1 && lane < h.nb) {
#pragma unroll
      for (int i = 0; i < 9; i++) Rloc[9 * lane + i] = Rl[i];
      if constexpr (SHAPED) { Rloc[9 * h.nb + 3 * lane] = bc[0]; Rloc[9 * h.nb + 3 * lane + 1] = bc[1]; Rloc[9 * h.nb + 3 * lane + 2] = bc[2]; }
    }
    w->sync();
    SS_FTICK(PF_K_PRO);
    if (lane >= 1 && lane < h.nb) {
      const int n = lane + 1, dn = ti(h.o_ndepth, n), row = h.o_chainnode + n * h.nlev;
      real Rw[9], Rp[9];
#pragma unroll
      for (int i = 0; i < 9; i++) { Rw[i] = R[i]; Rp[i] = Rw[i]; }
      rb[0] = rb[1] = rb[2] = 0.f;
      for (int kq = 2; kq <= dn; kq++) {                      // bodies on the chain below the root, ending with this one
        const int a_ = ti(row, kq) - 1;
        real o0, o1, o2;
        if constexpr (SHAPED) { const real *bo = Rloc + 9 * h.nb + 3 * a_; o0 = bo[0]; o1 = bo[1]; o2 = bo[2]; }
        else { o0 = tf(h.o_boff, 3 * a_); o1 = tf(h.o_boff, 3 * a_ + 1); o2 = tf(h.o_boff, 3 * a_ + 2); }
        real La[9];
#pragma unroll
        for (int i = 0; i < 9; i++) La[i] = Rloc[9 * a_ + i];
#pragma unroll
        for (int i = 0; i < 3; i++) rb[i] += Rw[3 * i] * o0 + Rw[3 * i + 1] * o1 + Rw[3 * i + 2] * o2;
#pragma unroll
        for (int i = 0; i < 9; i++) Rp[i] = Rw[i];
#pragma unroll
        for (int i = 0; i < 3; i++) {
          const real p0 = Rp[3 * i], p1 = Rp[3 * i + 1], p2 = Rp[3 * i + 2];
          Rw[3 * i] = p0 * La[0] + p1 * La[3] + p2 * La[6];
          Rw[3 * i + 1] = p0 * La[1] + p1 * La[4] + p2 * La[7];
          Rw[3 * i + 2] = p0 * La[2] + p1 * La[5] + p2 * La[8];
        }
      }
#pragma unroll
      for (int i = 0; i < 9; i++) Rb[i] = Rw[i];
      real sd[3][6];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const real p0 = Rp[3 * i], p1 = Rp[3 * i + 1], p2 = Rp[3 * i + 2];
        sd[0][i] = p0;                                             // world axes of the x, y, z hinges
        sd[1][i] = p1 * ayl[1] + p2 * ayl[2];
        sd[2][i] = p0 * azl[0] + p1 * azl[1] + p2 * azl[2];
      }
#pragma unroll
      for (int i = 0; i < 9; i++) R[9 * lane + i] = Rb[i];
#pragma unroll
      for (int i = 0; i < 3; i++) r[3 * lane + i] = rb[i];
#pragma unroll
      for (int j = 0; j < 3; j++) {
        sd[j][3] = rb[1] * sd[j][2] - rb[2] * sd[j][1];
        sd[j][4] = rb[2] * sd[j][0] - rb[0] * sd[j][2];
        sd[j][5] = rb[0] * sd[j][1] - rb[1] * sd[j][0];
#pragma unroll
        for (int c = 0; c < 6; c++) S[6 * (3 * n + j) + c] = sd[j][c];
      }
      if (with_dyn) {                                          // this node's terms of the body velocities S_n qd_n and of the
        const real q0 = v[3 * n], q1 = v[3 * n + 1], q2 = v[3 * n + 2];      // body accelerations of the warm start S_n a_n
        const real g0 = a[3 * n], g1 = a[3 * n + 1], g2 = a[3 * n + 2];
        real *w2 = w2p;                                       // free part of the (W, y) region behind R, r (aliased layout: behind the local rotations)
#pragma unroll
        for (int c = 0; c < 6; c++) {
          Ad[6 * n + c] = sd[0][c] * q0 + sd[1][c] * q1 + sd[2][c] * q2;
          w2[6 * n + c] = sd[0][c] * g0 + sd[1][c] * g1 + sd[2][c] * g2;
        }
      }
    } else if (lane == 0 && with_dyn) {                        // root: translation node (0 ; v_lin), rotation node (R w_local ; 0)
      const real wl0 = v[3], wl1 = v[4], wl2 = v[5];
      Ad[0] = Ad[1] = Ad[2] = 0.f; Ad[3] = v[0]; Ad[4] = v[1]; Ad[5] = v[2];
      Ad[6] = Rb[0] * wl0 + Rb[1] * wl1 + Rb[2] * wl2;
      Ad[7] = Rb[3] * wl0 + Rb[4] * wl1 + Rb[5] * wl2;
      Ad[8] = Rb[6] * wl0 + Rb[7] * wl1 + Rb[8] * wl2;
      Ad[9] = Ad[10] = Ad[11] = 0.f;
      real *w2 = w2p;
      const real g0 = a[3], g1 = a[4], g2 = a[5];
      w2[0] = w2[1] = w2[2] = 0.f; w2[3] = a[0]; w2[4] = a[1]; w2[5] = a[2];
      w2[6] = Rb[0] * g0 + Rb[1] * g1 + Rb[2] * g2;
      w2[7] = Rb[3] * g0 + Rb[4] * g1 + Rb[5] * g2;
      w2[8] = Rb[6] * g0 + Rb[7] * g1 + Rb[8] * g2;
      w2[9] = w2[10] = w2[11] = 0.f;
    }
    w->sync();
    SS_FTICK(PF_K_LEV);
    if (!with_dyn) return;
    // ---- body velocities V_b = chain sums of the node terms written above, then the velocity-product accelerations:
    // node terms + chain sums again
    chain_sum(Ad, V, w2p, Ab);                   // Ab: body accelerations of the iterate newton_begin starts from
    w->sync();
    SS_FTICK(PF_K_CHAIN);
    if (lane < h.nn) {
      const int n = lane;
      real d[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (n == 1) {                                          // free joint: (0 ; u x w)
        d[3] = V[4] * V[2] - V[5] * V[1];
        d[4] = V[5] * V[0] - V[3] * V[2];
        d[5] = V[3] * V[1] - V[4] * V[0];
      } else if (n >= 2) {
        const real *vp = V + 6 * ti(h.o_bparent, n - 1);
        real u[6];
#pragma unroll
        for (int c = 0; c < 6; c++) u[c] = vp[c];
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const real qd = v[3 * n + j];
          real sj[6];
#pragma unroll
          for (int c = 0; c < 6; c++) sj[c] = S[6 * (3 * n + j) + c];
          // (w;u) x_m (sw;su) = (w x sw ; w x su + u x sw)
          d[0] += (u[1] * sj[2] - u[2] * sj[1]) * qd; d[1] += (u[2] * sj[0] - u[0] * sj[2]) * qd; d[2] += (u[0] * sj[1] - u[1] * sj[0]) * qd;
          d[3] += (u[1] * sj[5] - u[2] * sj[4] + u[4] * sj[2] - u[5] * sj[1]) * qd;
          d[4] += (u[2] * sj[3] - u[0] * sj[5] + u[5] * sj[0] - u[3] * sj[2]) * qd;
          d[5] += (u[0] * sj[4] - u[1] * sj[3] + u[3] * sj[1] - u[4] * sj[0]) * qd;
#pragma unroll
          for (int c = 0; c < 6; c++) u[c] += sj[c] * qd;
        }
      }
#pragma unroll
      for (int c = 0; c < 6; c++) tmpb[6 * n + c] = d[c];
    }
    w->sync();
    chain_sum(tmpb, Ad);
    w->sync();
    SS_FTICK(PF_K_VPROD);
    if (lane < h.nb) {
#pragma unroll
      for (int c = 0; c < 6; c++) { vb[c] = V[6 * lane + c]; ab[c] = Ad[6 * lane + c]; }
    }
    // ---- body spatial inertia about the root origin (world axes), bias force, sensor velocities
    if (lane < h.nb) {
      const int b = lane;
      real cx_ = rb[0] + Rb[0] * bc[3] + Rb[1] * bc[4] + Rb[2] * bc[5];
      real cy_ = rb[1] + Rb[3] * bc[3] + Rb[4] * bc[4] + Rb[5] * bc[5];
      real cz_ = rb[2] + Rb[6] * bc[3] + Rb[7] * bc[4] + Rb[8] * bc[5];
      real m = bc[6];
      // Ibar = Rb Ibody Rb^T
      real Bm[9] = {bc[7], bc[8], bc[9], bc[8], bc[10], bc[11], bc[9], bc[11], bc[12]};
      real RB[9];
#pragma unroll
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
        RB[3 * i + j] = Rb[3 * i] * Bm[j] + Rb[3 * i + 1] * Bm[3 + j] + Rb[3 * i + 2] * Bm[6 + j];
      real Ixx = RB[0] * Rb[0] + RB[1] * Rb[1] + RB[2] * Rb[2];
      real Ixy = RB[0] * Rb[3] + RB[1] * Rb[4] + RB[2] * Rb[5];
      real Ixz = RB[0] * Rb[6] + RB[1] * Rb[7] + RB[2] * Rb[8];
      real Iyy = RB[3] * Rb[3] + RB[4] * Rb[4] + RB[5] * Rb[5];
      real Iyz = RB[3] * Rb[6] + RB[4] * Rb[7] + RB[5] * Rb[8];
      real Izz = RB[6] * Rb[6] + RB[7] * Rb[7] + RB[8] * Rb[8];
      real Ib[10];
      Ib[0] = m; Ib[1] = m * cx_; Ib[2] = m * cy_; Ib[3] = m * cz_;
      Ib[4] = Ixx + m * (cy_ * cy_ + cz_ * cz_); Ib[5] = Ixy - m * cx_ * cy_; Ib[6] = Ixz - m * cx_ * cz_;
      Ib[7] = Iyy + m * (cx_ * cx_ + cz_ * cz_); Ib[8] = Iyz - m * cy_ * cz_; Ib[9] = Izz + m * (cx_ * cx_ + cy_ * cy_);
      // f = I (a - a_grav) + v x* (I v)
      real ag[6] = {ab[0], ab[1], ab[2], ab[3], ab[4], ab[5] - h.grav};
      real Ia[6], Iv[6], fb[6];
      imul(Ib, ag, Ia); imul(Ib, vb, Iv);
#pragma unroll
      for (int i = 0; i < 10; i++) Iown[10 * b + i] = Ib[i];   // the body's own inertia stays in LDS for the solves of this pass
      fb[0] = Ia[0] + vb[1] * Iv[2] - vb[2] * Iv[1] + vb[4] * Iv[5] - vb[5] * Iv[4];
      fb[1] = Ia[1] + vb[2] * Iv[0] - vb[0] * Iv[2] + vb[5] * Iv[3] - vb[3] * Iv[5];
      fb[2] = Ia[2] + vb[0] * Iv[1] - vb[1] * Iv[0] + vb[3] * Iv[4] - vb[4] * Iv[3];
      fb[3] = Ia[3] + vb[1] * Iv[5] - vb[2] * Iv[4];
      fb[4] = Ia[4] + vb[2] * Iv[3] - vb[0] * Iv[5];
      fb[5] = Ia[5] + vb[0] * Iv[4] - vb[1] * Iv[3];
      // the bias force stays per body: qfrc_bias = sum_b J_b^T Fb_b is never formed in joint space — every solve of this pass takes
      // Fb as (part of) its per-body bias, and the sweep towards the root does the subtree sums it does anyway (rounds 1-3 summed
      // the subtrees here and projected onto the joints: 3.4 % of the step)
#pragma unroll
      for (int c = 0; c < 6; c++) Fb[6 * b + c] = fb[c];
      // framelinvel / frameangvel of the body frame origin (the env reads the LAST forward's values)
      if (write_sensors) {
        real *svo = gptr(k->st.body_vel) + ((size_t)env * h.nb + b) * 6;
        svo[0] = vb[3] + vb[1] * rb[2] - vb[2] * rb[1];
        svo[1] = vb[4] + vb[2] * rb[0] - vb[0] * rb[2];
        svo[2] = vb[5] + vb[0] * rb[1] - vb[1] * rb[0];
        svo[3] = vb[0]; svo[4] = vb[1]; svo[5] = vb[2];
      }
    }
    w->sync();
    SS_FTICK(PF_K_INERTIA);
  }
  // qfrc_bias in joint space (diagnostics only: ss_debug_forward): out = S^T (subtree sums of Fb)
  SS_DEV void joint_bias(real *out) {
    typename HT::type h = HT::view(k->h);
    subtree_sum<6>(Fb, Gb);
    w->sync();
    for (int i = lane; i < h.nv; i += 64) {
      const int n = i / 3, b = n > 0 ? n - 1 : 0;
      real s = 0.f;
#pragma unroll
      for (int c = 0; c < 6; c++) s += S[6 * i + c] * Gb[6 * b + c];
      out[i] = s;
    }
    w->sync();
  }

  // spatial inertia (10 params, about the origin) times motion vector (w;u) -> force (n;f)
  SS_DEV static void imul(const real *I, const real *x, real *y) {
    real m = I[0], cx = I[1], cy = I[2], cz = I[3];
    y[0] = I[4] * x[0] + I[5] * x[1] + I[6] * x[2] + cy * x[5] - cz * x[4];
    y[1] = I[5] * x[0] + I[7] * x[1] + I[8] * x[2] + cz * x[3] - cx * x[5];
    y[2] = I[6] * x[0] + I[8] * x[1] + I[9] * x[2] + cx * x[4] - cy * x[3];
    y[3] = m * x[3] - (cy * x[2] - cz * x[1]);
    y[4] = m * x[4] - (cz * x[0] - cx * x[2]);
    y[5] = m * x[5] - (cx * x[1] - cy * x[0]);
  }

  // ------------------------------------------------------------------ MuJoCo impedance d(r)
  SS_DEV real impedance(real pos, real margin) const {
    const real *si = k->h.solimp;
    real x = (pos - margin) / si[2];
    if (x < 0) x = -x;
    if (x >= 1.f) return si[1];
    if (x <= 0.f) return si[0];
    real y;
    if (si[4] == 1.f) y = x;
    else if (si[4] == 2.f) y = x <= si[3] ? x * x / si[3] : 1.f - (1.f - x) * (1.f - x) / (1.f - si[3]);   // MuJoCo default
    else if (x <= si[3]) y = SS_M(pow)(x, si[4]) / SS_M(pow)(si[3], si[4] - 1.f);
    else y = 1.f - SS_M(pow)(1.f - x, si[4]) / SS_M(pow)(1.f - si[3], si[4] - 1.f);
    return si[0] + y * (si[1] - si[0]);
  }

  // ------------------------------------------------------------------ floor contacts + joint limits
  SS_DEV void make_constraints() {
    fresh();
    typename HT::type h = HT::view(k->h);
    const real pz = q[2], mu = h.mu;
    touchmask = 0ull;
    // contact records are compacted through LDS (the solver region is free here) into one slot per lane:
    // box b keeps at most 4 corners -> slots 4b..4b+3, capsule end e -> slot 4*nbox + e
    real *rec = An;                                          // (An .. Aown .. IA: one contiguous stretch, nothing of it is live between the forward pass and the solves)
#pragma unroll
    for (int p = 0; p < SLOTP; p++) { int sl = p * 64 + lane; if (sl < h.nslot) rec[13 * sl] = 0.f; }
    w->sync();
#pragma unroll
    for (int p = 0; p < CANDP; p++) {
      int qual = 0;
      real dist = 0.f, px = 0, py = 0, pzr = 0, t1x = 0.f, t1y = 1.f;
      const int cidx = p * 64 + lane;
      const bool valid = cidx < h.ncand;
      const int cbp = valid ? k->candb[cidx] : 0;
      const int b = cbp & 255;
      const bool caps = valid && (cbp & 256);
      real cv[kCandC];
      {
        const float4_t *src = reinterpret_cast<const float4_t *>(candc() + (valid ? cidx : 0) * kCandC);
        const float4_t c0 = src[0], c1 = src[1];
        cv[0] = c0.x; cv[1] = c0.y; cv[2] = c0.z; cv[3] = c0.w; cv[4] = c1.x; cv[5] = c1.y; cv[6] = c1.z; cv[7] = c1.w;
      }
      if (valid) {
        const real *Rb = R + 9 * b, *rb = r + 3 * b;
        if (!caps) {
          real ldist = Rb[6] * cv[0] + Rb[7] * cv[1] + Rb[8] * cv[2];
          real dcen = pz + rb[2] + Rb[6] * cv[3] + Rb[7] * cv[4] + Rb[8] * cv[5];
          qual = !(dcen + ldist > h.margin || ldist > 0.f);
          dist = dcen + ldist;
          real lx = cv[0] + cv[3], ly = cv[1] + cv[4], lz = cv[2] + cv[5];
          px = rb[0] + Rb[0] * lx + Rb[1] * ly + Rb[2] * lz;
          py = rb[1] + Rb[3] * lx + Rb[4] * ly + Rb[5] * lz;
          pzr = rb[2] + Rb[6] * lx + Rb[7] * ly + Rb[8] * lz - 0.5f * dist;
        } else {
          real czw = pz + rb[2] + Rb[6] * cv[0] + Rb[7] * cv[1] + Rb[8] * cv[2];
          dist = czw - cv[6];
          qual = !(dist > h.margin);
          px = rb[0] + Rb[0] * cv[0] + Rb[1] * cv[1] + Rb[2] * cv[2];
          py = rb[1] + Rb[3] * cv[0] + Rb[4] * cv[1] + Rb[5] * cv[2];
          pzr = rb[2] + Rb[6] * cv[0] + Rb[7] * cv[1] + Rb[8] * cv[2] - (cv[6] + 0.5f * dist);
          real axx = Rb[0] * cv[3] + Rb[1] * cv[4] + Rb[2] * cv[5];
          real axy = Rb[3] * cv[3] + Rb[4] * cv[4] + Rb[5] * cv[5];
          real nn = SS_M(sqrt)(axx * axx + axy * axy);
          if (nn < real(1e-15)) { t1x = 0.f; t1y = 1.f; } else { t1x = axx / nn; t1y = axy / nn; }   // no hint (sphere, upright capsule): mju_makeFrame's default e_y, like the oracle
        }
      }
      unsigned long long bal = w->ballot(qual && !caps);
      int act = qual, slot = 0;
      if (valid && !caps) {                                   // plane-box: first 4 qualifying corners in index order
        int grp = lane & ~7;
        unsigned bits = (unsigned)((bal >> grp) & 0xFFull) & ((1u << (lane & 7)) - 1u);
        int rank = 0;
        for (unsigned t = bits; t; t &= t - 1) rank++;
        act = qual && rank < 4;
        slot = 4 * (cidx >> 3) + rank;
      } else if (valid) slot = 4 * h.nbox + (cidx - 8 * h.nbox);
      if (act) {
        const real *vb = V + 6 * b;
        real vx = vb[3] + vb[1] * pzr - vb[2] * py;
        real vy = vb[4] + vb[2] * px - vb[0] * pzr;
        real vz = vb[5] + vb[0] * py - vb[1] * px;
        real vt1 = t1x * vx + t1y * vy, vt2 = -t1y * vx + t1x * vy;
        real imp = impedance(dist, h.margin);
        real R0 = (1.f - imp) / imp * cv[7] * (1.f + mu * mu);
        if (R0 < real(1e-15)) R0 = real(1e-15);
        real Rpy = 2.f * mu * mu * R0;
        real kterm = h.K * imp * (dist - h.margin);
        real *o = rec + 13 * slot;
        o[0] = 1.f; o[1] = (real)b; o[2] = px; o[3] = py; o[4] = pzr; o[5] = t1x; o[6] = t1y; o[7] = 1.f / Rpy;
        o[8] = -h.B * (vz + mu * vt1) - kterm;
        o[9] = -h.B * (vz - mu * vt1) - kterm;
        o[10] = -h.B * (vz + mu * vt2) - kterm;
        o[11] = -h.B * (vz - mu * vt2) - kterm;
      }
      touchmask |= w->bor(act ? (1ull << b) : 0ull);         // wave-wide OR: bodies touching the floor
    }
    w->sync();
#pragma unroll
    for (int p = 0; p < SLOTP; p++) {
      Contact &c = con[p];
      const int sl = p * 64 + lane;
      c.active = 0;
      if (sl < h.nslot) {
        const real *o = rec + 13 * sl;
        if (o[0] != 0.f) {
          c.active = 1; c.body = (int)o[1]; c.rx = o[2]; c.ry = o[3]; c.rz = o[4]; c.t1x = o[5]; c.t1y = o[6]; c.D = o[7];
          c.aref[0] = o[8]; c.aref[1] = o[9]; c.aref[2] = o[10]; c.aref[3] = o[11];
        }
      }
    }
    w->sync();                                               // records consumed before the solver region is reused
#pragma unroll
    for (int p = 0; p < DOFP; p++) {
      int i = p * 64 + lane;
      Limit &l = lim[p];
      l.sign = 0.f; l.D = 0.f; l.aref = 0.f; l.jar = 0.f; l.jd = 0.f;
      // (lean models: the dof's constants come from global memory — one 16-byte read of (armature, lo, hi, limited) per dof, all of a
      // lane's dofs requested before the first is used, instead of dependent scalar reads)
      float4_t c0 = {};
      if (HT::lean(k->h)) { if (i >= 6 && i < h.nv) c0 = dcq(i, 0); }
      if (i >= 6 && i < h.nv && (HT::lean(k->h) ? c0.w : dc(i, 3)) != 0.f) {
        real qi = q[i + 1], lo = HT::lean(k->h) ? c0.y : dc(i, 1), hi = HT::lean(k->h) ? c0.z : dc(i, 2), pos = 0.f;
        if (qi - lo < 0.f) { l.sign = 1.f; pos = qi - lo; }
        else if (hi - qi < 0.f) { l.sign = -1.f; pos = hi - qi; }
        if (l.sign != 0.f) {
          real imp = impedance(pos, 0.f);
          real Rr = (1.f - imp) / imp * dof_invweight(i);
          if (Rr < real(1e-15)) Rr = real(1e-15);
          l.D = 1.f / Rr;
          l.aref = -h.B * (l.sign * v[i]) - h.K * imp * pos;
        }
      }
    }
  }

  // ------------------------------------------------------------------ body-body contacts (SELFCOL)
  // world frame of geom b relative to the root origin: centre gp, rotation gm (row-major), from the body frames in LDS
  SS_DEV void geom_frame(int b, const real *gcst, real *gp, real *gm) const {
    const real *Rb = R + 9 * b, *rb = r + 3 * b;
    for (int i = 0; i < 3; i++) {
      gp[i] = rb[i] + Rb[3 * i] * gcst[0] + Rb[3 * i + 1] * gcst[1] + Rb[3 * i + 2] * gcst[2];
      for (int j = 0; j < 3; j++) gm[3 * i + j] = Rb[3 * i] * gcst[6 + j] + Rb[3 * i + 1] * gcst[9 + j] + Rb[3 * i + 2] * gcst[12 + j];
    }
  }
  // contact record fields
  enum { RC_B1 = 0, RC_B2 = 1, RC_POS = 2, RC_N = 5, RC_T1 = 8, RC_D = 11, RC_AREF = 12, RC_JAR = 16, RC_JD = 20 };

  // Collision of the candidate body pairs (static table) at the pose of this pass: every contact found becomes this pass's
  // contact of one lane (registers).  Runs between forward_kin() and make_constraints(): it reads R, r (not yet overwritten by
  // a solve) and the body velocities V, and uses the solver region as scratch.
  SS_DEV void make_self_contacts(bool write_count) {
    if constexpr (SELFCOL) {
      fresh();
      SS_FT0();
      typename HT::type h = HT::view(k->h);
      const int npair = k->sc.npair;
      real *cand = this->cand;                               // candidates [kSelfCand][kCandRec]: pos3 n3 dist (b1 | b2 << 8); the (W, y) slot is idle here
      int32_t *plist = this->list;                           // pairs that passed the bounding-sphere test (<= 64 per round)
      if (lane < h.nb) {
        const real *gcst = geomc() + lane * kGeomC;
        const real *Rb = R + 9 * lane, *rb = r + 3 * lane;
        for (int i = 0; i < 3; i++) this->gc[3 * lane + i] = rb[i] + Rb[3 * i] * gcst[0] + Rb[3 * i + 1] * gcst[1] + Rb[3 * i + 2] * gcst[2];
      }
      w->sync();
      int ncand = 0, nlist = 0, over = 0;
      const int npass = (npair + 63) >> 6;
      // the pair table entry of the NEXT round is requested before this round's test: one global round trip per round would
      // otherwise sit in front of every ballot (the bodies' reaches are folded into the table: no second, dependent load)
      const int32_t *ptab = pairs();
      int pr_n = 0, rs_n = 0;
      if (lane < npair) { pr_n = ptab[2 * lane]; rs_n = ptab[2 * lane + 1]; }
      for (int p = 0; p <= npass; p++) {
        // ---- broad phase of 64 pairs: bounding spheres about the geom centres
        int pass_ = 0, q = p * 64 + lane;
        const int pr = pr_n, rs_bits = rs_n;
        if (p + 1 < npass && q + 64 < npair) { pr_n = ptab[2 * (q + 64)]; rs_n = ptab[2 * (q + 64) + 1]; }
        if (p < npass && q < npair) {
          const int b1 = pr & 255, b2 = pr >> 8;
          float rs; __builtin_memcpy(&rs, &rs_bits, 4);
          const real dx = this->gc[3 * b2] - this->gc[3 * b1], dy = this->gc[3 * b2 + 1] - this->gc[3 * b1 + 1], dz = this->gc[3 * b2 + 2] - this->gc[3 * b1 + 2];
          pass_ = !(dx * dx + dy * dy + dz * dz > (real)rs * (real)rs);
        }
        const unsigned long long bal = w->ballot(pass_);
        SS_FTICK(PF_ASM);
        int cnt = 0;
        for (unsigned long long t_ = bal; t_; t_ &= t_ - 1) cnt++;
        const bool flush = p == npass || nlist + cnt > 64;
        if (flush && nlist > 0) {
          // ---- narrow phase: one listed pair per lane; a pair function writes its contacts into the lane's private piece of the
          // solver region (ss_selfcol.h: no contact records in registers, nothing in scratch memory), from where they are appended
          // to the candidate list.  Pieces are handed out by prefix sums over the listed lanes (box-box pairs need kBoxBoxWork
          // reals, the others kPairOut); what does not fit the region waits for the next pass (usually there is one)
          w->sync();
          int pid = 0, b1 = 0, b2 = 0, kind = 0;              // kind: 0 capsule-capsule, 1 capsule-box, 2 box-box
          bool pending = lane < nlist;
          if (pending) {
            pid = plist[lane];
            const int pr = ptab[2 * pid];
            b1 = pr & 255; b2 = pr >> 8;
            const bool c1 = geomc()[b1 * kGeomC + 15] != real(SS_GEOM_BOX), c2 = geomc()[b2 * kGeomC + 15] != real(SS_GEOM_BOX);
            kind = c1 && c2 ? 0 : (c1 ? 1 : 2);
          }
          const int room = h.l_Wst - h.l_Aown;               // Aown and the level buffer: idle here (R, r live behind them)
#pragma nounroll
          for (;;) {
            const unsigned long long pm = w->ballot(pending);
            if (!pm) break;
            const unsigned long long bbm = w->ballot(pending && kind == 2), below = (1ull << lane) - 1ull;
            const int off = sc::kBoxBoxWork * __builtin_popcountll(bbm & below) + sc::kPairOut * __builtin_popcountll(pm & ~bbm & below);
            const bool go = pending && off + (kind == 2 ? sc::kBoxBoxWork : sc::kPairOut) <= room;
            real *out = Aown + off;
            int n = 0;
#ifndef SS_STUB_PAIRFN
            if (go) {
              const real *g1 = geomc() + b1 * kGeomC, *g2 = geomc() + b2 * kGeomC;
              real p1[3], m1[9], p2[3], m2[9];
              geom_frame(b1, g1, p1, m1); geom_frame(b2, g2, p2, m2);
              const real z1[3] = {g1[3], g1[4], g1[5]}, z2[3] = {g2[3], g2[4], g2[5]};
              const real a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]};
              if (kind == 0) n = sc::capsule_capsule(p1, a1, z1[0], z1[1], p2, a2, z2[0], z2[1], h.margin, out);
#ifndef SS_STUB_CB
              else if (kind == 1) n = sc::capsule_box(p1, a1, z1[0], z1[1], p2, m2, z2, h.margin, out);
#endif
#ifndef SS_STUB_BB
              else if (kind == 2) n = sc::box_box(p1, m1, z1, p2, m2, z2, h.margin, out);
#endif
            }
#endif
            SS_FTICK(PF_SOLVE);
#pragma nounroll
            for (int kq = 0; kq < 8; kq++) {
              const int has = n > kq;
              const unsigned long long m_ = w->ballot(has);
              if (!m_) break;
              int rank = 0, tot = 0;
              for (unsigned long long t_ = m_; t_; t_ &= t_ - 1) tot++;
              for (unsigned long long t_ = m_ & below; t_; t_ &= t_ - 1) rank++;
              const int idx = ncand + rank;
              if (has && idx < kSelfCand) {
                real *o = cand + kCandRec * idx;
                const real *c = out + sc::kConOut * kq;
                o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = c[3]; o[4] = c[4]; o[5] = c[5]; o[6] = c[6];
                o[7] = (real)(b1 | (b2 << 8));
              }
              ncand += tot;
            }
            pending = pending && !go;
          }
          if (ncand > kSelfCand) { ncand = kSelfCand; over = 1; }
          nlist = 0;
          w->sync();
        }
        if (pass_) {
          int rank = 0;
          for (unsigned long long t_ = bal & ((1ull << lane) - 1ull); t_; t_ &= t_ - 1) rank++;
          plist[nlist + rank] = q;
        }
        nlist += cnt;
      }
      w->sync();
      // ---- every candidate becomes a contact, one per lane in the order found (MuJoCo keeps them all).  The 65th and later ones are
      // dropped and the mj_step counts as truncated (3e-5 of the benchmark's mj_steps: humanoids folded into themselves a step or
      // two before their state blows up and is reset)
      const int src = lane < ncand ? lane : -1;               // the candidate this lane turns into its contact
      int nk = ncand;
      SelfCon c{};
      if (src >= 0) {
        const real *o = cand + kCandRec * src;
        const int b1 = (int)o[7] & 255, b2 = (int)o[7] >> 8;
        const real px = o[0], py = o[1], pz_ = o[2], nx = o[3], ny = o[4], nz = o[5], dist = o[6];
        // frame: first tangent from e_y (|n_y| < 0.5) or e_z, orthogonalised against the normal (mj_makeFrame with no hint)
        real t1x = 0, t1y = 0, t1z = 0;
        if (ny < real(0.5) && ny > real(-0.5)) t1y = 1; else t1z = 1;
        const real dp = nx * t1x + ny * t1y + nz * t1z;
        t1x -= nx * dp; t1y -= ny * dp; t1z -= nz * dp;
        const real tn = SS_M(sqrt)(t1x * t1x + t1y * t1y + t1z * t1z);
        if (tn < sc::kMin) { t1x = 1; t1y = 0; t1z = 0; } else { const real it = real(1) / tn; t1x *= it; t1y *= it; t1z *= it; }
        const real t2x = ny * t1z - nz * t1y, t2y = nz * t1x - nx * t1z, t2z = nx * t1y - ny * t1x;
        // relative velocity of the contact point (body 2 minus body 1), body velocities V = (omega ; v at the root origin)
        const real *v1 = V + 6 * b1, *v2 = V + 6 * b2;
        const real wx = v2[0] - v1[0], wy = v2[1] - v1[1], wz = v2[2] - v1[2];
        const real vx = v2[3] - v1[3] + wy * pz_ - wz * py, vy = v2[4] - v1[4] + wz * px - wx * pz_, vz = v2[5] - v1[5] + wx * py - wy * px;
        const real vn = nx * vx + ny * vy + nz * vz, vt1 = t1x * vx + t1y * vy + t1z * vz, vt2 = t2x * vx + t2y * vy + t2z * vz;
        const real mu = h.mu, imp = impedance(dist, h.margin);
        const real wsum = bodyc()[b1 * kBodyC + 13] + bodyc()[b2 * kBodyC + 13];
        real R0 = (real(1) - imp) / imp * wsum * (real(1) + mu * mu);
        if (R0 < real(1e-15)) R0 = real(1e-15);
        const real kterm = h.K * imp * (dist - h.margin);
        c.b1 = b1; c.b2 = b2; c.px = px; c.py = py; c.pz = pz_; c.nx = nx; c.ny = ny; c.nz = nz; c.t1x = t1x; c.t1y = t1y; c.t1z = t1z;
        c.D = real(1) / (real(2) * mu * mu * R0);
        c.aref[0] = -h.B * (vn + mu * vt1) - kterm; c.aref[1] = -h.B * (vn - mu * vt1) - kterm;
        c.aref[2] = -h.B * (vn + mu * vt2) - kterm; c.aref[3] = -h.B * (vn - mu * vt2) - kterm;
      }
      // the dense system holds nmax block rows: a coupled set (the contacts' bodies and their neighbours up to the root) larger than
      // that loses contacts from the end of the list (models with more bodies than kDenseMaxRows only; counts as truncated)
      {
        const int nmax = k->sc.nmax;
        for (;;) {
          unsigned long long pmk = 0ull;
          if (lane < nk) pmk = path_mask(c.b1) | path_mask(c.b2);
          pmk = w->bor(pmk);
          if (__builtin_popcountll(pmk) + 2 <= nmax || nk == 0) break;
          nk--; over = 1;
        }
      }
      if (lane >= nk) c = SelfCon{};
      this->sc = c; this->nself = nk;
      SS_FTICK(PF_SC_NARROW);
      if (write_count && lane == 0 && k->st.self_contacts) k->st.self_contacts[env] = nk;
      if (lane == 0 && k->self_trunc && over) k->self_trunc[env] += 1;   // this mj_step's list was cut
      if (write_count && k->dbg_self && lane < nk) {         // diagnostics: positions made absolute
        real *o = k->dbg_self + ((size_t)env * kMaxSelf + lane) * kSelfRec;
        o[0] = (real)c.b1; o[1] = (real)c.b2; o[2] = c.px + q[0]; o[3] = c.py + q[1]; o[4] = c.pz + q[2];
        o[5] = c.nx; o[6] = c.ny; o[7] = c.nz; o[8] = c.t1x; o[9] = c.t1y; o[10] = c.t1z; o[11] = c.D;
        for (int i = 0; i < 4; i++) { o[12 + i] = c.aref[i]; o[16 + i] = 0; o[20 + i] = 0; }
      }
      w->sync();
    }
  }
  SS_DEV unsigned long long path_mask(int b) const {
    if constexpr (SELFCOL) return (unsigned long long)(uint32_t)this->tab[3 * b + 1] | ((unsigned long long)(uint32_t)this->tab[3 * b + 2] << 32);
    else return 0ull;
  }

  // rows of this lane's body-body contact: jar from the iterate's body accelerations or jd from the direction's
  // (relative acceleration of the contact point, body 2 minus body 1, in the contact frame)
  SS_DEV void eval_self_rows(const real *A, int stride, bool is_delta) {
    if constexpr (SELFCOL) {
      if (lane < this->nself) {
        SelfCon &c = this->sc;
        const real mu = hdr().mu;
        const real *a1 = A + stride * c.b1, *a2 = A + stride * c.b2;
        const real wx = a2[0] - a1[0], wy = a2[1] - a1[1], wz = a2[2] - a1[2];
        const real ax = a2[3] - a1[3] + wy * c.pz - wz * c.py, ay = a2[4] - a1[4] + wz * c.px - wx * c.pz, az = a2[5] - a1[5] + wx * c.py - wy * c.px;
        const real t2x = c.ny * c.t1z - c.nz * c.t1y, t2y = c.nz * c.t1x - c.nx * c.t1z, t2z = c.nx * c.t1y - c.ny * c.t1x;
        const real an = c.nx * ax + c.ny * ay + c.nz * az, at1 = c.t1x * ax + c.t1y * ay + c.t1z * az, at2 = t2x * ax + t2y * ay + t2z * az;
        const real v[4] = {an + mu * at1, an - mu * at1, an + mu * at2, an - mu * at2};
#pragma unroll
        for (int i = 0; i < 4; i++) { if (is_delta) c.jd[i] = v[i]; else c.jar[i] = v[i] - c.aref[i]; }
      }
    }
  }
  // sum over this lane's self-contact rows of D x jd (x < 0 only) and D jd^2, at x = jar + al jd
  SS_DEV void self_ls_terms(real al, real &s1, real &s2) const {
    if constexpr (SELFCOL) {
      if (lane < this->nself) {
        const SelfCon &c = this->sc;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const real x = c.jar[i] + al * c.jd[i];
          if (x < 0) { s1 += c.D * x * c.jd[i]; s2 += c.D * c.jd[i] * c.jd[i]; }
        }
      }
    }
  }
  // world direction of pyramid row i of this lane's contact:  n +- mu t1 (i = 0, 1),  n +- mu t2 (i = 2, 3)
  SS_DEV void self_row_dir(int i, real mu, real *d) const {
    if constexpr (SELFCOL) {
      const SelfCon &c = this->sc;
      const real sg = (i & 1) ? -mu : mu;
      if (i < 2) { d[0] = c.nx + sg * c.t1x; d[1] = c.ny + sg * c.t1y; d[2] = c.nz + sg * c.t1z; }
      else {
        const real t2x = c.ny * c.t1z - c.nz * c.t1y, t2y = c.nz * c.t1x - c.nx * c.t1z, t2z = c.nx * c.t1y - c.ny * c.t1x;
        d[0] = c.nx + sg * t2x; d[1] = c.ny + sg * t2y; d[2] = c.nz + sg * t2z;
      }
    }
  }

  // rows: jar (from A = Ab, x = a) or jd (from A = Ad, x = delta)
  // A: spatial accelerations, body b at A + stride * b
  SS_DEV void eval_rows(const real *A, int stride, const real *x, bool is_delta) {
    eval_self_rows(A, stride, is_delta);
    const real mu = k->h.mu;
#pragma unroll
    for (int p = 0; p < SLOTP; p++) {
      Contact &c = con[p];
      if (!c.active) continue;
      const real *Ab_ = A + stride * c.body;
      real ax = Ab_[3] + Ab_[1] * c.rz - Ab_[2] * c.ry;
      real ay = Ab_[4] + Ab_[2] * c.rx - Ab_[0] * c.rz;
      real az = Ab_[5] + Ab_[0] * c.ry - Ab_[1] * c.rx;
      real t1 = mu * (c.t1x * ax + c.t1y * ay), t2 = mu * (-c.t1y * ax + c.t1x * ay);
      if (is_delta) { c.jd[0] = az + t1; c.jd[1] = az - t1; c.jd[2] = az + t2; c.jd[3] = az - t2; }
      else { c.jar[0] = az + t1 - c.aref[0]; c.jar[1] = az - t1 - c.aref[1]; c.jar[2] = az + t2 - c.aref[2]; c.jar[3] = az - t2 - c.aref[3]; }
    }
#pragma unroll
    for (int p = 0; p < DOFP; p++) {
      int i = p * 64 + lane;
      Limit &l = lim[p];
      if (l.sign == 0.f) continue;
      if (is_delta) l.jd = l.sign * x[i]; else l.jar = l.sign * x[i] - l.aref;
    }
  }

  // ------------------------------------------------------------------ articulated-body solve of  H x = b
  // H = sum_b J_b^T A_b J_b + diag  (J_b = body Jacobian, A_b = 6x6 generalized inertia of body b: its spatial
  // inertia, plus the contact matrix K_b in the Newton system; diag = armature (+ limit rows / Kd dt)).  That is a
  // mass matrix with generalized body inertias, so the system is solved like forward dynamics, by Featherstone's
  // articulated-body recursion over the ELIMINATION TREE (the body tree re-rooted at its centre, see aba_solve) — H is never
  // formed or factored.  All spatial quantities live in one world-aligned frame, so accumulation towards the root is a plain add.
  //   towards the root, node (body b, joint j):  IA = A_b + sum_children IA'_c,  pA = pb_b + sum_children pA'_c
  //        U = IA S'_j, D = S'_j^T U + diag_j, u = b_j - S'_j^T pA, W = U D^-1, y = D^-1 u     (D^-1 by substitution: Ldl3)
  //        IA' = IA - W U^T,  pA' = pA + U y            (handed to the neighbour towards the root)
  //   away from the root:                         x_j = y - W^T a_e,  a_b = a_e + S'_j x_j
  // Lane roles: 8 lanes per node of the level (lane = 8 * slot + r), lane r < 6 owns row r of the node's 6x6 / 6x3
  // quantities; the 3x3 joint-space algebra is redundant per lane.  Two wave syncs per level going towards the root, one going
  // away.  Rows of the level's IA', pA' go through ONE level buffer in LDS (ss_hdr.h) — or stay in registers along a limb
  // (carry, below); (W_r, y_r) are kept per body.
  // On return x holds the solution and An[8 (b + 1) ..] the body accelerations a_b = J_b x.
  // The level loops of the sweeps are fully unrolled where the tree's shape is a compile-time constant (fixed-layout instantiations:
  // 6 / 7 levels): level bounds and record offsets become immediates (+3 % on the SMPL headline, same bits —
  // profiles/r03_centred_elimination.md 16); with a runtime tree they stay loops.
  static constexpr int kUnrollLevels = HT::fixed ? 16 : 1;
  SS_DEV static void st4w(real *p, real a, real b, real c, real d) { float4_t v; v.x = a; v.y = b; v.z = c; v.w = d; *reinterpret_cast<float4_t *>(p) = v; }

  SS_DEV void write_own_inertia() {                          // Aown[b] = expand(Ib), packed upper triangle (ang;lin)
    if (lane < hdr().nb) {
      real *o = Aown + hdr().a_stride * lane;
      const real *Ib = Iown + 10 * lane;
      const real m = Ib[0], cx = Ib[1], cy = Ib[2], cz = Ib[3];
      o[0] = Ib[4]; o[1] = Ib[5]; o[2] = Ib[6]; o[3] = 0.f; o[4] = -cz; o[5] = cy;
      o[6] = Ib[7]; o[7] = Ib[8]; o[8] = cz; o[9] = 0.f; o[10] = -cx;
      o[11] = Ib[9]; o[12] = -cy; o[13] = cx; o[14] = 0.f;
      o[15] = m; o[16] = 0.f; o[17] = 0.f; o[18] = m; o[19] = 0.f; o[20] = m;
    }
  }

  // pb: per-body bias force (6 per body; Fb of the forward pass or Pb of the Newton iterate): the system solved is  H x = b - sum_b J_b^T pb_b
  //
  // The elimination tree is rooted at the CENTRE of the body tree (HdrC, ss_tables.h), not at the pelvis.  H x = b is the system of a
  // free-floating tree: the six free unknowns may sit on any body.  With the root at the tree's centre the sweeps are as deep as the
  // tree's radius (SMPL: 6 levels instead of 8 below the pelvis; SMPL-X: 7 instead of 10).  For a node (body b, joint j, neighbour e
  // towards the root):  a_b = a_e + S' q''_j  with S' = S_j when e is b's kinematic parent and -S_j when the edge is walked against
  // the kinematic direction (e is b's kinematic child and j is e's joint); minimising the node's energy over q''_j gives exactly the
  // recursion above with S':  U = IA S', D = S'^T U + diag, u = b_j - S'^T pA, W = U D^-1, y = D^-1 u, IA' = IA - W U^T,
  // pA' = pA + U y  and  q''_j = y - W^T a_e,  a_b = a_e + S' q''_j  on the way down.  The free joint (identity motion subspace
  // between the world and body 0) constrains nothing: its right-hand side is a bias force -S_fb b_fb on body 0, and its solution is
  // read off body 0's acceleration.  The root has no joint: IA a = -pA, a 6x6 system in world coordinates.
  // Storage: (W, y) per BODY in Wst, accelerations per body in An (slot b + 1, as the consumers expect), x per joint dof; the
  // body-body-contact instantiations keep D's factors per body and the root's block inverses for aba_resolve / aba_columns.
  SS_DEV void aba_solve(real *x, const real *pb, const unsigned long long cmask = 0ull) {
    fresh();
    typename HT::type h = HT::view(k->h);
    const typename HT::tree_type hc = HT::tree(k->hc);
    const int r_ = lane & 7, g = lane >> 3;
    int off[6];                                              // packed-symmetric offsets of row r_
#pragma unroll
    for (int c = 0; c < 6; c++) { const int lo = r_ < c ? r_ : c, hi = r_ < c ? c : r_; off[c] = (lo * (11 - lo)) / 2 + hi; }
    SS_FT0();
    const unsigned long long nk0 = hc.nkpack[0], nk1 = hc.nkpack[1];
    auto NKC = [&](int L) { return (int)((((L - 1) < 16 ? nk0 : nk1) >> (4 * ((L - 1) & 15))) & 15ull) + 1; };   // nodes of level L >= 1
    // the free joint's right-hand side as a force on body 0 (row r_):  (S_fb b_fb)_r = R b_rot (angular rows; node 1's S holds the
    // columns of R) ; b_trans (linear rows)
    auto fb_force = [&]() { return r_ < 3 ? S[18 + r_] * x[3] + S[24 + r_] * x[4] + S[30 + r_] * x[5] : x[r_ - 3]; };
    int s0 = h.nb - 1;                                        // records of level L start at s0(L) = (nodes of the levels before it)
    // 1:1 stretches of the tree (limbs: every node of level L has at most one child, and it sits in the same slot of level L + 1 —
    // ss_tables.h orders the levels for that and reports the levels as a bit mask, a constant of the fixed-layout instantiations): the
    // rows a node hands towards the root are the ones its lane group needs next, so they stay in registers instead of going through the
    // level buffer (2 stores, 2 loads, one hand-off per level)
    real carry[NPASS][7];
#pragma unroll
    for (int ps = 0; ps < NPASS; ps++)
#pragma unroll
      for (int c = 0; c < 7; c++) carry[ps][c] = 0.f;
    auto chain_level = [&](int L) -> bool {                  // level L's nodes take their children's rows from registers
      if constexpr (HT::fixed) return L >= 1 && L < hc.nlev && ((hc.chain >> (L - 1)) & 1ull);
      else return false;
    };
#pragma unroll kUnrollLevels
    for (int L = hc.nlev; L >= 1; --L) {
      const int nk = NKC(L);
      const bool chain_in = chain_level(L), chain_out = chain_level(L - 1);
      s0 -= nk;
      real *cur = IA + (L & 1) * h.ia_stride;
      const real *prev = IA + ((L + 1) & 1) * h.ia_stride;
      real row[NPASS][6], pa[NPASS], Ur[NPASS][3], red[NPASS][9];
      int nod[NPASS], jnt[NPASS];
      real sg[NPASS];
#pragma unroll
      for (int ps = 0; ps < NPASS; ps++) {                    // ---- part 1: articulated row, U_r = IA_r S_j, partial S_j^T U
        const int kk = ps * 8 + g;
        nod[ps] = -1; jnt[ps] = 0;
        if (ps > 0 && ps * 8 >= nk) continue;                 // wave-uniform: a level of <= 8 nodes runs one pass (SMPL-X: 4 of 7 levels)
#pragma unroll
        for (int t = 0; t < 9; t++) red[ps][t] = 0.f;
        if (r_ < 6 && kk < nk) {
          const int e0 = ti(hc.o_lev, 2 * (s0 + kk)), e1 = ti(hc.o_lev, 2 * (s0 + kk) + 1);
          const int b = e0 & 255, jn = (e0 >> 8) & 255, cfirst = e1 & 255, cc = (e1 >> 8) & 255;
          // (a constant of the unrolled level: only the levels between the tree's root and the kinematic root hold negated joints)
          const bool lev_neg = !HT::fixed || L > 32 || ((hc.neg >> (L - 1)) & 1ull);
          const real sgn = (lev_neg && ((e0 >> 24) & 1)) ? real(-1) : real(1);
          nod[ps] = b; jnt[ps] = jn; sg[ps] = sgn;
          real rw[6], pv = pb[6 * b + r_];
          if (!HT::fixed || L == hc.pel_level)               // (a constant of the unrolled level: the test is compiled into body 0's level only)
            if ((e0 >> 25) & 1) pv -= fb_force();
          const real *ao = Aown + h.a_stride * b;
#pragma unroll
          for (int c = 0; c < 6; c++) rw[c] = ao[off[c]];
          real sv[18];                                       // S_j: issued before the child loop, consumed after it
          const real *sn = S + 18 * jn;
#pragma unroll
          for (int t = 0; t < 18; t++) sv[t] = sn[t];
          const real sr0 = sn[r_], sr1 = sn[6 + r_], sr2 = sn[12 + r_];   // row r of S_j (the sign of S' cancels in U D^-1 U^T and in D)
          auto add_child = [&](int j) {
            const real *src = prev + ((cfirst + j) * 6 + r_) * 8;
            const float4_t v0 = ld4(src), v1 = ld4(src + 4);
            rw[0] += v0.x; rw[1] += v0.y; rw[2] += v0.z; rw[3] += v0.w; rw[4] += v1.x; rw[5] += v1.y; pv += v1.z;
          };
          if constexpr (HT::fixed) {                         // the most children of a node of this level is a constant of the unrolled level:
            const int cm = (int)((hc.cpack >> (3 * (L - 1))) & 7ull);   // predicated adds instead of a lane-varying loop (none on the leaf level)
            if (chain_in) {
              // one 1:1 stretch that runs out at the leaf level (SMPL): a slot's registers are zero unless its child wrote them — children
              // sit in their parent's slot, nothing deeper writes the slot of a leaf — so "has a child" need not be tested.  With two
              // stretches (SMPL-X: fingers, trunk) the finger levels' registers are still live at the trunk's: test.
              constexpr unsigned long long ch_ = decltype(hc)::chain, run_ = ch_ ? ch_ / (ch_ & (~ch_ + 1ull)) : 0ull;
              constexpr bool one_run = ch_ != 0ull && (run_ & (run_ + 1ull)) == 0ull && (ch_ >> (decltype(hc)::nlev - 2)) == 1ull;
              if (one_run || cc) {
#pragma unroll
                for (int c = 0; c < 6; c++) rw[c] += carry[ps][c];
                pv += carry[ps][6];
              }
            } else {
#pragma unroll
              for (int j = 0; j < 7; j++) if (j < cm && j < cc) add_child(j);
            }
          } else {
            for (int j = 0; j < cc; j++) add_child(j);
          }
#pragma unroll
          for (int j = 0; j < 3; j++) {
            real acc = 0.f;
#pragma unroll
            for (int c = 0; c < 6; c++) acc += rw[c] * sv[6 * j + c];
            Ur[ps][j] = acc;
          }
#pragma unroll
          for (int c = 0; c < 6; c++) row[ps][c] = rw[c];
          pa[ps] = pv;
          st4w(Ubuf + (kk * 6 + r_) * 4, Ur[ps][0], Ur[ps][1], Ur[ps][2], pv);
          red[ps][0] = sr0 * Ur[ps][0]; red[ps][1] = sr1 * Ur[ps][0]; red[ps][2] = sr1 * Ur[ps][1];
          red[ps][3] = sr2 * Ur[ps][0]; red[ps][4] = sr2 * Ur[ps][1]; red[ps][5] = sr2 * Ur[ps][2];
          red[ps][6] = sr0 * pv; red[ps][7] = sr1 * pv; red[ps][8] = sr2 * pv;
        }
      }
      SS_FTICK(PF_F_P13);
      w->sync();                                              // U rows visible to the node's other lanes
#pragma unroll
      for (int ps = 0; ps < NPASS; ps++) {
        if (ps > 0 && ps * 8 >= nk) continue;
#pragma unroll
        for (int t = 0; t < 9; t++) red[ps][t] = w->sum8(red[ps][t]);
      }
#pragma unroll
      for (int ps = 0; ps < NPASS; ps++) {                    // ---- part 2: joint-space 3x3 algebra, rows handed towards the root
        if (ps > 0 && ps * 8 >= nk) continue;
        const int kk = ps * 8 + g, b = nod[ps], jn = jnt[ps];
        if (b >= 0) {
          float4_t U[6];
#pragma unroll
          for (int c = 0; c < 6; c++) U[c] = ld4(Ubuf + (kk * 6 + c) * 4);
          // with S' = sgn S_j:  D, IA' = IA - W U^T and W U^T do not see the sign; u' = sgn u = sgn b_j - S_j^T pA gives y' = sgn y,
          // pA' = pA + U y' and, on the way back, z = sgn q''_j = y' - W^T a_e, a_b = a_e + S_j z: W and y' are stored for the unsigned S_j
          const real sgn = sg[ps];
          const real u0 = sgn * x[3 * jn] - red[ps][6], u1 = sgn * x[3 * jn + 1] - red[ps][7], u2 = sgn * x[3 * jn + 2] - red[ps][8];
          Ldl3 Dj;
          Dj.factor(red[ps][0] + diag[3 * jn], red[ps][1], red[ps][2] + diag[3 * jn + 1], red[ps][3], red[ps][4], red[ps][5] + diag[3 * jn + 2]);
          real y0, y1, y2, w0, w1, w2;
          Dj.solve(u0, u1, u2, y0, y1, y2);
          const real a0 = Ur[ps][0], a1 = Ur[ps][1], a2 = Ur[ps][2];
          Dj.solve(a0, a1, a2, w0, w1, w2);
          real rn[6];
#pragma unroll
          for (int c = 0; c < 6; c++) rn[c] = row[ps][c] - (w0 * U[c].x + w1 * U[c].y + w2 * U[c].z);
          real pn = pa[ps] + a0 * y0 + a1 * y1 + a2 * y2;
          real yr = r_ == 0 ? y0 : (r_ == 1 ? y1 : y2);
          if constexpr (SELFCOL) {
            // a joint of the coupled set is not eliminated: its body's composite rows pass unchanged (as through a locked joint), and
            // what the dense system needs of it — U = IC S and the bias force — takes the (W, y) slot of the body
            if ((cmask >> b) & 1ull) {
#pragma unroll
              for (int c = 0; c < 6; c++) rn[c] = row[ps][c];
              pn = pa[ps]; w0 = a0; w1 = a1; w2 = a2; yr = pa[ps];
            }
          }
          if (chain_out) {
#pragma unroll
            for (int c = 0; c < 6; c++) carry[ps][c] = rn[c];
            carry[ps][6] = pn;
          } else {
            real *dst = cur + (kk * 6 + r_) * 8;
            st4w(dst, rn[0], rn[1], rn[2], rn[3]); st4w(dst + 4, rn[4], rn[5], pn, 0.f);
          }
          st4w(Wst + (b * 6 + r_) * 4, w0, w1, w2, yr);
        }
      }
      SS_FTICK(PF_F_P2);
      w->sync();
    }
    // ---- root body: no joint, IA a = -pA (the free joint's force too when the root is body 0)
    {
      real *rows = IA;                                        // (over level 1's rows: the lanes that write have read them — ia_stride is 0, ss_hdr.h)
      const real *prev = IA + h.ia_stride;
      const int c_ = hc.root, cc = hc.nlev >= 1 ? NKC(1) : 0;   // every node of level 1 is a child of the root
      real rw[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, pv = 0.f;
      if (lane < 6) {
        pv = pb[6 * c_ + lane];
        if (c_ == 0) pv -= fb_force();
        const real *ao = Aown + h.a_stride * c_;
#pragma unroll
        for (int c = 0; c < 6; c++) rw[c] = ao[off[c]];
        for (int j = 0; j < cc; j++) {
          const real *src = prev + (j * 6 + lane) * 8;
          const float4_t v0 = ld4(src), v1 = ld4(src + 4);
          rw[0] += v0.x; rw[1] += v0.y; rw[2] += v0.z; rw[3] += v0.w; rw[4] += v1.x; rw[5] += v1.y; pv += v1.z;
        }
      }
      bool dense = false;
      if constexpr (SELFCOL) dense = cmask != 0ull;
      if (dense) dense_solve(cmask, x, rw, pv);               // the root's six unknowns are part of the dense system of the coupled joints
      else {
      if (lane < 6) { st4w(rows + 8 * lane, rw[0], rw[1], rw[2], rw[3]); st4w(rows + 8 * lane + 4, rw[4], rw[5], -pv, 0.f); }
      w->sync();
      real A6[6][6], f6[6];                                   // every lane solves the same 6x6 system
#pragma unroll
      for (int i = 0; i < 6; i++) {
        const float4_t v0 = ld4(rows + 8 * i), v1 = ld4(rows + 8 * i + 4);
        A6[i][0] = v0.x; A6[i][1] = v0.y; A6[i][2] = v0.z; A6[i][3] = v0.w; A6[i][4] = v1.x; A6[i][5] = v1.y; f6[i] = v1.z;
      }
#ifndef SS_ROOT_BLOCKS
      // L D L^T of the 6x6 (symmetric positive definite: no pivoting needed), unrolled into registers: 65 multiply-adds and six
      // reciprocals for the factors, 36 for the two substitutions (rounds 1-3: elimination by 3x3 blocks with closed-form inverses,
      // about 30 instructions more).  Pivot order: the linear rows first — their block is the mass of everything the root carries, well
      // conditioned, and what is left for the angular rows is the inertia about the centre of mass; with the angular rows first the
      // trunk joints lose a third of their float32 accuracy (tests/test_kernel_emu.py::test_tree_solve_keeps_float32_accuracy_...)
      {
        constexpr int P_[6] = {3, 4, 5, 0, 1, 2};
        auto Aat = [&](int i, int j) -> real & { const int a_ = P_[i], b_ = P_[j]; return a_ >= b_ ? A6[a_][b_] : A6[b_][a_]; };   // (symmetric: one triangle is used)
        real Lf[6][6], idg[6], dgl[6], z6[6];
#pragma unroll
        for (int j = 0; j < 6; j++) {
          real vk[6];
          real dj = Aat(j, j);
#pragma unroll
          for (int k2 = 0; k2 < j; k2++) { vk[k2] = Lf[j][k2] * dgl[k2]; dj -= Lf[j][k2] * vk[k2]; }
          dgl[j] = dj;
          idg[j] = rcp_nr(dj);
#pragma unroll
          for (int i = j + 1; i < 6; i++) {
            real t_ = Aat(i, j);
#pragma unroll
            for (int k2 = 0; k2 < j; k2++) t_ -= Lf[i][k2] * vk[k2];
            Lf[i][j] = t_ * idg[j];
          }
        }
#pragma unroll
        for (int i = 0; i < 6; i++) {
          real t_ = f6[P_[i]];
#pragma unroll
          for (int k2 = 0; k2 < i; k2++) t_ -= Lf[i][k2] * z6[k2];
          z6[i] = t_;
        }
#pragma unroll
        for (int i = 0; i < 6; i++) z6[i] *= idg[i];
#pragma unroll
        for (int i = 4; i >= 0; i--)
#pragma unroll
          for (int k2 = i + 1; k2 < 6; k2++) z6[i] -= Lf[k2][i] * z6[k2];
#pragma unroll
        for (int i = 0; i < 6; i++) f6[P_[i]] = z6[i];
      }
#else
      // 2x2 block elimination with closed-form 3x3 inverses (shallow dependency chains; an L D L^T over 6 pivots is
      // a 60-deep chain for a lone wave):  A = [P Q; Q^T T],  a_ang = (P - Q T^-1 Q^T)^-1 (f_a - Q T^-1 f_l),
      // a_lin = T^-1 (f_l - Q^T a_ang)
      real Ti[6], Si[6];
      sym3_inverse(A6[3][3], A6[4][3], A6[4][4], A6[5][3], A6[5][4], A6[5][5], Ti);
      real QT[3][3];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const real q0 = A6[i][3], q1 = A6[i][4], q2 = A6[i][5];
        QT[i][0] = q0 * Ti[0] + q1 * Ti[1] + q2 * Ti[2];
        QT[i][1] = q0 * Ti[1] + q1 * Ti[3] + q2 * Ti[4];
        QT[i][2] = q0 * Ti[2] + q1 * Ti[4] + q2 * Ti[5];
      }
      real Sc[3][3], ga[3];
#pragma unroll
      for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j <= i; j++) Sc[i][j] = A6[i][j] - (QT[i][0] * A6[j][3] + QT[i][1] * A6[j][4] + QT[i][2] * A6[j][5]);
        ga[i] = f6[i] - (QT[i][0] * f6[3] + QT[i][1] * f6[4] + QT[i][2] * f6[5]);
      }
      sym3_inverse(Sc[0][0], Sc[1][0], Sc[1][1], Sc[2][0], Sc[2][1], Sc[2][2], Si);
      const real aa0 = Si[0] * ga[0] + Si[1] * ga[1] + Si[2] * ga[2];
      const real aa1 = Si[1] * ga[0] + Si[3] * ga[1] + Si[4] * ga[2];
      const real aa2 = Si[2] * ga[0] + Si[4] * ga[1] + Si[5] * ga[2];
      const real gl0 = f6[3] - (A6[0][3] * aa0 + A6[1][3] * aa1 + A6[2][3] * aa2);
      const real gl1 = f6[4] - (A6[0][4] * aa0 + A6[1][4] * aa1 + A6[2][4] * aa2);
      const real gl2 = f6[5] - (A6[0][5] * aa0 + A6[1][5] * aa1 + A6[2][5] * aa2);
      f6[0] = aa0; f6[1] = aa1; f6[2] = aa2;
      f6[3] = Ti[0] * gl0 + Ti[1] * gl1 + Ti[2] * gl2;
      f6[4] = Ti[1] * gl0 + Ti[3] * gl1 + Ti[4] * gl2;
      f6[5] = Ti[2] * gl0 + Ti[4] * gl1 + Ti[5] * gl2;
#endif
      if (lane < 6) An[8 * (c_ + 1) + lane] = lane == 0 ? f6[0] : lane == 1 ? f6[1] : lane == 2 ? f6[2] : lane == 3 ? f6[3] : lane == 4 ? f6[4] : f6[5];
      if (c_ == 0) {                                          // the free joint's solution: x_trans = a_lin, x_rot = R^T a_ang
        if (lane < 3) x[lane] = lane == 0 ? f6[3] : (lane == 1 ? f6[4] : f6[5]);
        else if (lane < 6) { const int j = lane - 3; x[lane] = S[18 + 6 * j] * f6[0] + S[18 + 6 * j + 1] * f6[1] + S[18 + 6 * j + 2] * f6[2]; }
      }
      w->sync();
      }
    }
    SS_FTICK(PF_F_SYNC1);
    // ---- sweep away from the root:  q''_j = y - W^T a_e,  a_b = a_e + S' q''_j  (row-distributed: one W row per lane, DPP sums)
    s0 = 0;
    real accp[NPASS];                                        // along a 1:1 stretch the neighbour's acceleration is this lane's own of the level before
#pragma unroll
    for (int ps = 0; ps < NPASS; ps++) accp[ps] = 0.f;
#pragma unroll kUnrollLevels
    for (int L = 1; L <= hc.nlev; L++) {
      const int nk = NKC(L);
      const bool pel_level = L == hc.pel_level;               // wave-uniform: this level holds body 0
      const bool chain_dn = chain_level(L - 1);
#pragma unroll
      for (int ps = 0; ps < NPASS; ps++) {
        if (ps > 0 && ps * 8 >= nk) continue;
        const int kk = ps * 8 + g;
        // (only the inputs of the group sums are initialised for the lanes without a row: everything else is read under `active` only —
        //  a default for each of them was a move per value and level)
        const bool active = r_ < 6 && kk < nk;
#if defined(__HIPCC__)
        int b, jn, pel = 0;
        real p0 = 0.f, p1 = 0.f, p2 = 0.f, apr, s_0, s_1, s_2, nsg;
#else                                                         // (the host compiler cannot see that the reads are guarded)
        int b = -1, jn = 0, pel = 0;
        real p0 = 0.f, p1 = 0.f, p2 = 0.f, apr = 0.f, s_0 = 0.f, s_1 = 0.f, s_2 = 0.f, nsg = 0.f;
#endif
        if (active) {
          const int e0 = ti(hc.o_lev, 2 * (s0 + kk)), en = (e0 >> 16) & 255;
          b = e0 & 255; jn = (e0 >> 8) & 255; pel = (e0 >> 25) & 1;
          const bool lev_neg = !HT::fixed || L > 32 || ((hc.neg >> (L - 1)) & 1ull);
          const real sgn = (lev_neg && ((e0 >> 24) & 1)) ? real(-1) : real(1);
          const float4_t wr = ld4(Wst + (b * 6 + r_) * 4);
          const real *sn = S + 18 * jn + r_;
          s_0 = sn[0]; s_1 = sn[6]; s_2 = sn[12]; nsg = -sgn;
          apr = chain_dn ? accp[ps] : An[8 * (en + 1) + r_];
          p0 = wr.x * apr - (r_ == 0 ? wr.w : 0.f); p1 = wr.y * apr - (r_ == 1 ? wr.w : 0.f); p2 = wr.z * apr - (r_ == 2 ? wr.w : 0.f);
        }
        p0 = w->sum8(p0); p1 = w->sum8(p1); p2 = w->sum8(p2);   // = -z = -sgn q''_j in every lane of the group
        real acc = 0.f;
        if (active) {
          if constexpr (SELFCOL) {                             // a coupled joint's z is the dense system's
            if ((cmask >> b) & 1ull) { p0 = -this->zb[3 * b]; p1 = -this->zb[3 * b + 1]; p2 = -this->zb[3 * b + 2]; }
          }
          acc = apr - (s_0 * p0 + s_1 * p1 + s_2 * p2);
          An[8 * (b + 1) + r_] = acc;
          accp[ps] = acc;
          if (r_ < 3) x[3 * jn + r_] = nsg * (r_ == 0 ? p0 : (r_ == 1 ? p1 : p2));
        }
        if (pel_level) {                                       // body 0 reached: the free joint's solution from its acceleration
          real t0 = 0.f, t1 = 0.f, t2 = 0.f;                  // R^T a_ang: lane r < 3 holds component r of a_ang
          if (pel && r_ < 3) { t0 = S[18 + r_] * acc; t1 = S[24 + r_] * acc; t2 = S[30 + r_] * acc; }
          t0 = w->sum8(t0); t1 = w->sum8(t1); t2 = w->sum8(t2);
          if (pel) {
            if (r_ >= 3 && r_ < 6) x[r_ - 3] = acc;             // x_trans = a_lin
            else if (r_ < 3) x[3 + r_] = r_ == 0 ? t0 : (r_ == 1 ? t1 : t2);
          }
        }
      }
      s0 += nk;
      w->sync();
    }
    SS_FTICK(PF_F_BSOL);
  }

  // ---- dense part of aba_solve (SELFCOL): the joints of the coupled set and the root body's six unknowns.
  // In the elimination tree's coordinates a coupled body's acceleration is  a_b = a_root + sum_{j on its way to the root} S_j z_j
  // (z_j = sgn_j q''_j, so that the motion vectors enter unsigned as in the sweeps).  With IC_b the composite of b's own generalized
  // inertia, its coupled descendants' and the articulated inertias handed up by its eliminated ones, U_b = IC_b S_b (the sweep
  // towards the root left U_b and the composite bias force in b's (W, y) slot):
  //     H(i, i) = S_i^T U_i + diag_i,   H(i, k) = U_i^T S_k  (k between i and the root),   H(root, i) = U_i,   H(root, root) = IC_root,
  //     g_i = sgn_i b_i - S_i^T pC_i,   g_root = -pC_root,
  // plus, per contact c with an active row, between bodies b1 -> b2:  sum_rows w w^T  with  w_i = sigma_i S_i^T u  over the joints i on the
  // way from b1 to b2 through the tree (the joints above their meeting point move both bodies alike and drop out — no cancellation of
  // large terms), sigma = +1 on b2's side and -1 on b1's,  u = sqrt(D) (p x d ; d) for an active pyramid row d.
  //
  // Round 6: the system is SCALAR and TILED for the matrix core (v_mfma_f32_16x16x4_f32: exact float32, an fma chain).  Unknown
  // 3 rank(b) + axis for the coupled bodies by body index, the root's angular and linear parts behind them (N = 3 nc + 6), the right-hand
  // side as row N; the lower triangle of 16 x 16 tiles, row-major inside a tile (ss_hdr.h).  One contact = 4 pyramid rows = the K of ONE
  // matrix instruction per tile: lane l evaluates w for (row l >> 4, unknown 16 t + (l & 15)), tile (ti, tj) += W_ti^T W_tj.  Factorization
  // L D L^T by panels of 16 columns: lane = matrix row, the row's 16 panel entries in registers, pivot values passed by v_readlane (no LDS
  // round trip inside a panel, one hand-off per panel instead of one per 3 x 3 pivot block), the trailing tiles updated by 4 matrix
  // instructions each.  Back substitution with one column per lane.  Rounds 4-5 (3 x 3 blocks, per-contact projection + block passes
  // through LDS, one hand-off per block pivot) spent 31.6 k ticks per solve here; profiles/r06_selfcol_*.
  SS_DEV static int drow(int i) { const int t = i >> 4; return 16 * ((t + 1) * (8 * t + (i & 15))); }   // element (i, j <= i) = H[drow(i) + j]  (ss_hdr.h dense_floats)
#if defined(SS_DENSE_NOINLINE) && defined(__HIPCC__)
  __device__ __attribute__((noinline)) void dense_solve(const unsigned long long cmask, real *x, const real *rw, real pv) {
#else
  SS_DEV void dense_solve(const unsigned long long cmask, real *x, const real *rw, real pv) {
#endif
    if constexpr (SELFCOL) {
      fresh();
      SS_FT0();
      typename HT::type h = HT::view(k->h);
      const typename HT::tree_type hc = HT::tree(k->hc);
      real *H = this->H, *g = this->g;
      const int32_t *tab = this->tab;
      int32_t *list = this->list;
      const int nc = __builtin_popcountll(cmask), N = 3 * nc + 6, Np = N + 1, Nt = (Np + 15) >> 4;
      // a system larger than the env's own region takes the workgroup's shared block (ss_hdr.h) for the duration of this solve
      const int hf = dense_floats(Np);
      const bool pooled = hf > k->sc.hloc;
      if (pooled) { w->lock_acquire(reinterpret_cast<int *>(this->pool)); H = this->pool + 4; SS_FTICK(PF_SC_LOCK); }
      SS_FCOUNT(PF_N_DENSE, 1); SS_FCOUNT(PF_N_POOLED, pooled); SS_FCOUNT(PF_N_T1, Nt == 1); SS_FCOUNT(PF_N_T2, Nt == 2); SS_FCOUNT(PF_N_T3, Nt == 3); SS_FCOUNT(PF_N_T4, Nt == 4); SS_FCOUNT(PF_N_T5, Nt >= 5);
      SS_FCOUNT(PF_N_CONTACTS, __builtin_popcountll(this->amask)); SS_FCOUNT(PF_N_UNKNOWNS, N);
      auto rank = [&](int b) { return (int)__builtin_popcountll(cmask & ((1ull << b) - 1ull)); };
      const real mu = h.mu;
      const int l15 = lane & 15, q4 = lane >> 4;              // this lane's place in a tile as the matrix instruction sees it: row l15, columns 4 q4 .. + 3
      auto trow_ok = [&](int ti) { return 16 * ti + l15 < Np; };   // (the last tile row ends with row N: what lies behind it is not ours to write,
      // and a lane whose row is behind it reads the tile row's first row instead — its results are not stored, its operands must be finite and in range)
      auto tq = [&](int ti, int tj) { return 256 * ((ti * (ti + 1)) >> 1) + 16 * (ti + 1) * (trow_ok(ti) ? l15 : 0) + 16 * tj + 4 * q4; };
      // ---- zero the tiles (the level buffer and Aown they lie over are dead: the sweep towards the root is done)
      w->sync();                                              // (the root's lanes have read level 1's rows, which lie in this region)
      for (int i = 4 * lane; i < hf; i += 256) st4w(H + i, 0, 0, 0, 0);
      w->sync();
      // ---- tree part: lane = coupled body; with at most 32 bodies the two halves of the wave share a body's joints towards the root
      // (the chain walk is the long pole of this phase: 5 blocks for a hand or a toe)
      const int halves = h.nb <= 32 ? 2 : 1, half = halves == 2 ? lane >> 5 : 0, bl = halves == 2 ? lane & 31 : lane;
      const int rowN = drow(N);
      if (bl < h.nb && ((cmask >> bl) & 1ull)) {
        const int b = bl, ri = rank(b), t0 = tab[3 * b], jn = (t0 >> 8) & 255;
        const real sgn = (t0 >> 16) & 1 ? real(-1) : real(1);
        real U[6][3], pa[6], sv[18];
#pragma unroll
        for (int r_ = 0; r_ < 6; r_++) { const float4_t v = ld4(Wst + (b * 6 + r_) * 4); U[r_][0] = v.x; U[r_][1] = v.y; U[r_][2] = v.z; pa[r_] = v.w; }
        const real *sn = S + 18 * jn;
#pragma unroll
        for (int t = 0; t < 18; t++) sv[t] = sn[t];
        int dro[3];
#pragma unroll
        for (int a_ = 0; a_ < 3; a_++) dro[a_] = drow(3 * ri + a_);
        if (half == 0) {
          list[ri] = b | (jn << 8);                           // unknown -> body / joint node (the contact rows' lanes look their columns up here)
#pragma unroll
          for (int a_ = 0; a_ < 3; a_++) {
#pragma unroll
            for (int c = 0; c <= a_; c++) {
              real acc = a_ == c ? diag[3 * jn + a_] : real(0);
#pragma unroll
              for (int r_ = 0; r_ < 6; r_++) acc += sv[6 * a_ + r_] * U[r_][c];
              H[dro[a_] + 3 * ri + c] = acc;
            }
            real gg = sgn * x[3 * jn + a_];
#pragma unroll
            for (int r_ = 0; r_ < 6; r_++) gg -= sv[6 * a_ + r_] * pa[r_];
            H[rowN + 3 * ri + a_] = gg;
          }
        }
        if (half == 1 || halves == 1) {                       // the root body's rows
#pragma unroll
          for (int r_ = 0; r_ < 6; r_++) {
            const int rr = drow(3 * nc + r_);
#pragma unroll
            for (int c = 0; c < 3; c++) H[rr + 3 * ri + c] = U[r_][c];
          }
        }
        for (int kb = t0 & 255; kb != hc.root;) {             // the coupled joints between this body and the root: two per trip, one per half
          const int t1 = tab[3 * kb], k2 = t1 & 255;
          const int tgt = (halves == 2 && half == 1) ? k2 : kb;
          if (tgt != hc.root) {
            const int jk = (tab[3 * tgt] >> 8) & 255, rk = rank(tgt);
            const real *sk = S + 18 * jk;
#pragma unroll
            for (int c = 0; c < 3; c++) {
              const int rt = drow(3 * rk + c), ct = 3 * rk + c;
#pragma unroll
              for (int a_ = 0; a_ < 3; a_++) {
                real acc = 0;
#pragma unroll
                for (int r_ = 0; r_ < 6; r_++) acc += U[r_][a_] * sk[6 * c + r_];
                H[ri > rk ? dro[a_] + ct : rt + 3 * ri + a_] = acc;
              }
            }
          }
          kb = halves == 2 ? (k2 == hc.root ? k2 : (tab[3 * k2] & 255)) : k2;
        }
      }
      if (lane < 6) {                                         // the root body's composite rows (lower triangle) and right-hand side
        const int rr = drow(3 * nc + lane);
        for (int c = 0; c <= lane; c++) H[rr + 3 * nc + c] = rw[c];
        H[rowN + 3 * nc + lane] = -pv;
      }
      w->sync();
      SS_FTICK(PF_SC_BASE);
      // ---- the contacts with an active row write their pyramid rows as spatial vectors u = sqrt(D) (p x d ; d) — zero for an inactive row
      // (branch-free: with `if (jar < 0)` around per-row updates the compiler computed all four rows' products up front and parked
      // them in scratch) — and the two sides of their path through the tree as body masks into a staging area, kRec reals per contact:
      // [row][8] (6 used) | side masks (4 words).  The area is what the system leaves of the env's region (all of it when the system
      // is in the shared block), or the geom-centre / solution-by-body arrays (dead here) when that is less than four records;
      // more active contacts than records go in rounds.  (Round 6, first version: 24 v_readlane + 24 selects per contact.)
      constexpr int kRec = 36;
      real *stg = this->gc;
      int cap = (6 * h.nb) / kRec;
      { const int free_ = pooled ? k->sc.hloc : k->sc.hloc - hf;
        if (free_ >= 4 * kRec) { stg = this->H + (pooled ? 0 : hf); cap = free_ / kRec; } }
      auto stage = [&](unsigned long long rem) {               // the first `cap` contacts of rem, by rank
        const bool mine = (rem >> lane) & 1ull;
        const int slot = __builtin_popcountll(rem & ((1ull << lane) - 1ull));
        if (mine && slot < cap) {
          const SelfCon &c = this->sc;
          real *o = stg + kRec * slot;
          const real sd = SS_M(sqrt)(c.D);
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const real wgt = c.jar[i] < 0 ? sd : real(0);
            real d[3];
            self_row_dir(i, mu, d);
            st4w(o + 8 * i, wgt * (c.py * d[2] - c.pz * d[1]), wgt * (c.pz * d[0] - c.px * d[2]), wgt * (c.px * d[1] - c.py * d[0]), wgt * d[0]);
            o[8 * i + 4] = wgt * d[1]; o[8 * i + 5] = wgt * d[2];
          }
          const unsigned long long p1 = path_mask(c.b1), p2 = path_mask(c.b2), s1 = p1 & ~p2, s2 = p2 & ~p1;
          int32_t *om = reinterpret_cast<int32_t *>(o + 32);
          om[0] = (int32_t)(unsigned)s1; om[1] = (int32_t)(unsigned)(s1 >> 32); om[2] = (int32_t)(unsigned)s2; om[3] = (int32_t)(unsigned)(s2 >> 32);
        }
      };
      // ---- the two-body rows on the matrix core.  Lane l holds, per tile column t, the motion-vector column of unknown 16 t + (l & 15)
      // and its body as a bit; per contact it reads pyramid row l >> 4 of the staged record and forms
      // w = sigma S_col^T u; tile (ti, tj) += W_ti^T W_tj is one instruction (A = w of tj, B = w of ti: the instruction's D is the
      // transpose of the row-major tile, so that a lane's four results are one 16-byte row segment)
      struct Col { real s[6]; unsigned lo, hi; };
      auto colinfo = [&](int t, Col &c) {
        const int u = 16 * t + l15;
        c.lo = c.hi = 0u;
#pragma unroll
        for (int i = 0; i < 6; i++) c.s[i] = 0;
        if (u < 3 * nc) {
          const int e = list[u / 3], b = e & 255;
          const real *sn = S + 18 * (e >> 8) + 6 * (u % 3);
#pragma unroll
          for (int i = 0; i < 6; i++) c.s[i] = sn[i];
          if (b < 32) c.lo = 1u << b; else c.hi = 1u << (b - 32);
        }
      };
      struct Row { real u[6]; unsigned m[4]; };
      auto contact_row = [&](int slot) {                      // staged contact `slot`: pyramid row l >> 4 and the side masks
        Row r;
        const real *o = stg + kRec * slot;
        const float4_t v = ld4(o + 8 * q4);
        r.u[0] = v.x; r.u[1] = v.y; r.u[2] = v.z; r.u[3] = v.w; r.u[4] = o[8 * q4 + 4]; r.u[5] = o[8 * q4 + 5];
        const int32_t *om = reinterpret_cast<const int32_t *>(o + 32);
#pragma unroll
        for (int t = 0; t < 4; t++) r.m[t] = (unsigned)om[t];
        return r;
      };
      auto weval = [&](const Col &c, const Row &r) -> real {
        const real d = c.s[0] * r.u[0] + c.s[1] * r.u[1] + c.s[2] * r.u[2] + c.s[3] * r.u[3] + c.s[4] * r.u[4] + c.s[5] * r.u[5];
        const bool on1 = ((r.m[0] & c.lo) | (r.m[1] & c.hi)) != 0u, on2 = ((r.m[2] & c.lo) | (r.m[3] & c.hi)) != 0u;
        return on2 ? d : (on1 ? -d : real(0));
      };
      auto tile_add = [&](int ti, int tj, const real *acc) {
        if (trow_ok(ti)) {
          real *o = H + tq(ti, tj);
          const float4_t v = ld4(o);
          st4w(o, v.x + acc[0], v.y + acc[1], v.z + acc[2], v.w + acc[3]);
        }
      };
      const int Ntc = (3 * nc + 15) >> 4;                     // tile columns that hold joint unknowns
      if (this->amask) {
        if (Ntc <= 4) {                                       // (all but coupled sets of 22 and 23 bodies) every tile stays in registers over the contacts
          Col c0, c1, c2, c3;
          colinfo(0, c0); colinfo(1, c1); colinfo(2, c2); colinfo(3, c3);
          real a00[4] = {0, 0, 0, 0}, a10[4] = {0, 0, 0, 0}, a11[4] = {0, 0, 0, 0}, a20[4] = {0, 0, 0, 0}, a21[4] = {0, 0, 0, 0}, a22[4] = {0, 0, 0, 0},
               a30[4] = {0, 0, 0, 0}, a31[4] = {0, 0, 0, 0}, a32[4] = {0, 0, 0, 0}, a33[4] = {0, 0, 0, 0};
          for (unsigned long long rem = this->amask; rem;) {
          int cnt = __builtin_popcountll(rem);
          if (cnt > cap) cnt = cap;
          w->sync();                                          // (the records of the round before are read)
          stage(rem);
          w->sync();
          for (int i = 0; i < cnt; i++) rem &= rem - 1ull;
          Row rn = contact_row(0);
          for (int sl = 0; sl < cnt; sl++) {
            const Row r = rn;
            rn = contact_row(sl + 1 < cnt ? sl + 1 : sl);     // the next record is on its way while this one is used
            const real w0 = weval(c0, r);
            w->mfma16(w0, w0, a00);
            if (Ntc >= 2) {
              const real w1 = weval(c1, r);
              w->mfma16(w0, w1, a10); w->mfma16(w1, w1, a11);
              if (Ntc >= 3) {
                const real w2 = weval(c2, r);
                w->mfma16(w0, w2, a20); w->mfma16(w1, w2, a21); w->mfma16(w2, w2, a22);
                if (Ntc >= 4) {
                  const real w3 = weval(c3, r);
                  w->mfma16(w0, w3, a30); w->mfma16(w1, w3, a31); w->mfma16(w2, w3, a32); w->mfma16(w3, w3, a33);
                }
              }
            }
          }
          }
          tile_add(0, 0, a00);
          if (Ntc >= 2) { tile_add(1, 0, a10); tile_add(1, 1, a11); }
          if (Ntc >= 3) { tile_add(2, 0, a20); tile_add(2, 1, a21); tile_add(2, 2, a22); }
          if (Ntc >= 4) { tile_add(3, 0, a30); tile_add(3, 1, a31); tile_add(3, 2, a32); tile_add(3, 3, a33); }
        } else {                                              // any size: tile row by tile row, the row's tiles in registers (5 tile columns = the 22 / 23-body sets)
          for (int tb = 0; tb < Ntc; tb += 5) {
          for (int ti = tb; ti < Ntc; ti++) {
            Col ci, cj[5];
            colinfo(ti, ci);
#pragma unroll
            for (int j = 0; j < 5; j++) colinfo(tb + j, cj[j]);
            real acc[5][4];
#pragma unroll
            for (int j = 0; j < 5; j++)
#pragma unroll
              for (int r_ = 0; r_ < 4; r_++) acc[j][r_] = 0;
            for (unsigned long long rem = this->amask; rem;) {
              int cnt = __builtin_popcountll(rem);
              if (cnt > cap) cnt = cap;
              w->sync();
              stage(rem);
              w->sync();
              for (int i = 0; i < cnt; i++) rem &= rem - 1ull;
              for (int sl = 0; sl < cnt; sl++) {
                const Row r = contact_row(sl);
                const real wi = weval(ci, r);
#pragma unroll
                for (int j = 0; j < 5; j++)
                  if (tb + j <= ti) w->mfma16(weval(cj[j], r), wi, acc[j]);
              }
            }
#pragma unroll
            for (int j = 0; j < 5; j++)
              if (tb + j <= ti) tile_add(ti, tb + j, acc[j]);
          }
          }
        }
      }
      w->sync();
      SS_FTICK(PF_SC_COLS);
      // ---- L D L^T by panels of 16 columns.  Lane = matrix row (64 rs + lane for row set rs), a[] = the row's entries in the panel's
      // columns.  Pivot k: d = a_k[k] and the column below it, a_j[k], come by v_readlane from the pivot rows' lanes; every row
      // below does  l = a[k] / d,  a[j] -= l a_j[k]  (rows at or above the pivot run the same instructions on the unused upper triangle
      // of the diagonal tile).  Row N, the right-hand side, rides along: it ends as D^-1 L^-1 g.
      for (int p = 0; 16 * p < N; p++) {
        const int rs = p >> 2, lb = 16 * (p & 3), npiv = N - 16 * p < 16 ? N - 16 * p : 16;
        {
          const int row = 64 * rs + lane;
          const bool have = row >= 16 * p && row < Np;
          real *tr = H + drow(have ? row : 16 * p) + 16 * p;
          real a[16];
          {
            float4_t v0{}, v1{}, v2{}, v3{};
            if (have) { v0 = ld4(tr); v1 = ld4(tr + 4); v2 = ld4(tr + 8); v3 = ld4(tr + 12); }
            a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w; a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
            a[8] = v2.x; a[9] = v2.y; a[10] = v2.z; a[11] = v2.w; a[12] = v3.x; a[13] = v3.y; a[14] = v3.z; a[15] = v3.w;
          }
          real dsel = 0;
#pragma unroll
          for (int kk = 0; kk < 16; kk++) {                 // (all 16 steps, a dummy pivot of 1 behind the last unknown: with an early exit the compiler
            const real dkk = w->bcast(a[kk], lb + kk);       //  kept the loops rolled and indexed a[] through M0, 12 instructions per update)
            const real dk = kk < npiv ? dkk : real(1);       // (what the steps behind the last unknown touch are the unused columns N .. of the tile)
            dsel = l15 == kk ? dk : dsel;
            const real tl = a[kk] * rcp_nr(dk);
#pragma unroll
            for (int j = kk + 1; j < 16; j++) a[j] -= tl * w->bcast(a[kk], lb + j);
            a[kk] = tl;
          }
          if (have) { st4w(tr, a[0], a[1], a[2], a[3]); st4w(tr + 4, a[4], a[5], a[6], a[7]); st4w(tr + 8, a[8], a[9], a[10], a[11]); st4w(tr + 12, a[12], a[13], a[14], a[15]); }
          if (lane < npiv) g[16 * p + lane] = dsel;
        }
        w->sync();
        if (64 * (rs + 1) < Np) {                             // (more than 64 rows) the rows of the later row sets: the same elimination, the pivot rows'
          const int st_ = 16 * (p + 1);                       // entries read from the stored diagonal tile (uniform addresses) instead of the pivot lanes
          const real *Ld = H + drow(16 * p) + 16 * p;
          for (int rs2 = rs + 1; 64 * rs2 < Np; rs2++) {
            const int row = 64 * rs2 + lane;
            const bool have2 = row < Np;
            real *tr = H + drow(have2 ? row : 16 * p) + 16 * p;
            real a[16];
            {
              float4_t v0{}, v1{}, v2{}, v3{};
              if (have2) { v0 = ld4(tr); v1 = ld4(tr + 4); v2 = ld4(tr + 8); v3 = ld4(tr + 12); }
              a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w; a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
              a[8] = v2.x; a[9] = v2.y; a[10] = v2.z; a[11] = v2.w; a[12] = v3.x; a[13] = v3.y; a[14] = v3.z; a[15] = v3.w;
            }
#pragma unroll
            for (int kk = 0; kk < 16; kk++) {                 // (a later row set behind a panel means the panel is full: 16 pivots)
              const real tl = a[kk];
#pragma unroll
              for (int j = kk + 1; j < 16; j++) a[j] -= tl * Ld[st_ * j + kk];
              a[kk] = tl * rcp_nr(g[16 * p + kk]);
            }
            if (have2) { st4w(tr, a[0], a[1], a[2], a[3]); st4w(tr + 4, a[4], a[5], a[6], a[7]); st4w(tr + 8, a[8], a[9], a[10], a[11]); st4w(tr + 12, a[12], a[13], a[14], a[15]); }
          }
          w->sync();
        }
        // the trailing tiles:  T(ti, tj) -= L(ti, p) D L(tj, p)^T, four instructions of K = 4 each
        if (p + 1 < Nt) {
          const float4_t dq = ld4(g + 16 * p + 4 * q4);
          for (int ti = p + 1; ti < Nt; ti++) {
            const float4_t bv = ld4(H + tq(ti, p));
            const bool okr = trow_ok(ti);
            for (int tj = p + 1; tj <= ti; tj++) {
              const float4_t av = ld4(H + tq(tj, p));
              real *o = H + tq(ti, tj);
              const float4_t cv = ld4(o);
              real acc[4] = {cv.x, cv.y, cv.z, cv.w};
              w->mfma16(-(av.x * dq.x), bv.x, acc); w->mfma16(-(av.y * dq.y), bv.y, acc);
              w->mfma16(-(av.z * dq.z), bv.z, acc); w->mfma16(-(av.w * dq.w), bv.w, acc);
              if (okr) st4w(o, acc[0], acc[1], acc[2], acc[3]);
            }
          }
          w->sync();
        }
      }
      SS_FTICK(PF_SC_DENSE);
      // ---- back substitution  L^T z = (row N).  Up to 64 unknowns: lane = column, rows from the last; z_i = (row N)_i - t_i is final when the
      // rows behind it are in, passed on by v_readlane; the row's entries are requested one step ahead, the address walks down the rows
      // (8 vector instructions per row; a first version re-derived the address and branched around the read: 30)
      if (N <= 64) {
        const real zp = H[rowN + (lane < N ? lane : 0)];
        real tacc = 0;
        int ti_ = (N - 1) >> 4, ri_ = (N - 1) & 15;
        int st_ = 16 * (ti_ + 1);
        const real *rp = H + drow(N - 1) + lane;             // (a lane beyond the row's end reads the rows behind it: inside the system, masked below)
        real lnext = *rp;
        for (int i = N - 1; i > 0; i--) {
          const real lcur = lane < i ? lnext : real(0);
          if (ri_ == 0) { ti_--; ri_ = 15; st_ -= 16; rp -= st_; } else { ri_--; rp -= st_; }
          lnext = *rp;
          const real zi = w->bcast(zp - tacc, i);
          tacc += lcur * zi;
        }
        w->sync();
        if (lane < N) g[lane] = zp - tacc;
        w->sync();
      } else {
      // more than 64 unknowns: panel by panel from the last: what the rows behind a panel contribute is
      // t_c = sum_r z_r L[r][c], four matrix instructions per tile (A = z of four rows, the same for every i; B = the rows' entries in the
      // panel's columns: every lane ends with t of column l & 15); inside the panel lane = column with its column of the diagonal tile in
      // registers, 16 steps of v_readlane.  z goes to g[] (over the pivots, which are done).  (First version: one column per lane over
      // all N rows, an LDS read and 30 instructions per row.)
      for (int p = (N - 1) >> 4; p >= 0; p--) {
        real tacc[4] = {0, 0, 0, 0};
        for (int qt = p + 1; 16 * qt < N; qt++) {
          const int r0 = 16 * qt + 4 * q4, st_ = 16 * (qt + 1);
          const float4_t zq = ld4(g + r0);
          // (rows N .. of the last tile row: z is zero there; the address is held inside the system, row N - 1, so that 0 x garbage cannot make a NaN)
          const real *lq = H + drow(16 * qt) + 16 * p + l15;
          const int rl = 4 * q4, rmax = N - 1 - 16 * qt;
          const real b0 = lq[st_ * (rl < rmax ? rl : rmax)], b1 = lq[st_ * (rl + 1 < rmax ? rl + 1 : rmax)], b2 = lq[st_ * (rl + 2 < rmax ? rl + 2 : rmax)], b3 = lq[st_ * (rl + 3 < rmax ? rl + 3 : rmax)];
          w->mfma16(zq.x, b0, tacc); w->mfma16(zq.y, b1, tacc); w->mfma16(zq.z, b2, tacc); w->mfma16(zq.w, b3, tacc);
        }
        const int c0_ = 16 * p, kc = c0_ + l15;
        real cl[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {                          // (unconditional reads of rows inside the system, then selects: no divergent branch per entry)
          const real lv = H[drow(c0_ + i < N ? c0_ + i : c0_) + kc];
          cl[i] = (i > l15 && c0_ + i < N) ? lv : real(0);
        }
        const real yv = H[rowN + kc];
        const real v = (kc < N ? yv : real(0)) - tacc[0];
        real acc = 0, z = 0;
#pragma unroll
        for (int i = 15; i >= 0; i--) {
          const real zi = w->bcast(v - acc, i);
          z = l15 == i ? zi : z;
          acc += cl[i] * zi;
        }
        w->sync();
        if (lane < 16) g[c0_ + lane] = kc < N ? z : real(0);
        w->sync();
      }
      }
      // ---- hand the solution to the sweep away from the root: z by body, the root body's acceleration, the free joint if the root carries it
      if (lane < h.nb && ((cmask >> lane) & 1ull)) {
        const int ri = rank(lane);
        this->zb[3 * lane] = g[3 * ri]; this->zb[3 * lane + 1] = g[3 * ri + 1]; this->zb[3 * lane + 2] = g[3 * ri + 2];
      }
      const int c_ = hc.root;
      if (lane < 6) An[8 * (c_ + 1) + lane] = g[3 * nc + lane];
      if (c_ == 0) {
        if (lane < 3) x[lane] = g[3 * nc + 3 + lane];
        else if (lane < 6) { const int j = lane - 3; x[lane] = S[18 + 6 * j] * g[3 * nc] + S[18 + 6 * j + 1] * g[3 * nc + 1] + S[18 + 6 * j + 2] * g[3 * nc + 2]; }
      }
      w->sync();
      if (pooled) w->lock_release(reinterpret_cast<int *>(this->pool));
      SS_FTICK(PF_SC_FINAL);
    }
  }

  // The joint-space 3x3 D = S^T IA S + diag of a node, factored D = L E L^T (L unit lower triangular, E diagonal) and applied by
  // substitution.  The point is the residual, not the cost: W = U D^-1 has to satisfy W D = U to rounding for the downdate
  // IA' = IA - W U^T to take the joint's freedom out of IA exactly (IA' S = U - W D = 0).  A substitution solve is backward
  // stable whatever D's condition; an explicit cofactor inverse leaves W D - U ~ cond(D) x 6e-8, which acts as a spurious
  // joint stiffness.  D of a trunk joint carries the legs on one side in the centred tree (twist axis vs bending axes: condition
  // ~30): with the cofactor inverse its dofs were 4x less accurate than with this (profiles/r03_centred_elimination.md).
  struct Ldl3 {
    real ie0, ie1, ie2, l10, l20, l21;
    SS_DEV void factor(real d00, real d10, real d11, real d20, real d21, real d22) {
      ie0 = rcp_nr(d00); l10 = d10 * ie0; l20 = d20 * ie0;
      const real t21 = d21 - l20 * d10;
      ie1 = rcp_nr(d11 - l10 * d10); l21 = t21 * ie1;
      ie2 = rcp_nr(d22 - l20 * d20 - l21 * t21);
    }
    SS_DEV void solve(real b0, real b1, real b2, real &s0, real &s1, real &s2) const {
      const real z1 = b1 - l10 * b0, z2 = b2 - l20 * b0 - l21 * z1;
      s2 = z2 * ie2; s1 = z1 * ie1 - l21 * s2; s0 = b0 * ie0 - l10 * s1 - l20 * s2;
    }
  };

  // inverse of the symmetric 3x3 [d00 d10 d20; d10 d11 d21; d20 d21 d22] -> (i00 i01 i02 i11 i12 i22)
  SS_DEV static void sym3_inverse(real d00, real d10, real d11, real d20, real d21, real d22, real *o) {
    const real c00 = d11 * d22 - d21 * d21, c01 = d21 * d20 - d10 * d22, c02 = d10 * d21 - d11 * d20;
    const real id = rcp_nr(d00 * c00 + d10 * c01 + d20 * c02);
    o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
    o[3] = (d00 * d22 - d20 * d20) * id; o[4] = (d10 * d20 - d00 * d21) * id; o[5] = (d00 * d11 - d10 * d10) * id;
  }

  // 1/x: hardware reciprocal + one Newton step
  SS_DEV static real rcp_nr(real x) {
#if defined(__HIPCC__)
    real r = __builtin_amdgcn_rcpf(x);
#else
    real r = 1.0f / x;
#endif
    return r * (2.0f - x * r);
  }

  // joint-space matrix column j = M e_j by a body-level pass (diagnostics only: ss_debug_forward)
  SS_DEV void dump_mass_matrix(real *out) {
    typename HT::type h = HT::view(k->h);
    for (int j = 0; j < h.nv; j++) {
      for (int i = lane; i < h.nv; i += 64) delta[i] = i == j ? 1.f : 0.f;
      w->sync();
      body_accel(delta, Ab, tmpb);
      w->sync();
      if (lane < h.nb) {
        real Ia[6];
        imul(Iown + 10 * lane, Ab + 6 * lane, Ia);
#pragma unroll
        for (int c = 0; c < 6; c++) Ad[6 * lane + c] = Ia[c];
      }
      w->sync();
      subtree_sum<6>(Ad, Gb);
      w->sync();
      for (int i = lane; i < h.nv; i += 64) {
        const int n = i / 3, b = n > 0 ? n - 1 : 0;
        real s_ = i == j ? armature(i) : 0.f;
#pragma unroll
        for (int c = 0; c < 6; c++) s_ += S[6 * i + c] * Gb[6 * b + c];
        out[(size_t)j * h.nv + i] = s_;
      }
      w->sync();
    }
  }

  // ------------------------------------------------------------------ Newton solve of the constrained acceleration
  SS_DEV void ls_eval(real al, real c1, real c2, real &d1, real &d2) {
    real s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int p = 0; p < SLOTP; p++) {
      const Contact &c = con[p];
      if (!c.active) continue;
#pragma unroll
      for (int r_ = 0; r_ < 4; r_++) {
        real x = c.jar[r_] + al * c.jd[r_];
        if (x < 0.f) { s1 += c.D * x * c.jd[r_]; s2 += c.D * c.jd[r_] * c.jd[r_]; }
      }
    }
#pragma unroll
    for (int p = 0; p < DOFP; p++) {
      const Limit &l = lim[p];
      if (l.sign == 0.f) continue;
      real x = l.jar + al * l.jd;
      if (x < 0.f) { s1 += l.D * x * l.jd; s2 += l.D * l.jd * l.jd; }
    }
    self_ls_terms(al, s1, s2);
    d1 = c1 + al * c2 + w->sum(s1);
    d2 = c2 + w->sum(s2);
  }

  // Newton on MuJoCo's convex primal problem, split so that the shared articulated-body solve site sits
  // between newton_prepare() and newton_finish() in the driver's solver loop.
  SS_DEV void newton_begin() {
    fresh();
    eval_rows(Ab, 6, a, false);                              // Ab = J_b a was left by forward_kin's chain sums
  }

  // right-hand side of the Newton system in two parts — joint space (-> delta) and per-body forces Pb —, diagonal terms
  // and the per-body generalized inertias Aown = I_b + K_b
  SS_DEV void newton_prepare() {
    fresh();
    typename HT::type h = HT::view(k->h);
    const real mu = h.mu;
    iters++;
    SS_FT0();
    // ---- contact forces and K_b = sum_rows D u u^T, u = (rho x w ; w).  The (<= 4) contacts of a box sit in
    // 4 adjacent lanes and the 2 ends of a capsule in 2 adjacent lanes (slot layout of make_constraints), so the
    // per-body sums are quad shuffles; the group's first lane owns the body: it writes the body's inertial force
    // I_b a_b plus the contact force into Ad and its generalized inertia I_b + K_b into Aown (every body has one
    // such lane, so there is no separate "base" phase and no atomics).
#pragma unroll
    for (int p = 0; p < SLOTP; p++) {
      const Contact &c = con[p];
      const int sl = p * 64 + lane;
      const bool boxlane = sl < 4 * h.nbox;
      real vals[27];
#pragma unroll
      for (int t = 0; t < 27; t++) vals[t] = 0.f;
      if (c.active) {
        real fx = 0, fy = 0, fz = 0;
        real Wxx = 0, Wxy = 0, Wxz = 0, Wyy = 0, Wyz = 0, Wzz = 0;
#pragma unroll
        for (int r_ = 0; r_ < 4; r_++) {
          const real sgn = (r_ & 1) ? -mu : mu;
          const real wx = r_ < 2 ? sgn * c.t1x : -sgn * c.t1y;
          const real wy = r_ < 2 ? sgn * c.t1y : sgn * c.t1x;
          if (c.jar[r_] < 0.f) {
            const real f = -c.D * c.jar[r_];
            fx += f * wx; fy += f * wy; fz += f;
            Wxx += c.D * wx * wx; Wxy += c.D * wx * wy; Wxz += c.D * wx;
            Wyy += c.D * wy * wy; Wyz += c.D * wy; Wzz += c.D;
          }
        }
        // force part: -(rho x f ; f)
        vals[0] = -(c.ry * fz - c.rz * fy); vals[1] = -(c.rz * fx - c.rx * fz); vals[2] = -(c.rx * fy - c.ry * fx);
        vals[3] = -fx; vals[4] = -fy; vals[5] = -fz;
        // Y = [rho]x W (ang-lin block), Z = rows rho x Y[i,:] (ang-ang block)
        const real Y0 = c.ry * Wxz - c.rz * Wxy, Y1 = c.ry * Wyz - c.rz * Wyy, Y2 = c.ry * Wzz - c.rz * Wyz;
        const real Y3 = c.rz * Wxx - c.rx * Wxz, Y4 = c.rz * Wxy - c.rx * Wyz, Y5 = c.rz * Wxz - c.rx * Wzz;
        const real Y6 = c.rx * Wxy - c.ry * Wxx, Y7 = c.rx * Wyy - c.ry * Wxy, Y8 = c.rx * Wyz - c.ry * Wxz;
        vals[6] = c.ry * Y2 - c.rz * Y1; vals[7] = c.rz * Y0 - c.rx * Y2; vals[8] = c.rx * Y1 - c.ry * Y0;
        vals[9] = Y0; vals[10] = Y1; vals[11] = Y2;
        vals[12] = c.rz * Y3 - c.rx * Y5; vals[13] = c.rx * Y4 - c.ry * Y3;
        vals[14] = Y3; vals[15] = Y4; vals[16] = Y5;
        vals[17] = c.rx * Y7 - c.ry * Y6;
        vals[18] = Y6; vals[19] = Y7; vals[20] = Y8;
        vals[21] = Wxx; vals[22] = Wxy; vals[23] = Wxz; vals[24] = Wyy; vals[25] = Wyz; vals[26] = Wzz;
      }
      // group sums: xor 1 (pairs), then xor 2 for box quads
#pragma unroll
      for (int t = 0; t < 27; t++) {
        real v1 = vals[t] + w->quad_xor1(vals[t]);
        real v2 = w->quad_xor2(v1);
        vals[t] = boxlane ? v1 + v2 : v1;
      }
      const bool leader = (boxlane ? (sl & 3) == 0 : (sl & 1) == 0) && sl < h.nslot;
      if (leader) {
        const int b = slot_body[p];
        const real *I = Iown + 10 * b;
        real Ia[6];
        imul(I, Ab + 6 * b, Ia);
        real *g = Pb + 6 * b, *o = Aown + h.a_stride * b;
#pragma unroll
        for (int t = 0; t < 6; t++) g[t] = Ia[t] + vals[t] + Fb[6 * b + t];
        const real m = I[0], cx = I[1], cy = I[2], cz = I[3];
        o[0] = I[4] + vals[6]; o[1] = I[5] + vals[7]; o[2] = I[6] + vals[8]; o[3] = vals[9]; o[4] = vals[10] - cz; o[5] = vals[11] + cy;
        o[6] = I[7] + vals[12]; o[7] = I[8] + vals[13]; o[8] = vals[14] + cz; o[9] = vals[15]; o[10] = vals[16] - cx;
        o[11] = I[9] + vals[17]; o[12] = vals[18] - cy; o[13] = vals[19] + cx; o[14] = vals[20];
        o[15] = m + vals[21]; o[16] = vals[22]; o[17] = vals[23]; o[18] = m + vals[24]; o[19] = vals[25]; o[20] = m + vals[26];
      }
    }
    SS_FTICK(PF_P_CONTACT);
    // ---- joint-space part of the gradient (the body part S^T sum_subtree Pb is folded into the sweeps as their bias
    // force, so there is no subtree summation here) and the diagonal terms; delta <- right-hand side
#pragma unroll
    for (int p = 0; p < DOFP; p++) {
      int i = p * 64 + lane;
      if (i < h.nv) {
        real s_ = armature(i) * a[i] - tau[i];               // (the bias force is part of Pb)
        real dg = armature(i);
        const Limit &l = lim[p];
        if (l.sign != 0.f && l.jar < 0.f) { s_ += l.sign * l.D * l.jar; dg += l.D; }
        diag[i] = dg; delta[i] = -s_;
      }
    }
    w->sync();
    // ---- body-body rows: the force of this lane's contact as a wrench on its two bodies, through LDS into the per-body forces
    // (the six lanes that add them own one component each, so no two lanes write one word); the rows' part of the Hessian is
    // built by the dense part of the solve from the same registers
    if constexpr (SELFCOL) {
      this->amask = 0ull;
      if (this->nself > 0) {
        const SelfCon &c = this->sc;
        int act = 0;
        real fx = 0, fy = 0, fz = 0;
        if (lane < this->nself) {
#pragma unroll
          for (int i = 0; i < 4; i++)
            if (c.jar[i] < 0) { real d[3]; self_row_dir(i, mu, d); const real f = -c.D * c.jar[i]; fx += f * d[0]; fy += f * d[1]; fz += f * d[2]; act = 1; }
        }
        const unsigned long long am = w->ballot(act);
        this->amask = am;
        if (am) {
          if (act) {
            real *o = this->stage + 8 * lane;
            o[0] = c.py * fz - c.pz * fy; o[1] = c.pz * fx - c.px * fz; o[2] = c.px * fy - c.py * fx; o[3] = fx; o[4] = fy; o[5] = fz;
            o[6] = (real)c.b1; o[7] = (real)c.b2;
          }
          w->sync();
          // lane = body: every lane walks the staged wrenches (uniform addresses) and keeps the sum of those on its own body in registers,
          // one read-modify-write of Pb per body at the end.  (Rounds 2-5: six lanes, one per component, with two LDS read-modify-writes per
          // contact one after the other — a serial chain of ~250 cycles per contact in every Newton iteration, 7 k cycles with 30 contacts.)
          if (lane < h.nb) {
            real f0 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0, f5 = 0;
            const real me = (real)lane;
            for (unsigned long long m_ = am; m_; m_ &= m_ - 1ull) {
              const real *o = this->stage + 8 * __builtin_ctzll(m_);
              const float4_t v0 = ld4(o), v1 = ld4(o + 4);
              const real sg = v1.z == me ? real(1) : (v1.w == me ? real(-1) : real(0));   // + on body 1, - on body 2
              f0 += sg * v0.x; f1 += sg * v0.y; f2 += sg * v0.z; f3 += sg * v0.w; f4 += sg * v1.x; f5 += sg * v1.y;
            }
            real *pb = Pb + 6 * lane;
            pb[0] += f0; pb[1] += f1; pb[2] += f2; pb[3] += f3; pb[4] += f4; pb[5] += f5;
          }
          w->sync();
        }
      }
    }
    SS_FTICK(PF_P_GRAD);
  }

  // exact line search along delta, step, termination test; returns true when the Newton iteration of this mj_step ends.
  // Termination follows mj_solPrimal (MuJoCo engine_solver.c; call site: reference humanoid_env.py:450):
  //     improvement * scale < tolerance  ||  gradient * scale < tolerance ,   scale = 1 / (meaninertia * max(1, nv))
  // with tol = tolerance / scale handed over in cfg.solver_tolerance (ss_api.h):
  //   improvement   the cost decrease of this iteration, evaluated along the search line from the same per-row terms as the
  //                 line search (cost(0) - cost(al) = -(c1 al + c2 al^2 / 2 + sum_rows D/2 [(x1)_-^2 - (x0)_-^2]), differences of
  //                 squares factored): relative accuracy of the rounding unit, where the difference of two evaluated costs
  //                 (what MuJoCo forms in float64) would be all rounding error in float32;
  //   gradient      never formed in joint space here (the body part of the gradient enters the sweeps as bias forces).  The
  //                 test is replaced by its two observable consequences: a full Newton step inside one quadratic piece of the
  //                 cost (al = 1 accepted, active set unchanged) lands on that piece's minimiser, gradient = 0 up to rounding;
  //                 and a Newton decrement -delta.grad below the rounding error of its own evaluation (SS_DG_NOISE times the
  //                 sum of the absolute values of its terms) means the gradient is below the rounding error of the forces it is
  //                 the sum of — the iterate is then left as it is (the direction is rounding noise).
  SS_DEV bool newton_finish() {
    fresh();
    typename HT::type h = HT::view(k->h);
    eval_rows(An + 8, 8, delta, true);                       // aba_solve left the body accelerations of delta in An
    real dg_ = 0.f, dgabs = 0.f, s_a = 0.f, s_b = 0.f;
#pragma unroll
    for (int p = 0; p < DOFP; p++) {                          // delta . gradient, joint-space part (same terms as newton_prepare)
      int i = p * 64 + lane;
      if (i < h.nv) {
        real s_ = armature(i) * a[i] - tau[i];
        const Limit &l = lim[p];
        if (l.sign != 0.f && l.jar < 0.f) s_ += l.sign * l.D * l.jar;
        const real t_ = delta[i] * s_;
        dg_ += t_; dgabs += SS_M(fabs)(t_);
      }
    }
    if (lane < h.nb) {                                        // body part: (J_b delta) . Pb_b
      const real *ab_ = An + 8 * (lane + 1), *pb_ = Pb + 6 * lane;
#pragma unroll
      for (int c = 0; c < 6; c++) { const real t_ = ab_[c] * pb_[c]; dg_ += t_; dgabs += SS_M(fabs)(t_); }
    }
#pragma unroll
    for (int p = 0; p < SLOTP; p++) {
      const Contact &c = con[p];
      if (!c.active) continue;
#pragma unroll
      for (int r_ = 0; r_ < 4; r_++) if (c.jar[r_] < 0.f) { s_a += c.D * c.jar[r_] * c.jd[r_]; s_b += c.D * c.jd[r_] * c.jd[r_]; }
    }
#pragma unroll
    for (int p = 0; p < DOFP; p++) {
      const Limit &l = lim[p];
      if (l.sign != 0.f && l.jar < 0.f) { s_a += l.D * l.jar * l.jd; s_b += l.D * l.jd * l.jd; }
    }
    // body-body rows (SELFCOL): their forces are part of Pb (newton_prepare), so dg_ holds their share of delta . gradient already
    if constexpr (SELFCOL) {
      if (lane < this->nself) {
        const SelfCon &c = this->sc;
#pragma unroll
        for (int i = 0; i < 4; i++) if (c.jar[i] < 0) { s_a += c.D * c.jar[i] * c.jd[i]; s_b += c.D * c.jd[i] * c.jd[i]; }
      }
    }
    dg_ = w->sum(dg_); dgabs = w->sum(dgabs); s_a = w->sum(s_a); s_b = w->sum(s_b);
    // the Newton decrement is rounding noise of its own terms (or the direction is not a descent direction, or NaN): done,
    // the iterate stays
    if (!(-dg_ > SS_DG_NOISE * dgabs)) {
#if defined(SS_TRACE_NEWTON) && !defined(__HIPCC__)
      if (lane == 0) fprintf(stderr, "NT env %d it %d dg %.4e dgabs %.4e : decrement at rounding level\n", env, iters, (double)dg_, (double)dgabs);
#endif
      // a NaN direction (singular pivot, inf bias of a diverging env) is not "converged": MuJoCo would carry it into qacc and
      // mj_checkAcc would reset the env and count a warning — do the same
      if (!(dgabs == dgabs) || !(dg_ == dg_)) { if (lane == 0) a[0] = dg_ + dgabs; w->sync(); }
      return true;
    }
    const real c1 = dg_ - s_a, c2 = -dg_ - s_b;            // phi'(al) = c1 + al c2 + sum_active(al) D (jar + al jd) jd
    real al = 1.f, d1, d2;
    ls_eval(1.f, c1, c2, d1, d2);
    // accept when |phi'| is SS_LS_TOL of phi'(0) = delta.grad, or at the rounding level of its terms
    const real tol = SS_LS_TOL_EFF * SS_M(fabs)(dg_) + SS_ROUND_REL * (SS_M(fabs)(dg_) + SS_M(fabs)(s_a) + SS_M(fabs)(s_b));
    bool exact = true;
    if (!(SS_M(fabs)(d1) <= tol)) {
      exact = false;
      real lo = 0.f, hi = 1.f;
      for (int ls = 0; ls < SS_LS_MAXIT; ls++) {                     // one ls_eval site: expand while phi' < 0 at hi, then safeguarded Newton
        if (SS_M(fabs)(d1) <= tol) break;
        real nx;
        if (d1 < 0.f && al >= hi) { lo = al; hi = 2.f * al; nx = hi; }
        else {
          if (d1 < 0.f) lo = al; else hi = al;
          nx = d2 > 0.f ? al - d1 / d2 : 0.5f * (lo + hi);
          if (!(nx > lo && nx < hi)) nx = 0.5f * (lo + hi);
        }
        al = nx;
        ls_eval(al, c1, c2, d1, d2);
      }
    }
    int changed = 0;
    real dcost = 0.f;                                        // sum over this lane's rows of D [(x1)_-^2 - (x0)_-^2], x1 = x0 + al jd
#pragma unroll
    for (int p = 0; p < DOFP; p++) {
      int i = p * 64 + lane;
      if (i < h.nv) a[i] += al * delta[i];
    }
    for (int idx = lane; idx < 6 * h.nb; idx += 64) { const int b = idx / 6; Ab[idx] += al * An[8 + 2 * b + idx]; }
#pragma unroll
    for (int p = 0; p < SLOTP; p++) {
      Contact &c = con[p];
      if (!c.active) continue;
#pragma unroll
      for (int r_ = 0; r_ < 4; r_++) {
        const real x0 = c.jar[r_], st_ = al * c.jd[r_], nj = x0 + st_;
        changed |= (nj < 0.f) != (x0 < 0.f);
        dcost += c.D * row_dcost(x0, st_, nj);
        c.jar[r_] = nj;
      }
    }
#pragma unroll
    for (int p = 0; p < DOFP; p++) {
      Limit &l = lim[p];
      if (l.sign == 0.f) continue;
      const real x0 = l.jar, st_ = al * l.jd, nj = x0 + st_;
      changed |= (nj < 0.f) != (x0 < 0.f);
      dcost += l.D * row_dcost(x0, st_, nj);
      l.jar = nj;
    }
    if constexpr (SELFCOL) {
      if (lane < this->nself) {
        SelfCon &c = this->sc;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const real x0 = c.jar[i], st_ = al * c.jd[i], nj = x0 + st_;
          changed |= (nj < 0) != (x0 < 0);
          dcost += c.D * row_dcost(x0, st_, nj);
          c.jar[i] = nj;
        }
      }
    }
    dcost = w->sum(dcost);
    const real improvement = -(al * (c1 + 0.5f * c2 * al) + 0.5f * dcost);
    w->sync();
    const bool piece_min = !w->any(changed) && exact;        // full Newton step within one quadratic piece: its minimiser
#if defined(SS_TRACE_NEWTON) && !defined(__HIPCC__)
    if (lane == 0) fprintf(stderr, "NT env %d it %d dg %.4e dgabs %.4e sa %.4e sb %.4e al %.6f exact %d piece_min %d improvement %.4e tol %.3e\n", env, iters, (double)dg_, (double)dgabs, (double)s_a, (double)s_b, (double)al, (int)exact, (int)piece_min, (double)improvement, (double)k->cfg.solver_tolerance);
#endif
    return piece_min || !(improvement >= (real)k->cfg.solver_tolerance);
  }
  // one row's term of the cost change along the line, times 2 / D:  (x1)_-^2 - (x0)_-^2  with x1 = x0 + st
  SS_DEV static real row_dcost(real x0, real st, real x1) {
    const bool a0 = x0 < 0.f, a1 = x1 < 0.f;
    return a0 ? (a1 ? st * (x0 + x1) : -x0 * x0) : (a1 ? x1 * x1 : real(0));
  }

  // ------------------------------------------------------------------ controllers (torque for the NEXT mj_step)
  // `pd` (PIDController with zero integral gain, reference controllers.py:335-346) and `torque`
  // (SimpleTorqueController :45-46)
  SS_DEV void simple_controller(real abias) {
    typename HT::type h = HT::view(k->h);
    const int mode = k->cfg.control_mode;
    const real dtp = h.dt * (real)k->cfg.control_freq_inv;   // the dt SimplePID is constructed with (humanoid_env.py:319)
#pragma unroll
    for (int p = 0; p < DOFP; p++) {
      int i = p * 64 + lane;
      if (i < h.nv) {
        real t = 0.f;
        if (dc(i, 10) != 0.f) {
          const int ai = (int)dc(i, 11);
          real act = action_at(ai) + abias, lim_ = dc(i, 7);
          if (mode == SS_CTRL_DEFAULT) t = act;                // ctrl = action, unscaled and unclipped (humanoid_env.py:409-410)
          else {
            if (mode == SS_CTRL_PD) t = -dc(i, 5) * (q[i + 1] - (act * dc(i, 8) + dc(i, 9))) - dc(i, 6) * v[i];
            else if (mode == SS_CTRL_SIMPLE_PID) {             // SimplePID (controllers.py:224-262), ki = 1, state in HBM
              real *ip = gptr(k->st.pid_integral) + (size_t)env * h.nu + ai, *ep = gptr(k->st.pid_last_error) + (size_t)env * h.nu + ai;
              const real err = act * dc(i, 8) + dc(i, 9) - q[i + 1];
              const real derr = pid_on ? err - *ep : 0.f;
              real in = *ip + err * dtp;
              in = SS_M(fmin)(SS_M(fmax)(in, -lim_), lim_);
              t = dc(i, 5) * err + in + dc(i, 6) * derr / dtp;
              *ip = in; *ep = err;
            } else t = act * k->cfg.power_scale * lim_;
            t = SS_M(fmin)(SS_M(fmax)(t, -lim_), lim_);
          }
        }
        tau[i] = t;
      }
    }
    if (mode == SS_CTRL_SIMPLE_PID) pid_on = 1;
    w->sync();
  }

  // Stable PD (reference controllers.py:116-190): (M + Kd dt) qdd = -C - Kp e - Kd v on the M, C of the
  // forward pass that is in LDS (the "stale" qM / qfrc_bias) with the current q, v
  SS_DEV void spd_prepare(real abias) {
    fresh();
    typename HT::type h = HT::view(k->h);
    write_own_inertia();
    if (HT::lean(k->h)) {
      // lean models: a dof's constants are two 16-byte global reads — (invweight, kp, kd, torque limit), (scale, offset, actuated,
      // actuator) — requested for all of the lane's dofs at once, then the tracked action (one dependent read each), then the arithmetic
      // of the loop below, term for term
      float4_t c1[DOFP], c2[DOFP];
      real av[DOFP];
#pragma unroll
      for (int p = 0; p < DOFP; p++) { const int i = p * 64 + lane; c1[p] = float4_t{}; c2[p] = float4_t{}; if (i < h.nv) { c1[p] = dcq(i, 1); c2[p] = dcq(i, 2); } }
#pragma unroll
      for (int p = 0; p < DOFP; p++) av[p] = c2[p].z != 0.f ? this->action_ptr()[(int)c2[p].w] : real(0);
#pragma unroll
      for (int p = 0; p < DOFP; p++) {
        const int i = p * 64 + lane;
        perr[p] = 0.f;
        if (i < h.nv) {
          const real kp = c1[p].y, kd = c1[p].z;
          if (c2[p].z != 0.f) perr[p] = q[i + 1] + v[i] * h.dt - ((av[p] + abias) * c2[p].x + c2[p].y);
          diag[i] = armature(i) + kd * h.dt;
          delta[i] = -kp * perr[p] - kd * v[i];
        }
      }
      w->sync();
      return;
    }
#pragma unroll
    for (int p = 0; p < DOFP; p++) {
      int i = p * 64 + lane;
      perr[p] = 0.f;
      if (i < h.nv) {
        real kp = dc(i, 5), kd = dc(i, 6);
        if (dc(i, 10) != 0.f) perr[p] = q[i + 1] + v[i] * h.dt - ((action_at((int)dc(i, 11)) + abias) * dc(i, 8) + dc(i, 9));
        diag[i] = armature(i) + kd * h.dt;
        delta[i] = -kp * perr[p] - kd * v[i];                // (-C: the solve takes Fb as its per-body bias)
      }
    }
    w->sync();
  }
  SS_DEV void spd_finish() {
    fresh();
    typename HT::type h = HT::view(k->h);
    if (HT::lean(k->h)) {                                    // (as in spd_prepare: the rows first, then the loop below term for term)
      float4_t c1[DOFP], c2[DOFP];
#pragma unroll
      for (int p = 0; p < DOFP; p++) { const int i = p * 64 + lane; c1[p] = float4_t{}; c2[p] = float4_t{}; if (i < h.nv) { c1[p] = dcq(i, 1); c2[p] = dcq(i, 2); } }
#pragma unroll
      for (int p = 0; p < DOFP; p++) {
        const int i = p * 64 + lane;
        if (i < h.nv) {
          real t = 0.f;
          if (c2[p].z != 0.f) {
            const real lim_ = c1[p].w;
            t = -c1[p].y * perr[p] - c1[p].z * (v[i] + delta[i] * h.dt);
            t = SS_M(fmin)(SS_M(fmax)(t, -lim_), lim_);
          }
          tau[i] = t;
        }
      }
      w->sync();
      return;
    }
#pragma unroll
    for (int p = 0; p < DOFP; p++) {
      int i = p * 64 + lane;
      if (i < h.nv) {
        real t = 0.f;
        if (dc(i, 10) != 0.f) {
          real lim_ = dc(i, 7);
          t = -dc(i, 5) * perr[p] - dc(i, 6) * (v[i] + delta[i] * h.dt);
          t = SS_M(fmin)(SS_M(fmax)(t, -lim_), lim_);
        }
        tau[i] = t;
      }
    }
    w->sync();
  }

  // ------------------------------------------------------------------ semi-implicit Euler
  SS_DEV void integrate() {
    fresh();
    typename HT::type h = HT::view(k->h);
    const real dt = h.dt;
    if (lane == 0) {
      real wx = v[3] + dt * a[3], wy = v[4] + dt * a[4], wz = v[5] + dt * a[5];
      real nw = SS_M(sqrt)(wx * wx + wy * wy + wz * wz);
      real ang = dt * nw, ax, ay, az;
      if (nw < real(1e-15)) { ax = 1; ay = 0; az = 0; ang = 0; } else { ax = wx / nw; ay = wy / nw; az = wz / nw; }
      real sh, ch; sincos_small(0.5f * ang, &sh, &ch);
      real qw = q[3], qx = q[4], qy = q[5], qz = q[6];
      real n = SS_M(sqrt)(qw * qw + qx * qx + qy * qy + qz * qz);
      if (n < real(1e-15)) { qw = 1; qx = qy = qz = 0; } else { real in = 1.f / n; qw *= in; qx *= in; qy *= in; qz *= in; }
      real rw = ch, rx = ax * sh, ry = ay * sh, rz = az * sh;
      q[3] = qw * rw - qx * rx - qy * ry - qz * rz;
      q[4] = qw * rx + qx * rw + qy * rz - qz * ry;
      q[5] = qw * ry - qx * rz + qy * rw + qz * rx;
      q[6] = qw * rz + qx * ry - qy * rx + qz * rw;
    }
    w->sync();                                              // lane 0 read v[3:6] before lanes 3-5 update them
#pragma unroll
    for (int p = 0; p < DOFP; p++) {
      int i = p * 64 + lane;
      if (i < h.nv) {
        real vn = v[i] + dt * a[i];
        v[i] = vn;
        if (i < 3) q[i] += dt * vn;
        else if (i >= 6) q[i + 1] += dt * vn;
      }
    }
    w->sync();
  }

  SS_DEV bool any_bad(const real *x, int n) {
    int bad = 0;
    for (int i = lane; i < n; i += 64) bad |= is_bad(x[i]);
    return w->any(bad);
  }

  // mj_resetData after a bad qpos / qvel / qacc (MuJoCo autoreset)
  SS_DEV void reset_data() {
    typename HT::type h = HT::view(k->h);
    for (int i = lane; i < h.nq; i += 64) q[i] = 0.f;
    for (int i = lane; i < h.nv; i += 64) { v[i] = 0.f; a[i] = 0.f; tau[i] = 0.f; }
    w->sync();
    if (lane == 0) { q[0] = h.qpos0_root[0]; q[1] = h.qpos0_root[1]; q[2] = h.qpos0_root[2]; q[3] = 1.f; }
    nwarn_add++;
    w->sync();
  }

  // ------------------------------------------------------------------ observations (self_obs_v 1 / 2) + task tail
  SS_DEV void write_obs(real *obs, real tar, real tar_y, real tar_z) {
    fresh();
    typename HT::type h = HT::view(k->h);
    const ss_env_cfg &cf = k->cfg;
    // heading from remove_base_rot(root quat): rotated x axis = third column of the root rotation
    real hx = R[2], hy = R[5];
    real hn = SS_M(sqrt)(hx * hx + hy * hy);
    real ch = 1.f, sh = 0.f;
    if (hn > 0.f) { ch = hx / hn; sh = hy / hn; }
    int o = 0;
    if (cf.root_height_obs) { if (lane == 0) obs[0] = q[2]; o = 1; }
    const int nb = h.nb, nd = 3 * (nb - 1);
    if (lane >= 1 && lane < nb) {
      const real *rb = r + 3 * lane;
      real *dst = obs + o + 3 * (lane - 1);
      dst[0] = ch * rb[0] + sh * rb[1]; dst[1] = -sh * rb[0] + ch * rb[1]; dst[2] = rb[2];
    }
    o += nd;
    if (lane < nb) {
      const real *Rb = R + 9 * lane;
      real *dst = obs + o + 6 * lane;
      dst[0] = ch * Rb[0] + sh * Rb[3]; dst[1] = -sh * Rb[0] + ch * Rb[3]; dst[2] = Rb[6];
      dst[3] = ch * Rb[2] + sh * Rb[5]; dst[4] = -sh * Rb[2] + ch * Rb[5]; dst[5] = Rb[8];
    }
    o += 6 * nb;
    if (cf.self_obs_v == 1) {
      if (lane < 2) {
        const real *x = v + 3 * lane;
        real *dst = obs + o + 3 * lane;
        dst[0] = ch * x[0] + sh * x[1]; dst[1] = -sh * x[0] + ch * x[1]; dst[2] = x[2];
      }
      o += 6;
      for (int i = lane; i < nd; i += 64) obs[o + i] = v[6 + i];
      o += nd;
    } else {
      if (lane < nb) {
        const real *sv = gptr(k->st.body_vel) + ((size_t)env * nb + lane) * 6;   // written by this lane in forward_kin
        real *d0 = obs + o + 3 * lane, *d1 = obs + o + 3 * nb + 3 * lane;
        d0[0] = ch * sv[0] + sh * sv[1]; d0[1] = -sh * sv[0] + ch * sv[1]; d0[2] = sv[2];
        d1[0] = ch * sv[3] + sh * sv[4]; d1[1] = -sh * sv[3] + ch * sv[4]; d1[2] = sv[5];
      }
      o += 6 * nb;
    }
    if (lane == 0) {
      if (cf.task == SS_TASK_SPEED) { obs[o] = ch; obs[o + 1] = -sh; obs[o + 2] = tar; }
      else if (cf.task == SS_TASK_GETUP) obs[o] = tar;
      else if (cf.task == SS_TASK_REACH) {                  // heading^-1 (tar_pos - root_pos), humanoid_reach.py:21-30
        const real dx = tar - q[0], dy = tar_y - q[1];
        obs[o] = ch * dx + sh * dy; obs[o + 1] = -sh * dx + ch * dy; obs[o + 2] = tar_z - q[2];
      }
    }
  }
};

// ---------------------------------------------------------------------- per-env driver (all modes)
// One loop over "forward passes"; every heavy stage has exactly one call site in its body (code size:
// the stages are force-inlined, and the instruction cache is shared by the CU's wavefronts).
//   PROLOGUE  forward at the state of the previous launch's last mj_forward -> stale M, C; first torque
//   SUBSTEP   mj_step: forward, constraints, Newton, Euler; then the controller torque for the next one
//   RESETFWD  reset_sim(): mj_forward at the reset state (sensors, contacts; no solve needed)
//   FINAL     mj_kinematics on the new qpos for the observation
enum { K_PROLOGUE = 0, K_SUBSTEP = 1, K_RESETFWD = 2, K_FINAL = 3 };
enum { SOLVE_NEWTON = 1, SOLVE_SPD = 2 };

// `mode` is k->mode, or MODE_RESET for the second, fused pass of ss_step_autoreset (the caller loops: one call site).
// Returns true when the env's step ended its episode and the fused Default reset has to run next.
// BODYOUT: the instantiation whose step / reset passes also write the body frames (ss_set_body_outputs).  A separate
// instantiation because the extra epilogue costs the headline step kernel 3.7% (register allocation of the hot loops
// shifts) even when the pointer is null — callers that do not ask for it keep the plain one.
template <class W, int DOFP, int CANDP, int SLOTP, int NPASS, bool BODYOUT = false, bool SHAPED = false, class HT = HdrRuntime, bool SELFCOL = false>
SS_DEV bool run_env(W *w, const KArgs *k, const uint32_t *T, real *L, int env, int mode, real *pool = nullptr) {
  typename HT::type h = HT::view(k->h);
  const ss_env_cfg &cf = k->cfg;
  const ss_state &st = k->st;
  const bool fused_pass = mode != k->mode;                    // the in-launch reset of an env that just finished
  if (!fused_pass && k->mask && !k->mask[env]) return false;
  Sim<W, DOFP, CANDP, SLOTP, NPASS, SHAPED, HT, SELFCOL> sim;
  sim.init(w, k, T, L, env, pool);
  int lane = sim.lane;
  real *qg = gptr(st.qpos) + (size_t)env * h.nq, *vg = gptr(st.qvel) + (size_t)env * h.nv;
  real *qpg = gptr(st.qpos_prev) + (size_t)env * h.nq, *vpg = gptr(st.qvel_prev) + (size_t)env * h.nv;
  real *wg = gptr(st.qacc_warm) + (size_t)env * h.nv;
  real *tk = gptr(st.task) + (size_t)env * 4;
  const real *act = k->actions ? k->actions + (size_t)env * h.nu : nullptr;
  const real *trand_base = fused_pass ? k->task_rand2 : k->task_rand;
  const real *trand = trand_base ? trand_base + (size_t)env * 4 : nullptr;
  // (written with the mode test first: `k->fall_actions ? ...` alone costs the headline kernel 1.2 % — register allocation)
  const real *fa = (mode == MODE_RESET && k->fall_actions) ? k->fall_actions + (size_t)env * 3 * h.nu : nullptr;   // reset passes, fused ones too
  real *obs_base = fused_pass ? k->obs2 : k->obs;
  const int ostride = BODYOUT ? k->obs_stride : k->obs_size;   // rows wider than the observation: a task part follows (imitation)
  real *obs = obs_base ? obs_base + (size_t)env * ostride : nullptr;
  // step pass of a fused launch: the post-step observation also goes to obs2 (envs that do not reset keep it)
  real *obs_also = (!fused_pass && k->fused_reset && k->obs2) ? k->obs2 + (size_t)env * ostride : nullptr;
  const int maxit = cf.newton_iters > 0 ? cf.newton_iters : 100;   // mjOption.iterations

#if defined(SS_COST_KEY) && defined(__HIPCC__)
  const unsigned long long cost_t0 = __builtin_readcyclecounter();   // study build: the env's own cycles of this pass go to the truncation counter's array
#endif
  int cur_t = st.cur_t[env];
  // task scalars: speed/getup [target, change_steps, recovery, -] ; reach [tx, ty, tz, change_steps]
  const bool is_reach = cf.task == SS_TASK_REACH;
  real tar = tk[0], tar_y = tk[1], tar_z = tk[2];
  real change = is_reach ? tk[3] : tk[1], recov = is_reach ? 0.f : tk[2];
  int nsub = k->nsub;
  // StateInit.Fall draws action = U[0,1) - 0.5 (humanoid_env.py:487): the -0.5 is applied in the controller
  const real abias = (mode == MODE_RESET) ? -0.5f : 0.f;
  const bool is_debug = mode == MODE_DEBUG_FORWARD;

  // ---- task bookkeeping that precedes the physics (HumanoidTask.reset / pre_physics_step: update_task)
  const bool resample = (mode == MODE_RESET && cf.task != SS_TASK_BASE) ||
                        (mode == MODE_STEP && cf.task != SS_TASK_BASE && (real)cur_t >= change);
  if (mode == MODE_RESET && cf.task == SS_TASK_GETUP) recov = (real)cf.recovery_steps;
  if (resample) {                                            // uses the OLD cur_t on reset (reference quirk)
    const real u0 = trand ? trand[0] : 0.f, u1 = trand ? trand[1] : 0.f, u2 = trand ? trand[2] : 0.f, u3 = trand ? trand[3] : 0.f;
    if (cf.task == SS_TASK_SPEED) {
      tar = (cf.tar_speed_max - cf.tar_speed_min) * u0 + cf.tar_speed_min;
      change = (real)(cur_t + cf.speed_change_min + (int)SS_M(floor)(u1 * (real)(cf.speed_change_max - cf.speed_change_min)));
    } else if (cf.task == SS_TASK_GETUP) {
      tar = (cf.tar_height_max - cf.tar_height_min) * u0 + cf.tar_height_min;
      change = (real)(cur_t + cf.height_change_min + (int)SS_M(floor)(u1 * (real)(cf.height_change_max - cf.height_change_min)));
    } else {                                                 // reach (humanoid_reach.py:81-92)
      tar = cf.tar_dist_max * (2.f * u0 - 1.f); tar_y = cf.tar_dist_max * (2.f * u1 - 1.f);
      tar_z = (cf.tar_height_max - cf.tar_height_min) * u2 + cf.tar_height_min;
      change = (real)(cur_t + cf.height_change_min + (int)SS_M(floor)(u3 * (real)(cf.height_change_max - cf.height_change_min)));
    }
  }

  // ---- initial LDS state
  sim.load(sim.a, wg, h.nv);
  if (mode == MODE_RESET) {
    if (cf.state_init == SS_INIT_EXTERNAL) {                  // reference-state init: the caller wrote qpos / qvel
      sim.load(sim.q, qg, h.nq); sim.load(sim.v, vg, h.nv);
    } else {
      for (int i = lane; i < h.nq; i += 64) sim.q[i] = 0.f;
      for (int i = lane; i < h.nv; i += 64) sim.v[i] = 0.f;
      w->sync();
      if (lane == 0) {
        if (cf.state_init == SS_INIT_DEFAULT) { sim.q[2] = real(0.94); sim.q[3] = sim.q[4] = sim.q[5] = sim.q[6] = 0.5f; }
        else { sim.q[2] = real(0.3); sim.q[3] = 1.f; }
      }
    }
    nsub = cf.state_init == SS_INIT_FALL ? 3 * cf.control_freq_inv : 0;
  } else if (mode == MODE_KINEMATICS || is_debug) {
    sim.load(sim.q, qg, h.nq); sim.load(sim.v, vg, h.nv);
    if (is_debug) for (int i = lane; i < h.nv; i += 64) { int ai = (int)sim.dc(i, 11); sim.tau[i] = (act && ai >= 0) ? act[ai] : 0.f; }
    nsub = is_debug ? 1 : 0;
  } else {
    sim.load(sim.q, qpg, h.nq); sim.load(sim.v, vpg, h.nv);
  }
  w->sync();

  real prev_x = 0.f, prev_y = 0.f;
  const real *cached_action = nullptr;
  // pass sequence: [PROLOGUE] SUBSTEP*nsub [RESETFWD | FINAL]
  int s = (nsub > 0 && !is_debug) ? -1 : 0;
  const int last_kind = mode == MODE_RESET ? K_RESETFWD : ((mode == MODE_STEP || mode == MODE_KINEMATICS) ? K_FINAL : -1);
  SS_T0();
  for (;;) {
    // opaque(): keeps the optimizer from jump-threading the state variables (it would clone every stage per state)
    const int kind = w->opaque(s < 0 ? K_PROLOGUE : (s < nsub ? K_SUBSTEP : last_kind));
    // opaque_v(): lane-derived LDS/table addresses are recomputed per pass instead of being hoisted out of the
    // state machine and spilled to scratch for the whole launch
    lane = sim.lane = w->opaque_v(sim.lane);
    sim.drop_constraints();
    if (kind < 0) break;
    if (kind == K_SUBSTEP && !is_debug) {
      if (sim.any_bad(sim.q, h.nq) || sim.any_bad(sim.v, h.nv)) sim.reset_data();   // mj_checkPos / mj_checkVel
      if (s == nsub - 1) { sim.store(qpg, sim.q, h.nq); sim.store(vpg, sim.v, h.nv); }   // stale source of the next launch
    }
    // the env reads the sensors of the LAST forward only: write them on the last mj_step / the reset forward
    SS_TICK(PF_MISC);
    sim.forward_kin(kind != K_FINAL, kind == K_RESETFWD || (kind == K_SUBSTEP && s == nsub - 1));
    SS_TICK(PF_FWD);
    if (kind == K_FINAL) break;
    if (kind == K_SUBSTEP || kind == K_RESETFWD) {
      sim.make_self_contacts(kind == K_RESETFWD || s == nsub - 1);   // first: the pair functions are the register-hungriest part of the kernel,
      sim.make_constraints();                                          // and the floor contacts' registers are not live across them this way
    }
    SS_TICK(PF_CONS);
    if (kind == K_RESETFWD) break;
    int solve = SOLVE_SPD;
    const real *next_action = nullptr;
    if (kind == K_PROLOGUE) {
      if (mode != MODE_RESET) { sim.load(sim.q, qg, h.nq); sim.load(sim.v, vg, h.nv); w->sync(); }
      prev_x = sim.q[0]; prev_y = sim.q[1];
      next_action = (mode == MODE_RESET) ? fa : act;
    } else {
      if (is_debug) {                                        // diagnostics: dense mass matrix and bias force
        sim.dump_mass_matrix(k->out0 + (size_t)env * h.nv * h.nv);
        sim.joint_bias(sim.delta);
        sim.store(k->out1 + (size_t)env * h.nv, sim.delta, h.nv);
        sim.body_accel(sim.a, sim.Ab, sim.tmpb);              // the dump used Ab as scratch
        w->sync();
      }
      sim.newton_begin();
      SS_TICK(PF_NBEGIN);
      solve = SOLVE_NEWTON;
      if (s + 1 < nsub) next_action = (mode == MODE_RESET) ? fa + (size_t)((s + 1) / cf.control_freq_inv) * h.nu : act;
    }
    bool redo = false;
    int it = 0;
    for (;;) {                                               // solver loop: one articulated-body solve site
      solve = w->opaque(solve);
      lane = sim.lane = w->opaque_v(sim.lane);
      if (solve == SOLVE_NEWTON) { sim.newton_prepare(); SS_TICK(PF_NPREP); }
      else if (solve == SOLVE_SPD) {
        if (next_action != cached_action) {                  // the action in LDS: one HBM read per control step (per Fall-reset segment)
          if (HT::lean(k->h)) { sim.set_action_ptr(next_action); cached_action = next_action; }   // (lean models read the action where it is)
          else { sim.load(sim.actl, next_action, h.nu); cached_action = next_action; w->sync(); }
        }
        if (cf.control_mode != SS_CTRL_UHC_PD) { sim.simple_controller(abias); break; }
        sim.spd_prepare(abias);
        SS_TICK(PF_SPDPREP);
      }
      unsigned long long cmask = 0ull;                        // coupled set: the bodies of the contacts with an active row, up to the root
      if constexpr (SELFCOL)
        if (solve == SOLVE_NEWTON && sim.amask) cmask = w->bor(((sim.amask >> lane) & 1ull) ? (sim.path_mask(sim.sc.b1) | sim.path_mask(sim.sc.b2)) : 0ull);
#if !defined(__HIPCC__)
      if constexpr (SELFCOL)                                 // test hook of the emulator build: couple these bodies whatever the contacts say
        if (const char *fc = getenv("SS_EMU_FORCE_COUPLED")) if (solve == SOLVE_NEWTON) { unsigned long long pm_ = 0ull; if (lane < h.nb && ((strtoull(fc, nullptr, 16) >> lane) & 1ull)) pm_ = sim.path_mask(lane); cmask |= w->bor(pm_); }
#endif
      sim.aba_solve(sim.delta, solve == SOLVE_NEWTON ? sim.Pb : sim.Fb, cmask);
      SS_TICK(PF_FACTOR);
      if (solve == SOLVE_SPD) { sim.spd_finish(); SS_TICK(PF_SPDFIN); break; }
      const bool conv = sim.newton_finish();
      SS_TICK(PF_NFIN);
      if (!conv && ++it < maxit) continue;
      if (is_debug) break;
      // mj_checkAcc -> autoreset; the mj_step is redone from the reset state (s is not advanced), so its power row is still written,
      // and the rows of the earlier mj_steps of this control step stay (the reference's env appended them before MuJoCo reset the data)
      if (sim.any_bad(sim.a, h.nv)) { sim.reset_data(); redo = true; break; }
      sim.integrate();
      if constexpr (BODYOUT)                                 // HumanoidEnv.curr_power_usage: |qfrc_actuator * qvel| of this mj_step's torque and the new velocity
        if (k->power && mode == MODE_STEP)
          for (int i = 6 + lane; i < h.nv; i += 64) k->power[((size_t)env * nsub + s) * (h.nv - 6) + i - 6] = SS_M(fabs)(sim.tau[i] * sim.v[i]);
      SS_TICK(PF_INTEG);
      if (!next_action) break;
      solve = SOLVE_SPD;
    }
    if (!redo) s++;
    if (is_debug) break;
  }

#ifdef SS_PROFILE
  if (lane == 0 && k->prof) for (int i = 0; i < PF_COUNT; i++) w->atomic_add_u64(k->prof + i, sim.prof[i]);
#ifdef SS_PROF_ENV
  // per-env record of the control step (profile studies): total ticks >> 10 | dense-part ticks >> 10 << 16, in the truncation counter's array
  if constexpr (SELFCOL) if (lane == 0 && k->self_trunc && mode == MODE_STEP) {
    unsigned long long tt = 0, td = sim.prof[PF_SC_BASE] + sim.prof[PF_SC_COLS] + sim.prof[PF_SC_DENSE] + sim.prof[PF_SC_FINAL];
    for (int i = 0; i < 12; i++) tt += sim.prof[i];
#ifdef SS_PROF_REALTIME
    td = (__builtin_amdgcn_s_memrealtime() - sim.rt0) << 6;   // 100 MHz wall clock of the control step, in units of 160 ns
#endif
    k->self_trunc[env] = (int)((tt >> 10) & 0xFFFF) | (int)(((td >> 10) & 0x7FFF) << 16);
  }
#endif
#endif
  // body frames of the last forward: mj_kinematics readback, and (ss_set_body_outputs) a by-product of every step / reset
  if ((mode == MODE_KINEMATICS || (BODYOUT && k->out0 && !is_debug && mode != MODE_SUBSTEP)) && lane < h.nb) {
    for (int c = 0; c < 3; c++) k->out0[((size_t)env * h.nb + lane) * 3 + c] = sim.r[3 * lane + c] + sim.q[c];
    for (int c = 0; c < 9; c++) k->out1[((size_t)env * h.nb + lane) * 9 + c] = sim.R[9 * lane + c];
  }
  if (mode == MODE_KINEMATICS) return false;
  const unsigned long long touch = sim.touchmask;
#if defined(SS_COST_KEY) && defined(__HIPCC__)
  if (lane == 0 && k->self_trunc && mode == MODE_STEP) k->self_trunc[env] = (int)((__builtin_readcyclecounter() - cost_t0) >> 10);
#endif
  if (lane == 0) {
    st.touch[2 * env] = (int)(touch & 0xFFFFFFFFull); st.touch[2 * env + 1] = (int)(touch >> 32);
    if (!fused_pass) st.solver_iters[env] = sim.iters;      // the scheduling hint is the step's count, not the reset's
    if (cf.control_mode == SS_CTRL_SIMPLE_PID && st.pid_started) st.pid_started[env] = sim.pid_on;
    if (sim.nwarn_add) st.nwarn[env] += sim.nwarn_add;
  }
  if (is_debug) { sim.store(k->out2 + (size_t)env * h.nv, sim.a, h.nv); return false; }
  if (mode == MODE_RESET) { sim.store(qpg, sim.q, h.nq); sim.store(vpg, sim.v, h.nv); cur_t = 0; }
  sim.store(qg, sim.q, h.nq); sim.store(vg, sim.v, h.nv); sim.store(wg, sim.a, h.nv);
  if (mode == MODE_SUBSTEP) return false;

  // ---- post_physics_step: cur_t, observation, reward, reset flags
  if (mode == MODE_STEP) cur_t += 1;
  if (obs) sim.write_obs(obs, tar, tar_y, tar_z);
  if (obs && obs_also) {                                     // copy through the lanes that wrote it: fence, then read back
    w->mem_fence();
    for (int i = lane; i < k->obs_size; i += 64) obs_also[i] = obs[i];
  }
  int term = 0, trunc = 0;                                   // wave-uniform (all inputs are)
  {
    if (mode == MODE_STEP) {
      real rew = 0.f;
      trunc = cur_t > cf.episode_length;
      const int illegal = (touch & k->illegal_mask) != 0ull;
      if (cf.task == SS_TASK_SPEED) {
        real dtc = (real)cf.control_freq_inv * h.dt;
        real vx = (sim.q[0] - prev_x) / dtc, vy = (sim.q[1] - prev_y) / dtc;
        real err = tar - vx;
        rew = SS_M(exp)(-0.25f * (err * err + real(0.1) * vy * vy));
        term = illegal;
      } else if (cf.task == SS_TASK_GETUP) {
        real diff = tar - sim.q[2];
        rew = SS_M(exp)(-4.f * diff * diff);
        if (recov > 0.f) { recov -= 1.f; term = 0; trunc = 0; }
        else term = illegal;
      } else if (cf.task == SS_TASK_REACH) {                 // reach_reward on the reach body's world position
        const real *rb = sim.r + 3 * cf.reach_body;
        const real dx = tar - (sim.q[0] + rb[0]), dy = tar_y - (sim.q[1] + rb[1]), dz = tar_z - (sim.q[2] + rb[2]);
        rew = SS_M(exp)(-4.f * (dx * dx + dy * dy + dz * dz));
        term = illegal;
      }
      if (lane == 0) {
        if (k->reward) k->reward[env] = rew;
        if (k->terminated) k->terminated[env] = (uint8_t)term;
        if (k->truncated) k->truncated[env] = (uint8_t)trunc;
      }
    }
  }
  if (lane == 0) {
    st.cur_t[env] = cur_t;
    if (is_reach) { tk[0] = tar; tk[1] = tar_y; tk[2] = tar_z; tk[3] = change; }
    else { tk[0] = tar; tk[1] = change; tk[2] = recov; }
  }
  const bool again = !fused_pass && k->fused_reset && mode == MODE_STEP && (term || trunc);
  if (again) w->mem_fence();                                 // the fused reset pass re-reads cur_t / task state from HBM
  return again;
}

}  // namespace ss
