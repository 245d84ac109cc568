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
#define SS_DG_NOISE 1.8e-15                  /* Newton decrement vs the sum of |its terms|: 16 rounding units */
#define SS_LS_MAXIT 60
#else
namespace ss { typedef float real; }
#define SS_M(fn) fn##f
#define SS_LS_TOL_EFF SS_LS_TOL
#define SS_ROUND_REL 2e-6f
#define SS_DG_NOISE 1e-6f
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
  int nb, nn, nv, nq, nu, ncand, nlev, a_stride, nbox, nslot, maxlev;
  int l_Rloc, l_w2;                  // a_stride, l_Rloc, l_w2 and arm_lean (round 5: the aliased layout below and the lean tables of
  unsigned long long arm_lean;       // ss_tables.h; low word = o_arm, high word = lean: ONE 64-bit member — two ints in its place lower the
                                     // struct's alignment to 4, which moves the kernel arguments behind it and cost the SMPL headline kernel
                                     // 950 more v_readlane reloads, -1.6 %) sit in the slots of the pelvis-rooted level tables of rounds 1-2,
                                     // which had been kept as padding: the offsets of the fields behind them are part of
                                     // the kernels' register allocation, and closing the gaps cost 0.7 % on the headline (same-box A/B,
                                     // profiles/r03_centred_elimination.md)
  // shared-blob word offsets
  // (o_dofc, o_boff: real-valued tables, offsets in reals from the start of the blob; the layout of this struct is part of
  // the kernel's register allocation — one more field here cost 850 SGPR reloads in the step kernel)
  int o_dofc, o_boff, o_chainnode, o_ndepth, l_act, o_bparent, o_sumsmall, o_sumbig, o_sumcover, n_sumsmall, n_sumbig, shared_words;
  // per-env LDS float offsets.  Z = solver region: Aown | IA (the level buffer) | Ubuf | Wst ; aliases: contact records
  // at Z, R/r inside Wst, Gb and the body_accel scratch inside IA, Ad = Ubuf = An, V = Pb.  (l_act sits in the slot of a table
  // offset of rounds 1-2, ia_stride is 0 since round 4: see reserved1)
  int l_q, l_v, l_a, l_tau, l_Fb, l_Pb, l_delta, l_diag, l_S, l_Ab, l_An, l_Aown, l_IA, l_Ubuf, l_Wst,
      l_R, l_r, l_Gb, l_tmp, l_V, l_Iown, ia_stride, env_floats;
  real dt, grav, margin, mu, solimp[5], K, B;   // K, B of aref (from solref, dmax)
  real qpos0_root[3];
};

// Per-env LDS layout (float offsets) as a function of the body count and the widest tree level.  constexpr: the host fills
// Hdr with it, and the kernel instantiations specialised for one model size (HdrFixed below) fold it into instruction
// immediates.  Arrays with disjoint lifetimes share storage (LDS capacity sets the number of resident envs per CU).
struct Layout {
  int l_q, l_v, l_a, l_tau, l_Fb, l_Pb, l_delta, l_diag, l_S, l_Ab, l_An, l_Aown, l_IA, l_Ubuf, l_Wst, l_R, l_r, l_Gb, l_tmp, l_V, l_Iown,
      ia_stride, l_act, env_floats, a_stride, l_Rloc, l_w2;
};
// alias_w (round 5; the SMPL-X size class, where LDS and not the register file caps the resident envs: 5 -> 6 per CU, 7 together with the
// lean tables of ss_tables.h, which models on this layout also get): the (W, y) rows of
// a body take the place of its generalized inertia.  A body's Aown row is read in part 1 of ITS level of the sweep towards the root and
// never again in that solve (the next solve rebuilds it from Iown), its (W, y) is written in part 2 of the same level and read on the way
// back — one slot of 24 reals per body serves both (a_stride 24 instead of the packed 21), and the 24 nn reals of the separate (W, y)
// region go away.  What else lived in that region moves into the stretch An | Aown | IA, which is idle outside the solves:
//   forward pass      An: node terms (Ad) | Aown: local rotations (+ per-env body offsets) 12 nb, w2 6 nn (tmp later over the rotations) | ... | R, r
//   constraints       An .. : the contact records (13 per slot, checked against l_R by ss_tables.h) | ... | R, r (the last 12 nb of the stretch)
//   solves            An: accelerations / U rows | Aown = (W, y), 24 per body | IA: the level buffer (48 per node of the widest level)
constexpr Layout make_layout(int nb, int maxlev, bool alias_w = false) {
  const int nn = nb + 1, nv = 6 + 3 * (nb - 1);
  Layout y{};
  int o = 0;
  auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
  y.l_q = take(nv + 1); y.l_v = take(nv); y.l_a = take(nv); y.l_tau = take(nv);
  y.l_Fb = take(6 * nb);                                   // per-body bias force of the forward pass (gravity, velocity products): qfrc_bias = sum_b J_b^T Fb_b
  y.l_act = alias_w ? o : take(nv - 6);                    // the action the controller is tracking (one HBM read per control step, not per mj_step; the lean layout reads it from global memory)
  y.l_delta = take(nv);
  y.l_Pb = take(6 * nb);                                   // per-body force I a - f of the Newton iterate (bias of the sweeps)
  y.l_V = y.l_Pb;                                          // V (body velocities): dead after make_constraints
  y.l_diag = take(nv);
  y.l_S = take(6 * nv);
  y.l_Ab = take(6 * nb);
  y.l_Iown = take(10 * nb);                                // own spatial inertia of every body (10 parameters)
  y.l_An = take(8 * nn);                                   // node accelerations of the last solve; Ad (6 nb) aliases it
  // solver region Z
  if (alias_w) {
    y.a_stride = 24;
    y.l_Aown = take(24 * nb);                              // I_b + K_b packed symmetric (21) in a slot of 24: the body's (W, y) rows overwrite it
    y.ia_stride = 0;
    const int gb = (6 * nb + 3) & ~3, t6 = (6 * nn + 3) & ~3;
    y.l_IA = take(48 * maxlev);
    y.l_Gb = y.l_IA;                                       // (subtree sums: diagnostics only — ss_debug_forward — never while the level buffer is live)
    y.l_Rloc = y.l_Aown; y.l_tmp = y.l_Aown;               // local rotations (9 nb) + per-env body offsets (3 nb); tmp over them (a sync apart)
    y.l_w2 = y.l_Aown + 12 * nb;
    y.l_Ubuf = 24 * maxlev <= 8 * nn ? y.l_An : take(24 * maxlev);
    y.l_Wst = y.l_Aown;
    y.l_R = o - 12 * nb; y.l_r = y.l_R + 9 * nb;           // the end of the stretch: behind the contact records, over the level buffer
    // (12 nb + 6 nn <= 24 nb and gb <= 48 maxlev hold for every tree ss_tables.h asks this layout for; it checks them)
    (void)gb; (void)t6;
    y.env_floats = o;
    return y;
  }
  y.a_stride = 21;
  y.l_Aown = take(21 * nb);                                // per-body generalized inertia I_b + K_b, packed symmetric
  // ONE level buffer: a level's part 1 has consumed its children's rows before its part 2 writes the level's own (one wavefront per
  // env: its LDS operations retire in program order), so the rows handed towards the root overwrite the ones they were built from
  y.ia_stride = 0;
  const int gb = (6 * nb + 3) & ~3;
  const int ia_need = 48 * maxlev > gb + 6 * nn ? 48 * maxlev : gb + 6 * nn;
  y.l_IA = take(ia_need);                                  // articulated rows of the level in flight
  y.l_Gb = y.l_IA; y.l_tmp = y.l_IA + gb;                  // subtree sums and body_accel scratch live outside solves
  // U rows of the level in flight: only live in the upward sweep, An only from the downward sweep on
  y.l_Ubuf = 24 * maxlev <= 8 * nn ? y.l_An : take(24 * maxlev);
  y.l_Wst = take(24 * nn);                                 // (W_r, y_r) per node row, kept for the downward sweep
  y.l_R = y.l_Wst; y.l_r = y.l_Wst + 9 * nb;               // R, r: forward kinematics .. constraints / observations
  y.l_Rloc = y.l_IA; y.l_w2 = y.l_Wst + 12 * nb;           // local rotations of the kinematics in the level buffer; w2: the free part of (W, y) behind R, r
  y.env_floats = o;
  return y;
}
// can a tree use the aliased layout?  (the forward pass's scratch must fit the Aown region, the contact records must end before R)
constexpr bool alias_layout_fits(int nb, int maxlev, int nslot) {
  const int nn = nb + 1;
  const Layout y = make_layout(nb, maxlev, true);
  return ((6 * nb + 3) & ~3) <= 48 * maxlev && y.l_w2 + ((6 * nn + 3) & ~3) <= y.l_R && 13 * nslot <= y.l_R - y.l_An;
}

// Body-body contacts (SELFCOL instantiations, ss_env_cfg.self_collision): extra per-env LDS arrays behind the base layout and the
// static pair table.  Kept out of Hdr: Hdr's layout is part of the plain kernels' register allocation.
//
// Solver (round 4; rounds 2-3 used a Woodbury correction on top of the tree solve, whose (3c)^2 Delassus block capped the contact
// count at 8): the bodies of the contacts with an active row and their neighbours towards the root of the elimination tree form the
// COUPLED SET, a subtree that contains the root.  Everything outside it is eliminated by the articulated-body recursion as in the
// plain solve; the coupled joints stay unknowns of one dense system (3x3 blocks, CRBA on the articulated inertias handed up by the
// eliminated subtrees, plus the two-body rows' terms) that the wave factorizes in LDS.  No cap on the contact count beyond one
// contact per lane; the dense system is at most (bodies + 1) block rows.
constexpr int kMaxSelf = SS_MAX_SELF_CONTACTS;   // body-body contacts kept per env: one per lane of the wavefront
constexpr int kSelfRec = 24;                     // floats per contact record of ss_debug_self_contacts: b1 b2 | pos3 | n3 | t13 | D | aref4 | jar4 | jd4 | pad
constexpr int kSelfCand = 64;                    // narrow-phase candidates (kCandRec reals each); every candidate becomes a contact, the 65th is dropped
constexpr int kCandRec = 8;                      // pos3 n3 dist (b1 | b2 << 8 as a real)
constexpr int kGeomC = 16;                       // floats per body in the geom table: gpos3 gsize3 gmat9 (row-major) type
constexpr int kDenseMaxRows = 64;                // block rows of the dense system at most (one block row per lane in its back substitution)
struct HdrSC {
  int npair;                                     // candidate body pairs of the model (b1 | b2 << 8 each)
  int l_H, l_g, l_list, l_Wst2, hloc, o_sctab, l_gc, l_tab, nmax, l_cand, l_zb;   // float offsets in the env slice (o_sctab: ints from k->pairs)
  //   l_H     dense system (round 6): the lower triangle by TILE ROWS of 16 matrix rows (the M and N of the matrix instruction), tile row ti =
  //           16 rows of 16 (ti + 1) reals, row-major, at 256 ti (ti + 1) / 2; over the scalar unknowns 3 rank(body) + axis, the root body's six
  //           behind them and the right-hand side as one more ROW (index N = 3 nc + 6), the last tile row cut behind it (dense_floats).  It lies
  //           over the base layout's An, Aown, IA and (W, y) regions (all dead between the sweep towards the root and the one away from it) + kDenseExtra: hloc
  //           reals.  A coupled set that needs more (SMPL: more than 12 bodies; rare) takes the workgroup's ONE shared block behind the env
  //           slices instead (ss_pool_floats; a lock word in front of it) — sizing every env for the worst case would cost a resident env per CU
  //   l_g     pivots d of the factorization, then the solution, by unknown: 16 floats per tile row of the largest system
  //   l_list  coupled bodies by rank (body | joint node << 8); pair list of the broad phase (64 words)
  //   l_Wst2  (W, y) per body of the articulated-body sweeps: the base layout's slot is part of H.  Outside the solves the slot holds
  //           the narrow phase's candidates (l_cand) and the wrenches of the active contacts on their way into the per-body forces
  //   l_tab   per body: neighbour towards the root | joint node << 8 | S negated << 16 ; path mask (bodies from it up to, not including, the root), 2 words
  //   l_zb    solution of the coupled joints by body, 3 floats each (read by the sweep away from the root)
  //   nmax    block rows of the largest dense system (bodies + 1: all joints coupled, + 2 for the root body's six unknowns)
  int env_floats;                                // slice size of a SELFCOL env
};
constexpr int kDenseTile = 16;                   // the tile edge = the M and N of v_mfma_f32_16x16x4_f32
constexpr int dense_tile_rows(int nblock) { return (3 * nblock + 1 + kDenseTile - 1) / kDenseTile; }   // tile rows of a system of nblock 3x3 block rows (+ the right-hand-side row)
constexpr int dense_floats(int np) {                // reals of a system of np matrix rows (right-hand side included): whole tile rows + the cut one
  const int t = (np - 1) / kDenseTile;
  return kDenseTile * kDenseTile * (t * (t + 1) / 2) + kDenseTile * (t + 1) * (np - kDenseTile * t);
}
constexpr int kDenseExtra = 256;                 // reals appended to the dead stretch for the dense system (what 8 resident SMPL envs leave of the CU's 160 KiB)
constexpr HdrSC make_layout_sc(int nb, int base_floats, int l_An) {
  HdrSC y{};
  int o = base_floats;
  auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
  y.nmax = nb + 1 < kDenseMaxRows ? nb + 1 : kDenseMaxRows;
  // An (the solve's output, written when the dense system is done) | Aown | IA | (W, y) of the base layout, and kDenseExtra reals behind them
  y.l_H = l_An;
  take(kDenseExtra);
  y.hloc = o - y.l_H;
  y.l_g = take(kDenseTile * dense_tile_rows(y.nmax)); y.l_list = take(64);
  const int w2 = 24 * (nb + 1) > kCandRec * kSelfCand ? 24 * (nb + 1) : kCandRec * kSelfCand;
  y.l_Wst2 = take(w2); y.l_cand = y.l_Wst2;
  y.l_gc = take(3 * nb); y.l_zb = take(3 * nb); y.l_tab = take(3 * nb);
  y.env_floats = o;
  return y;
}
// reals of the workgroup's shared dense block behind the env slices (0: every coupled set fits the envs' own regions); [0] is the lock word
constexpr int ss_pool_floats(const HdrSC &y) { return dense_floats(3 * y.nmax + 1) > y.hloc ? 4 + ((dense_floats(3 * y.nmax + 1) + 3) & ~3) : 0; }

// Elimination tree of the articulated-body solves: the body tree re-rooted at its CENTRE.  H x = b is a free-floating tree's system,
// any body can carry the six free unknowns; eliminating towards the centre instead of towards the pelvis makes the sweeps as deep as
// the tree's radius, not its height (SMPL: 6 levels instead of 8, SMPL-X: 7 instead of 10).  An edge walked against the kinematic
// direction uses the same joint with S -> -S; the free joint becomes a bias force on body 0.  Built by ss_tables.h; a member of KArgs
// rather than of Hdr (Hdr's layout is part of the kernels' register allocation).
struct HdrC {
  int nlev;                    // levels below the root (level 1 = the root's neighbours)
  int o_lev;                   // word offset of the level records in the shared blob, 2 words per node, level 1 first:
                               //   word 0: body | joint node << 8 | neighbour towards the root << 16 | (S negated) << 24 | (body 0) << 25
                               //   word 1: first child's position in the next level | child count << 8
  int root;                    // root body
  int pel_level;               // level of body 0 (0 = it is the root)
  unsigned long long nkpack[2];   // (nodes in level L) - 1, 4 bits per level, level L at bit 4 (L - 1)
  unsigned long long cpack;       // most children of a node of level L (clamped to 7), 3 bits per level, level L at bit 3 (L - 1)
  unsigned long long chain;       // bit L - 1: every node of level L has at most one child, in its own slot of level L + 1 (rows stay in registers)
  unsigned long long neg;         // bit L - 1: level L holds a node whose edge is walked against the kinematic direction (S negated)
};

// The header as the kernel sees it.  HdrRuntime: the Hdr itself (any model that fits a variant).  HdrFixedT<NB, MAXLEV, tree...>: body
// count, dof counts and the whole LDS layout are compile-time constants — LDS addresses become one base register plus an
// instruction immediate and ~30 wave-uniform values leave the SGPR file (the generic kernel reloads spilled SGPRs with
// ~1300 v_readlane per instantiation) — everything else is still read from the runtime header.
#if defined(__HIPCC__)
#define SS_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define SS_HD inline __attribute__((always_inline))
#endif
SS_HD int hdr_o_arm(const Hdr &h) { return (int)(h.arm_lean & 0xffffffffull); }
SS_HD bool hdr_lean(const Hdr &h) { return (h.arm_lean >> 32) != 0ull; }
struct HdrRuntime {
  typedef const Hdr &type;
  static constexpr bool fixed = false;
  static SS_HD bool lean(const Hdr &h) { return hdr_lean(h); }
  static constexpr bool maybe_lean = true;
  static SS_HD type view(const Hdr &h) { return h; }
  typedef const HdrC &tree_type;
  static SS_HD tree_type tree(const HdrC &c) { return c; }
};
// The elimination tree's shape as compile-time constants (fixed-layout instantiations): the level bounds fold into immediates and the
// level loops of the sweeps have constant trip counts (+1.3 % on the SMPL headline: profiles/r03_centred_elimination.md 15); only
// the offset of the level records stays a runtime value.
template <int NLEV, int ROOT, int PEL, unsigned long long NK0, unsigned long long CP, unsigned long long CH, unsigned long long NG>
struct TreeFixed {
  static constexpr int nlev = NLEV, root = ROOT, pel_level = PEL;
  static constexpr unsigned long long nkpack[2] = {NK0, 0ull};
  static constexpr unsigned long long cpack = CP, chain = CH, neg = NG;
  int o_lev;
};
template <int NB, int MAXLEV, bool ALIAS = false>
struct HdrFixed {
  static constexpr Layout LY = make_layout(NB, MAXLEV, ALIAS);
  static constexpr int nb = NB, nn = NB + 1, nv = 6 + 3 * (NB - 1), nq = 7 + 3 * (NB - 1), maxlev = MAXLEV;
  static constexpr int l_q = LY.l_q, l_v = LY.l_v, l_a = LY.l_a, l_tau = LY.l_tau, l_Fb = LY.l_Fb, l_Pb = LY.l_Pb, l_delta = LY.l_delta,
                       l_diag = LY.l_diag, l_S = LY.l_S, l_Ab = LY.l_Ab, l_An = LY.l_An, l_Aown = LY.l_Aown, l_IA = LY.l_IA,
                       l_Ubuf = LY.l_Ubuf, l_Wst = LY.l_Wst, l_R = LY.l_R, l_r = LY.l_r, l_Gb = LY.l_Gb, l_tmp = LY.l_tmp, l_V = LY.l_V,
                       l_Iown = LY.l_Iown, ia_stride = LY.ia_stride, l_act = LY.l_act, env_floats = LY.env_floats,
                       a_stride = LY.a_stride, l_Rloc = LY.l_Rloc, l_w2 = LY.l_w2;
  const int &nu, &ncand, &nlev, &nbox, &nslot;
  const int &o_dofc, &o_boff, &o_chainnode, &o_ndepth, &o_bparent, &o_sumsmall, &o_sumbig, &o_sumcover, &n_sumsmall, &n_sumbig,
      &shared_words;
  const real &dt, &grav, &margin, &mu;
  const real (&solimp)[5];
  const real &K, &B;
  const real (&qpos0_root)[3];
  SS_HD explicit HdrFixed(const Hdr &h)
      : nu(h.nu), ncand(h.ncand), nlev(h.nlev), nbox(h.nbox), nslot(h.nslot), o_dofc(h.o_dofc),
        o_boff(h.o_boff), o_chainnode(h.o_chainnode), o_ndepth(h.o_ndepth), o_bparent(h.o_bparent),
        o_sumsmall(h.o_sumsmall), o_sumbig(h.o_sumbig), o_sumcover(h.o_sumcover), n_sumsmall(h.n_sumsmall), n_sumbig(h.n_sumbig),
        shared_words(h.shared_words), dt(h.dt), grav(h.grav), margin(h.margin), mu(h.mu), solimp(h.solimp), K(h.K), B(h.B),
        qpos0_root(h.qpos0_root) {}
};
template <int NB, int MAXLEV, int NLEV, int ROOT, int PEL, unsigned long long NK0, unsigned long long CP, unsigned long long CH, unsigned long long NG, bool ALIAS = false, bool LEAN = ALIAS>
struct HdrFixedT {
  typedef const HdrFixed<NB, MAXLEV, ALIAS> type;
  static constexpr bool fixed = true;
  static SS_HD constexpr bool lean(const Hdr &) { return LEAN; }    // (lean tables belong to the model, the aliased layout to its plain batches: ss_tables.h)
  static constexpr bool maybe_lean = LEAN;
  static SS_HD HdrFixed<NB, MAXLEV, ALIAS> view(const Hdr &h) { return HdrFixed<NB, MAXLEV, ALIAS>(h); }
  typedef const TreeFixed<NLEV, ROOT, PEL, NK0, CP, CH, NG> tree_type;
  static SS_HD TreeFixed<NLEV, ROOT, PEL, NK0, CP, CH, NG> tree(const HdrC &c) { TreeFixed<NLEV, ROOT, PEL, NK0, CP, CH, NG> t; t.o_lev = c.o_lev; return t; }
  static bool matches(const Hdr &h, const HdrC &c) {
    return h.nb == NB && h.maxlev == MAXLEV && h.a_stride == (ALIAS ? 24 : 21) && hdr_lean(h) == LEAN && h.env_floats == HdrFixed<NB, MAXLEV, ALIAS>::env_floats && c.nlev == NLEV && c.root == ROOT && c.pel_level == PEL && c.nkpack[0] == NK0 && c.nkpack[1] == 0ull && c.cpack == CP && c.chain == CH && c.neg == NG;
  }
};


// The two packaged fixtures' instantiations, defined once for the GPU launcher (ss_env_kernel.h) and the emulator (tests/wave_emu):
// (body count, widest level | elimination tree: levels, root body, level of body 0, packed level widths - 1, packed most-children per level,
//  mask of the 1:1 levels, mask of the levels with a negated motion subspace)
typedef HdrFixedT<24, 5, 6, 10, 2, 0x333431ull, 0x1253ull, 0x1cull, 0x3ull> HdrSmplFixed;       // SMPL: 24 bodies; rooted at the Spine: levels of 2 4 5 4 4 4 nodes
typedef HdrFixedT<52, 12, 7, 11, 3, 0xbbb3233ull, 0x9a89ull, 0x33ull, 0x7ull, true> HdrSmplxFixed;    // SMPL-X/H: 52 bodies; rooted at the Chest: levels of 4 4 3 4 12 12 12 nodes; aliased (W, y)
typedef HdrFixedT<52, 12, 7, 11, 3, 0xbbb3233ull, 0x9a89ull, 0x33ull, 0x7ull, false, true> HdrSmplxFixedSC;  // the same tree on the plain layout: body-body-contact batches (their dense system lives in Aown .. (W, y))

// compiled kernel variants: 0 = SMPL-sized (<= 128 dofs / candidates, <= 64 contact slots, <= 8 nodes per tree level),
// 1 = SMPL-X/H-sized (<= 192 dofs / candidates, <= 128 slots, <= 16 nodes per level); -1 = none fits
inline int kernel_variant(const Hdr &h) {
  const int dofp = (h.nv + 63) / 64, candp = (h.ncand + 63) / 64, slotp = (h.nslot + 63) / 64, npass = (h.maxlev + 7) / 8;
  if (dofp <= 2 && candp <= 2 && slotp <= 1 && npass <= 1) return 0;
  if (dofp <= 3 && candp <= 3 && slotp <= 2 && npass <= 2) return 1;
  return -1;
}

template <class H> constexpr int shape_stride(const H &h) { return h.nb * kBodyC + ((h.nv + 3) & ~3); }   // floats per shape in the shaped bodyc array

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
  int32_t *work_counter;      // device word, zero at launch (the previous launch cleared it): persistent waves pull env ids from it
  real *obs, *reward;
  // fused autoreset (ss_step_autoreset): envs whose step ends an episode run the Default reset in the same launch;
  // obs2 receives the observation AFTER the (possible) reset for every env, task_rand2 feeds the reset's reset_task
  int fused_reset;
  real *obs2;
  const real *task_rand2;
  uint8_t *terminated, *truncated;
  real *out0, *out1, *out2;  // kinematics: xpos, xmat ; debug forward: M [N,nv,nv], bias [N,nv], qacc [N,nv]
  // body-body contacts (SELFCOL instantiations only; appended so that the plain kernels' argument layout is what it was)
  HdrSC sc;
  const int32_t *pairs;       // [sc.npair][2] b1 | b2 << 8, float bits of the pair's bounding-sphere reach (per shape like geomc)
  const real *geomc;          // [nb][kGeomC]
  real *dbg_self;             // optional [N][kMaxSelf][kSelfRec]: the contact records of the last forward (ss_debug_self_contacts)
  // per-env body shapes (ss_model_create_shapes): bodyc holds num_shapes consecutive blocks of [nb][kBodyC] body constants
  // followed by [nv] dof inverse weights (block stride shape_stride(h) floats), candc num_shapes consecutive tables;
  // st.shape_id [N] selects per env (null = single-shape model)
  int obs_stride;             // floats between the observation rows of obs / obs2 (body-output instantiations; = obs_size otherwise)
  const void *im;             // ss::mo::ImFused on the device (ss_imitation_step_fused), or null
  const float *im_rand;       // [N,2] uniform draws for the re-initialisation of finished envs, or null = none
  int32_t *work_counter_next; // the counter of the NEXT launch on this batch: zeroed by this one (no memset between launches)
  real *power;                // optional [N, nsub, nv - 6]: |torque * velocity| per mj_step (ss_set_power_output; body-output instantiations)
  int32_t *self_trunc;        // optional [N]: += 1 per mj_step whose body-body contact list was cut to kMaxSelf (ss_debug_self_truncation)
  HdrC hc;                    // elimination tree of aba_solve / aba_resolve / aba_columns
};

// floats of one env's LDS slice for this launch
inline int env_slice_floats(const KArgs &k) { return k.cfg.self_collision ? k.sc.env_floats : k.h.env_floats; }
// bytes of dynamic LDS of a launch with e envs per workgroup
inline size_t launch_lds_bytes(const KArgs &k, int e) {
  return (size_t)((k.h.shared_words + 3) & ~3) * 4 + ((size_t)e * env_slice_floats(k) + (k.cfg.self_collision ? ss_pool_floats(k.sc) : 0)) * sizeof(real);
}

}  // namespace ss
