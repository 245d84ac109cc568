// ss_tables.h — host-side construction of the index/constant tables the wavefront kernel uses.
//
// One environment is stepped by one 64-lane wavefront.  The kernel works on "nodes": node 0 = the
// root's 3 translational dofs, node 1 = its 3 rotational dofs, node b+1 = the 3 hinges of body b,
// so dof d belongs to node d/3.  Every node is a 3-dof joint of an articulated-body recursion: node 0
// carries a massless virtual body, node b+1 carries body b.  The joint-space matrices (mass matrix,
// Newton Hessian M + J^T D J, Stable-PD matrix M + Kd dt) are never formed: systems with them are solved
// by the articulated-body sweeps of ss_kernel.h (aba_solve), level by level over this node tree.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/smplsim_hip.h"
#include "ss_hdr.h"

namespace ss {

struct HostModel {
  Hdr h{};
  Hdr h_sc{};                     // the same header on the plain (non-aliased) LDS layout: what body-body-contact batches run on (= h when h is not aliased)
  std::vector<uint32_t> shared;   // tables copied into LDS once per workgroup: integer tables, then the real-valued ones
  std::vector<real> bodyc;        // [nb][kBodyC]: off3 ipos3 mass Ibody6 invw_tr  (loaded into registers)
  std::vector<real> candc;        // [ncand][kCandC]
  std::vector<int32_t> candb;     // [ncand] body of each candidate, bit 8: capsule
  std::vector<float> actc;        // [nv][4]: kp kd tlim (per dof, 0 for unactuated), [nv][2] scale offset -> in dofc
  std::vector<int32_t> dof_act;   // [nv] actuator index of a dof or -1
  std::vector<uint8_t> legal;     // [nb]
  uint64_t illegal_mask = 0;      // bodies whose floor contact terminates the episode
  int o_real = 0;                 // word offset of the real-valued part of `shared`
  std::vector<int32_t> pairs;     // body-body candidate pairs after MuJoCo's static filters, 2 words each: b1 | b2 << 8 (normal points
                                  // b1 -> b2), and the float bits of the pair's bounding-sphere reach r1 + r2 + margin (broad phase)
  std::vector<real> geomc;        // [nb][kGeomC] geoms in their body frames (pair functions of the SELFCOL kernels)
  std::vector<int32_t> sctab;     // [nb][3] elimination-tree neighbour / joint / path mask per body (SELFCOL kernels), uploaded behind the pair table
  HdrSC sc{};
  HdrC hc{};                      // centred elimination tree of the plain solves
  double meaninertia = 0;         // mjModel.stat.meaninertia (scale of the solver's termination test)
  std::string error;
  std::string self_collision_unavailable;   // non-empty: the model cannot run with ss_env_cfg.self_collision (reason)
};

// real-valued part of the shared blob (HostModel::shared from word h.o_real on)
inline real *shared_reals(HostModel &m) { return reinterpret_cast<real *>(m.shared.data()); }   // index with Hdr::o_dofc / o_boff

inline void quat2mat(const double *q, double *m) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  double n = std::sqrt(w * w + x * x + y * y + z * z);
  w /= n; x /= n; y /= n; z /= n;
  m[0] = 1 - 2 * (y * y + z * z); m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = 1 - 2 * (x * x + z * z); m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = 1 - 2 * (x * x + y * y);
}

// mean diagonal of the joint-space inertia matrix at qpos0 (armature included) = mjModel.stat.meaninertia.  At qpos0 every body frame of
// these humanoids is aligned with the world (hinges at zero, root orientation identity), so dof (body j, axis a) sees
//     M_ii = sum over the bodies b of j's subtree of  a^T I_b a + m_b |a x (c_b - p_j)|^2   (+ armature),
// and the free joint the total mass (3 x) and the same sum about the root's position (3 x).
// (ss_model_desc has neither body quaternions nor joint axes: every model it can describe has world-aligned body frames at
// the zero pose and three hinges x, y, z per body, which is what the sums below use; the root orientation in qpos0 does not enter —
// the diagonal of M is invariant under a rigid rotation of the whole tree, the root's angular dofs being body-frame ones.)
inline double meaninertia_at_qpos0(const ss_model_desc &d) {
  const int nb = d.nbody, nv = 6 + 3 * (nb - 1);
  std::vector<double> p(3 * nb), c(3 * nb), I(9 * nb);
  for (int b = 0; b < nb; b++) {
    for (int k = 0; k < 3; k++) p[3 * b + k] = b == 0 ? d.qpos0[k] : p[3 * d.body_parent[b] + k] + d.body_pos[3 * b + k];
    for (int k = 0; k < 3; k++) c[3 * b + k] = p[3 * b + k] + d.body_ipos[3 * b + k];
    double R[9]; quat2mat(d.body_iquat + 4 * b, R);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
      double s = 0; for (int k = 0; k < 3; k++) s += R[3 * i + k] * d.body_inertia[3 * b + k] * R[3 * j + k];
      I[9 * b + 3 * i + j] = s;
    }
  }
  double trace = 0;
  for (int j = 0; j < nb; j++)
    for (int a = 0; a < 3; a++) {
      double s = d.dof_armature[j == 0 ? 3 + a : 6 + 3 * (j - 1) + a];
      for (int b = j; b < nb; b++) {
        int t = b; while (t > j) t = d.body_parent[t];
        if (t != j) continue;                                // b is not in j's subtree
        const double r[3] = {c[3 * b] - p[3 * j], c[3 * b + 1] - p[3 * j + 1], c[3 * b + 2] - p[3 * j + 2]};
        const double r2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
        s += I[9 * b + 4 * a] + d.body_mass[b] * (r2 - r[a] * r[a]);
        if (j == 0 && a == 0) trace += 3 * d.body_mass[b];   // the three translational dofs
      }
      trace += s;
    }
  for (int a = 0; a < 3; a++) trace += d.dof_armature[a];
  return trace / (nv > 1 ? nv : 1);
}

inline bool build_host_model(const ss_model_desc &d, HostModel &out) {
  Hdr &h = out.h;
  const int nb = d.nbody;
  if (nb < 1 || nb > 63) { out.error = "nbody must be in [1,63]"; return false; }
  if (d.body_parent[0] != -1) { out.error = "body 0 must be the root"; return false; }
  for (int b = 1; b < nb; b++)
    if (d.body_parent[b] < 0 || d.body_parent[b] >= b) { out.error = "parents must precede children"; return false; }
  h.nb = nb; h.nn = nb + 1; h.nv = 6 + 3 * (nb - 1); h.nq = h.nv + 1; h.nu = d.nu;
  // mjModel.stat.meaninertia; <= 0 (e.g. a caller of the pre-round-3 struct, zero-initialised): computed here from the description
  out.meaninertia = d.meaninertia > 0.0 ? d.meaninertia : meaninertia_at_qpos0(d);
  if (!(out.meaninertia > 0.0)) { out.error = "meaninertia: the inertia matrix at qpos0 has no positive diagonal"; return false; }
  const int nn = h.nn, nv = h.nv;

  // ---- node tree
  std::vector<int> nparent(nn), ndepth(nn);
  nparent[0] = -1; ndepth[0] = 0; nparent[1] = 0; ndepth[1] = 1;
  for (int b = 1; b < nb; b++) {
    int p = d.body_parent[b];
    nparent[b + 1] = p + 1; ndepth[b + 1] = ndepth[p + 1] + 1;
  }
  int nlev = 0;
  for (int n = 0; n < nn; n++) nlev = std::max(nlev, ndepth[n] + 1);
  h.nlev = nlev;
  const int CN = nlev;                                     // chain node table row width
  std::vector<int> chainnode(nn * CN, 0);
  for (int n = 0; n < nn; n++) {
    std::vector<int> anc;                                  // root -> parent
    for (int a = nparent[n]; a >= 0; a = nparent[a]) anc.insert(anc.begin(), a);
    for (size_t k = 0; k < anc.size(); k++) chainnode[n * CN + k] = anc[k];
    chainnode[n * CN + anc.size()] = n;                    // convenient: chain includes self at its depth
  }
  // bodies must come in depth-first order (the subtree of b is the index range [b, b + size): subtree_sum): b's parent is
  // b - 1 or one of its ancestors
  for (int b = 2; b < nb; b++) {
    int a = b - 1;
    while (a > 0 && a != d.body_parent[b]) a = d.body_parent[a];
    if (a != d.body_parent[b]) { out.error = "bodies must be in depth-first order"; return false; }
  }

  for (int i = 0; i < 6; i++)                              // the root solve treats the free joint as a plain 6x6 system
    if (d.dof_armature[i] != 0.0) { out.error = "armature on the free joint is not supported"; return false; }
  // ---- dof constants: arm, lo, hi, limited, invw, kp, kd, tlim, ascale, aoffset, actuated, pad
  std::vector<real> dofc(nv * kDofC, real(0));
  out.dof_act.assign(nv, -1);
  for (int i = 0; i < d.nu; i++) {
    int dof = d.actuator_dof[i];
    if (dof < 6 || dof >= nv || out.dof_act[dof] >= 0) { out.error = "bad actuator_dof"; return false; }
    out.dof_act[dof] = i;
  }
  for (int i = 0; i < nv; i++) {
    real *c = &dofc[i * kDofC];
    c[0] = (real)d.dof_armature[i];
    c[1] = (real)d.jnt_range[2 * i]; c[2] = (real)d.jnt_range[2 * i + 1];
    c[3] = (i >= 6 && d.jnt_limited[i]) ? 1.f : 0.f;
    c[4] = (real)d.dof_invweight0[i];
    int a = out.dof_act[i];
    if (a >= 0) {
      c[5] = (real)d.kp[a]; c[6] = (real)d.kd[a]; c[7] = (real)d.torque_lim[a];
      c[8] = (real)d.act_scale[a]; c[9] = (real)d.act_offset[a]; c[10] = 1.f; c[11] = (real)a;
    } else c[11] = -1.f;
  }

  // ---- body constants (registers): off3 ipos3 mass Ibody(xx xy xz yy yz zz) invw_tr
  out.bodyc.assign(nb * kBodyC, real(0));
  for (int b = 0; b < nb; b++) {
    real *c = &out.bodyc[b * kBodyC];
    for (int k = 0; k < 3; k++) { c[k] = (real)d.body_pos[3 * b + k]; c[3 + k] = (real)d.body_ipos[3 * b + k]; }
    c[6] = (real)d.body_mass[b];
    double R[9]; quat2mat(d.body_iquat + 4 * b, R);
    double I[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
      double s = 0; for (int k = 0; k < 3; k++) s += R[3 * i + k] * d.body_inertia[3 * b + k] * R[3 * j + k];
      I[3 * i + j] = s;
    }
    c[7] = (real)I[0]; c[8] = (real)I[1]; c[9] = (real)I[2]; c[10] = (real)I[4]; c[11] = (real)I[5]; c[12] = (real)I[8];
    c[13] = (real)d.body_invweight0[2 * b];
  }
  // root offset is the initial position, not a parent-frame offset
  out.bodyc[0] = out.bodyc[1] = out.bodyc[2] = 0.f;

  // ---- contact candidates: boxes first (8 corners each, 8-aligned), then capsule ends
  out.candc.clear(); out.candb.clear();
  for (int b = 0; b < nb; b++) {
    if (d.geom_type[b] != SS_GEOM_BOX) continue;
    double G[9]; quat2mat(d.geom_quat + 4 * b, G);
    for (int i = 0; i < 8; i++) {
      double v[3] = {(i & 1) ? d.geom_size[3 * b] : -d.geom_size[3 * b], (i & 2) ? d.geom_size[3 * b + 1] : -d.geom_size[3 * b + 1],
                     (i & 4) ? d.geom_size[3 * b + 2] : -d.geom_size[3 * b + 2]};
      real c[kCandC] = {0};
      for (int r = 0; r < 3; r++) {
        c[r] = (real)(G[3 * r] * v[0] + G[3 * r + 1] * v[1] + G[3 * r + 2] * v[2]);    // corner vector (body frame)
        c[3 + r] = (real)d.geom_pos[3 * b + r];                                         // box centre (body frame)
      }
      c[7] = (real)d.body_invweight0[2 * b];
      out.candc.insert(out.candc.end(), c, c + kCandC); out.candb.push_back(b);
    }
  }
  for (int b = 0; b < nb; b++) {
    if (d.geom_type[b] == SS_GEOM_BOX) continue;
    if (d.geom_type[b] != SS_GEOM_CAPSULE && d.geom_type[b] != SS_GEOM_SPHERE) { out.error = "unknown geom type"; return false; }
    const bool sphere = d.geom_type[b] == SS_GEOM_SPHERE;    // a capsule of zero half length: ONE floor contact (mjc_PlaneSphere), no tangent hint
    double G[9]; quat2mat(d.geom_quat + 4 * b, G);
    for (int s = 0; s < 2; s++) {
      double sg = s ? -1.0 : 1.0;
      real c[kCandC] = {0};
      for (int r = 0; r < 3; r++) {
        c[r] = (real)(d.geom_pos[3 * b + r] + (sphere ? 0.0 : sg * G[3 * r + 2] * d.geom_size[3 * b + 1]));   // end-sphere centre (body frame)
        c[3 + r] = sphere ? real(0) : (real)G[3 * r + 2];                                    // capsule axis (body frame): the first tangent's hint
      }
      // (the slot pairing of the solver wants two candidates per non-box body: a sphere's second one has a radius that never qualifies)
      c[6] = (sphere && s == 1) ? real(-1e30) : (real)d.geom_size[3 * b];                    // radius
      c[7] = (real)d.body_invweight0[2 * b];
      out.candc.insert(out.candc.end(), c, c + kCandC); out.candb.push_back(b | 256);
    }
  }
  h.ncand = (int)out.candb.size();
  h.nbox = 0;
  for (int b = 0; b < nb; b++) h.nbox += d.geom_type[b] == SS_GEOM_BOX;
  h.nslot = 4 * h.nbox + 2 * (nb - h.nbox);
  out.legal.assign(nb, 0);
  out.illegal_mask = 0;
  for (int b = 0; b < nb; b++) {
    out.legal[b] = d.legal_contact ? d.legal_contact[b] : 0;
    if (!out.legal[b]) out.illegal_mask |= (1ull << b);
  }

  // ---- shared blob (copied into LDS once per workgroup)
  if (nlev > 32) { out.error = "tree too deep"; return false; }
  int maxlev = 0;                                           // widest level of the elimination tree (below)
  auto &S = out.shared; S.clear();
  auto push_i = [&](const std::vector<int> &v) { int o = (int)S.size(); for (int x : v) S.push_back((uint32_t)x); return o; };
  std::vector<real> Sf;                                    // real-valued tables, appended behind the integer ones below (assembled there)
  std::vector<real> boffv;                                 // body frame offsets in the parent frame (chain walk of forward_kin)
  for (int b = 0; b < nb; b++) for (int k = 0; k < 3; k++) boffv.push_back(b == 0 ? real(0) : (real)d.body_pos[3 * b + k]);
  h.o_chainnode = push_i(chainnode);
  h.o_ndepth = push_i(ndepth);
  std::vector<int> bpar(d.body_parent, d.body_parent + nb);
  h.o_bparent = push_i(bpar);
  {
    // subtree sizes: bodies are in depth-first order, so the subtree of b is the index range [b, b + size)
    std::vector<int> subsize(nb, 1);
    for (int b = nb - 1; b >= 1; b--) subsize[d.body_parent[b]] += subsize[b];
    for (int b = 0; b < nb; b++) {
      for (int c = b + 1; c < b + subsize[b]; c++) {
        int a = c; while (a > b) a = d.body_parent[a];
        if (a != b) { out.error = "bodies must be in depth-first order"; return false; }
      }
    }
    // subtree sums in two phases (ss_kernel.h subtree_sum): bodies whose subtree has <= 8 bodies sum their index range
    // directly (one trip of 8 reads); the few large ones add their own value, the finished sums of their small children
    // and, recursively, the terms of their large children ("cover": index t < 64 = input of body t, t >= 64 = phase-1
    // output of body t - 64).
    const int kSmall = 8;
    std::vector<int> small_l, big_l, cover;
    std::vector<int> order(nb);
    for (int b = 0; b < nb; b++) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return subsize[x] > subsize[y]; });
    for (int b : order) if (subsize[b] <= kSmall) small_l.push_back(b | (subsize[b] << 8));
    std::vector<std::vector<int>> cov(nb);
    for (int b = nb - 1; b >= 0; b--) {                        // children before parents (depth-first order)
      if (subsize[b] <= kSmall) continue;
      cov[b].push_back(b);
      for (int c = b + 1; c < nb; c++) {
        if (d.body_parent[c] != b) continue;
        if (subsize[c] <= kSmall) cov[b].push_back(64 + c);
        else cov[b].insert(cov[b].end(), cov[c].begin(), cov[c].end());
      }
    }
    for (int b : order) {
      if (subsize[b] <= kSmall) continue;
      if (cov[b].size() > 255 || cover.size() > 4095) { out.error = "subtree cover table overflow"; return false; }
      big_l.push_back(b | ((int)cover.size() << 8) | ((int)cov[b].size() << 20));
      cover.insert(cover.end(), cov[b].begin(), cov[b].end());
    }
    h.n_sumsmall = (int)small_l.size(); h.n_sumbig = (int)big_l.size();
    h.o_sumsmall = push_i(small_l); h.o_sumbig = push_i(big_l); h.o_sumcover = push_i(cover);
  }
  {
    // ---- elimination tree of the articulated-body sweeps (HdrC): the body tree re-rooted at its centre.  The root minimises the
    // depth (ties: the narrower widest level, then the lower index); the widest level may not exceed 16 nodes (two passes of 8
    // lane groups)
    std::vector<std::vector<int>> adj(nb);
    for (int b = 1; b < nb; b++) { adj[b].push_back(d.body_parent[b]); adj[d.body_parent[b]].push_back(b); }
    auto bfs = [&](int c, std::vector<int> &dep, std::vector<int> &towards) {
      dep.assign(nb, -1); towards.assign(nb, -1);
      std::vector<int> q{c}; dep[c] = 0;
      for (size_t i = 0; i < q.size(); i++) for (int o : adj[q[i]]) if (dep[o] < 0) { dep[o] = dep[q[i]] + 1; towards[o] = q[i]; q.push_back(o); }
    };
    int best = 0, bestd = 1 << 30, bestw = 1 << 30;
    std::vector<int> dep, tw;
    for (int c = 0; c < nb; c++) {
      bfs(c, dep, tw);
      int dmax = 0; for (int b = 0; b < nb; b++) dmax = std::max(dmax, dep[b]);
      std::vector<int> wd(dmax + 1, 0); for (int b = 0; b < nb; b++) wd[dep[b]]++;
      int wmax = 0; for (int L = 1; L <= dmax; L++) wmax = std::max(wmax, wd[L]);
      if (wmax > 16) continue;
      if (dmax < bestd || (dmax == bestd && wmax < bestw)) { best = c; bestd = dmax; bestw = wmax; }
    }
    bfs(best, dep, tw);
    bestd = 0; for (int b = 0; b < nb; b++) bestd = std::max(bestd, dep[b]);
    HdrC &hc = out.hc;
    hc.root = best; hc.nlev = bestd; hc.pel_level = dep[0]; hc.nkpack[0] = hc.nkpack[1] = 0ull; hc.cpack = 0ull;
    if (bestd > 32) { out.error = "tree too deep"; return false; }
    // level lists: children of one node contiguous, in the order of their parents' positions; a node's children by falling height of
    // their subtrees (ties: body index) — the limbs' nodes then keep one slot from level to level and the leaves come last, which is
    // what lets the sweep towards the root keep a limb's rows in registers (hc.chain)
    std::vector<int> hgt(nb, 0);
    for (int L = bestd; L >= 1; L--) for (int b = 0; b < nb; b++) if (dep[b] == L) hgt[tw[b]] = std::max(hgt[tw[b]], hgt[b] + 1);
    std::vector<std::vector<int>> levb(bestd + 1);
    levb[0].push_back(best);
    for (int L = 1; L <= bestd; L++)
      for (int pb_ : levb[L - 1]) {
        std::vector<int> ch;
        for (int b = 0; b < nb; b++) if (dep[b] == L && tw[b] == pb_) ch.push_back(b);
        std::stable_sort(ch.begin(), ch.end(), [&](int x, int y) { return hgt[x] > hgt[y]; });
        for (int b : ch) levb[L].push_back(b);
      }
    hc.chain = 0ull; hc.neg = 0ull;
    std::vector<int> rec;
    bool cpack_unrepresentable = false;
    for (int L = 1; L <= bestd; L++) {
      const int nk = (int)levb[L].size();
      maxlev = std::max(maxlev, nk);
      hc.nkpack[(L - 1) >> 4] |= (unsigned long long)(nk - 1) << (4 * ((L - 1) & 15));
      int cmaxL = 0, slot_ = 0;
      bool chainL = L < bestd;
      for (int b : levb[L]) {
        const int e = tw[b];                                    // neighbour towards the root
        const bool kin = d.body_parent[b] == e;                 // walked along the kinematic direction: b's own joint
        const int jn = (kin ? b : e) + 1;                       // node index of the joint on this edge (its dofs are 3 jn ..)
        int cfirst = 0, cc = 0;
        if (L < bestd)
          for (int k2 = 0; k2 < (int)levb[L + 1].size(); k2++)
            if (tw[levb[L + 1][k2]] == b) { if (cc == 0) cfirst = k2; cc++; }
        rec.push_back(b | (jn << 8) | (e << 16) | ((kin ? 0 : 1) << 24) | ((b == 0 ? 1 : 0) << 25));
        if (!kin) hc.neg |= 1ull << (L - 1);
        rec.push_back(cfirst | (cc << 8));
        cmaxL = std::max(cmaxL, cc);
        if (cc > 1 || (cc == 1 && cfirst != slot_)) chainL = false;
        slot_++;
      }
      if (chainL) hc.chain |= 1ull << (L - 1);
      // the fixed-layout instantiations read this as "most children of a node of level L" (3 bits per level): a tree with a node of
      // more than 7 children below the root, or deeper than 21 levels, gets a value no instantiation is built for (runtime kernel)
      if (L <= 21 && cmaxL <= 7) hc.cpack |= (unsigned long long)cmaxL << (3 * (L - 1));
      else cpack_unrepresentable = true;
    }
    if (rec.empty()) { rec.push_back(0); rec.push_back(0); }
    if (cpack_unrepresentable) hc.cpack = ~0ull;
    hc.o_lev = push_i(rec);
    if (maxlev > 16) { out.error = "more than 16 nodes in one tree level"; return false; }
    if (maxlev < 1) maxlev = 1;
    h.maxlev = maxlev;                                        // sizes the level buffer of the LDS layout
  }
  // The SMPL-X size class (more than 32 bodies: LDS, not the register file, caps its resident envs; round 5) runs LEAN: the aliased
  // env-slice layout (ss_hdr.h make_layout) and the dof-constant table left in global memory — 12 reals per dof = 7.6 KB of the
  // workgroup's LDS for 52 bodies, read once or twice per mj_step per lane (limits, Stable PD: L1-resident), except the armature, which
  // the Newton iteration reads and which keeps a column of its own in LDS.  Together with the action read from global memory instead
  // of an LDS copy that is a seventh resident env per CU.  The LDS copy of the blob is its prefix of h.shared_words words; the dof
  // table lies behind it.
  const bool alias = nb > 32 && alias_layout_fits(nb, maxlev, h.nslot) && !std::getenv("SS_NO_ALIAS_LAYOUT");
  int lds_reals, o_arm = 0;
  if (alias) {
    h.o_boff = (int)Sf.size(); Sf.insert(Sf.end(), boffv.begin(), boffv.end());
    o_arm = (int)Sf.size(); for (int i = 0; i < nv; i++) Sf.push_back(dofc[i * kDofC]);
    while (Sf.size() % 4) Sf.push_back(real(0));
    lds_reals = (int)Sf.size();
    h.o_dofc = (int)Sf.size(); Sf.insert(Sf.end(), dofc.begin(), dofc.end());
  } else {
    h.o_dofc = (int)Sf.size(); Sf.insert(Sf.end(), dofc.begin(), dofc.end());
    h.o_boff = (int)Sf.size(); Sf.insert(Sf.end(), boffv.begin(), boffv.end());
    lds_reals = (int)Sf.size();
  }
  while (S.size() % 4) S.push_back(0u);                     // reals start 16-byte aligned
  out.o_real = (int)S.size();
  S.resize(S.size() + Sf.size() * (sizeof(real) / 4));
  std::memcpy(S.data() + out.o_real, Sf.data(), Sf.size() * sizeof(real));
  h.shared_words = out.o_real + lds_reals * (int)(sizeof(real) / 4);   // what a workgroup copies into LDS (the whole blob unless lean)
  const int real0 = out.o_real / (int)(sizeof(real) / 4);   // kernel-side offsets count reals from the start of the blob
  h.o_dofc += real0; h.o_boff += real0;
  h.arm_lean = (unsigned long long)(uint32_t)(alias ? o_arm + real0 : 0) | ((unsigned long long)(alias ? 1 : 0) << 32);

  // ---- per-env LDS layout (floats); arrays with disjoint lifetimes share storage (LDS capacity sets the number
  // of resident envs per CU)
  if (nn > 64) { out.error = "too many nodes"; return false; }
  // The SMPL-X size class (more than 32 bodies: LDS, not the register file, caps its resident envs) gets the aliased layout when its
  // records fit (ss_hdr.h make_layout); body-body-contact batches of such a model run on the plain one (h_sc), whose Aown .. (W, y)
  // stretch holds their dense system.
  auto fill = [&](Hdr &g, const Layout &y) {
    g.l_q = y.l_q; g.l_v = y.l_v; g.l_a = y.l_a; g.l_tau = y.l_tau; g.l_Fb = y.l_Fb; g.l_act = y.l_act; g.l_delta = y.l_delta; g.l_Pb = y.l_Pb; g.l_V = y.l_V;
    g.l_diag = y.l_diag; g.l_S = y.l_S; g.l_Ab = y.l_Ab; g.l_Iown = y.l_Iown; g.l_An = y.l_An; g.l_Aown = y.l_Aown; g.ia_stride = y.ia_stride;
    g.l_IA = y.l_IA; g.l_Gb = y.l_Gb; g.l_tmp = y.l_tmp; g.l_Ubuf = y.l_Ubuf; g.l_Wst = y.l_Wst; g.l_R = y.l_R; g.l_r = y.l_r;
    g.a_stride = y.a_stride; g.l_Rloc = y.l_Rloc; g.l_w2 = y.l_w2;
    g.env_floats = y.env_floats;
  };
  {
    const Layout y = make_layout(nb, maxlev, false);
    fill(h, y);
    if (13 * h.nslot > h.l_Wst - h.l_An) { out.error = "contact record buffer does not fit"; return false; }
    if (alias) fill(h, make_layout(nb, maxlev, true));
  }

  // ---- body-body collision: candidate pairs (mj_collision's static filters: contype / conaffinity masks, parent-child
  // filter, <exclude> pairs) and the geom table of the pair functions.  The pair's first geom is the one MuJoCo hands to the
  // pair function first (capsule before box, then the lower id); contact normals point from it to the second.
  out.pairs.clear();
  for (int i = 0; i < nb; i++) for (int j = i + 1; j < nb; j++) {
    const int ct1 = d.geom_contype ? d.geom_contype[i] : 1, ca1 = d.geom_conaffinity ? d.geom_conaffinity[i] : 1;
    const int ct2 = d.geom_contype ? d.geom_contype[j] : 1, ca2 = d.geom_conaffinity ? d.geom_conaffinity[j] : 1;
    if (!((ct1 & ca2) || (ct2 & ca1))) continue;
    if (d.body_parent[j] == i || d.body_parent[i] == j) continue;
    bool ex = false;
    for (int e = 0; e < d.nexclude; e++)
      ex |= (d.exclude[2 * e] == i && d.exclude[2 * e + 1] == j) || (d.exclude[2 * e] == j && d.exclude[2 * e + 1] == i);
    if (ex) continue;
    // MuJoCo hands the geom of the lower type id to the pair function first (sphere 2 < capsule 3 < box 6), the lower geom id among equals
    auto rank = [&](int b) { return d.geom_type[b] == SS_GEOM_SPHERE ? 0 : (d.geom_type[b] == SS_GEOM_CAPSULE ? 1 : 2); };
    const int first = rank(j) < rank(i) ? j : i;
    out.pairs.push_back(first | ((first == i ? j : i) << 8));
    // bounding spheres about the geom centres: half diagonal of a box, radius + half length of a capsule; a hair of slack so that
    // the float rounding of the sum can never prune a pair the pair function would report (those are strictly inside)
    auto reach = [&](int b) {
      const double *z = d.geom_size + 3 * b;
      return d.geom_type[b] == SS_GEOM_BOX ? std::sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]) : z[0] + (d.geom_type[b] == SS_GEOM_SPHERE ? 0.0 : z[1]);
    };
    const float rs = (float)((reach(i) + reach(j) + d.margin) * (1.0 + 1e-5));
    int32_t bits; std::memcpy(&bits, &rs, 4);
    out.pairs.push_back(bits);
  }
  out.geomc.assign((size_t)nb * kGeomC, real(0));
  for (int b = 0; b < nb; b++) {
    real *c = &out.geomc[(size_t)b * kGeomC];
    double G[9]; quat2mat(d.geom_quat + 4 * b, G);
    for (int k = 0; k < 3; k++) { c[k] = (real)d.geom_pos[3 * b + k]; c[3 + k] = (real)d.geom_size[3 * b + k]; }
    if (d.geom_type[b] == SS_GEOM_SPHERE) { c[4] = 0; c[5] = 0; }                                // half length 0 whatever the caller left there
    for (int k = 0; k < 9; k++) c[6 + k] = (real)G[k];
    c[15] = (real)d.geom_type[b];
  }
  {
    const Layout yp = make_layout(nb, maxlev, false);
    out.sc = make_layout_sc(nb, yp.env_floats, yp.l_An);
  }
  out.sc.npair = (int)out.pairs.size() / 2;
  // per body of the elimination tree (SELFCOL kernels copy this into the env's LDS slice): neighbour towards the root (255: it is the
  // root) | joint node << 8 | (S negated) << 16, and the mask of the bodies on its way to the root (itself included, the root not)
  {
    const HdrC &hc = out.hc;
    std::vector<int> tw(nb, 255), jn(nb, 0), ng(nb, 0);
    const uint32_t *rec = out.shared.data() + hc.o_lev;
    for (int i = 0; i < nb - 1; i++) {
      const int e0 = (int)rec[2 * i], b = e0 & 255;
      tw[b] = (e0 >> 16) & 255; jn[b] = (e0 >> 8) & 255; ng[b] = (e0 >> 24) & 1;
    }
    out.sctab.assign(3 * nb, 0);
    for (int b = 0; b < nb; b++) {
      unsigned long long pm = 0ull;
      for (int a = b; a != hc.root; a = tw[a]) pm |= 1ull << a;
      out.sctab[3 * b] = tw[b] | (jn[b] << 8) | (ng[b] << 16);
      out.sctab[3 * b + 1] = (int32_t)(uint32_t)(pm & 0xFFFFFFFFull); out.sctab[3 * b + 2] = (int32_t)(uint32_t)(pm >> 32);
    }
  }
  if (nb < 2) out.self_collision_unavailable = "a single body has no body-body contacts";
  // the dense assembly keeps 12 reals per joint between a contact's two bodies (at most 2 x levels of them) in the 6 nb reals of gc + zb
  else if (8 * out.hc.nlev > 64) out.self_collision_unavailable = "elimination tree deeper than 8 levels (one lane per (joint, row) of a contact's joints)";
  else if (12 * 2 * out.hc.nlev > out.sc.l_tab - out.sc.l_gc) out.self_collision_unavailable = "tree too deep for its body count (body-body contacts need 4 x levels <= bodies)";

  h.dt = (real)d.timestep; h.grav = (real)d.gravity; h.margin = (real)d.margin; h.mu = (real)d.friction;
  for (int k = 0; k < 5; k++) h.solimp[k] = (real)d.solimp[k];
  double dmax = d.solimp[1];
  double tc = d.solref[0] < 2 * d.timestep ? 2 * d.timestep : d.solref[0];
  h.K = (real)(1.0 / (dmax * dmax * tc * tc * d.solref[1] * d.solref[1]));
  h.B = (real)(2.0 / (dmax * tc));
  for (int k = 0; k < 3; k++) h.qpos0_root[k] = (real)d.qpos0[k];
  if (d.impratio != 1.0) { out.error = "impratio != 1 is not supported"; return false; }
  out.h_sc = h;
  fill(out.h_sc, make_layout(nb, maxlev, false));
  return true;
}

inline int obs_size(const Hdr &h, const ss_env_cfg &c) {
  int nd = 3 * (h.nb - 1);
  int n = (c.root_height_obs ? 1 : 0) + nd + (c.self_obs_v == 1 ? 6 * h.nb + 6 + nd : 12 * h.nb);
  if (c.task == SS_TASK_SPEED || c.task == SS_TASK_REACH) n += 3;
  if (c.task == SS_TASK_GETUP) n += 1;
  return n;
}

}  // namespace ss
