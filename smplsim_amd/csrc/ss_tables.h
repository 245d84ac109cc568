// ss_tables.h — host-side construction of the index/constant tables the wavefront kernel uses.
//
// One environment is stepped by one 64-lane wavefront.  The kernel works on "nodes": node 0 = the
// root's 3 translational dofs, node 1 = its 3 rotational dofs, node b+1 = the 3 hinges of body b,
// so dof d belongs to node d/3 and every node is a 3x3 block of the joint-space matrices.  The
// tree-sparse matrix H (mass matrix / Newton Hessian) is stored as chain-dense block rows:
// node n at depth L has 3 rows of W = 3L+3 floats (columns = the dofs along its ancestor chain in
// root-to-node order, then its own 3), which is exactly the sparsity pattern of M (Featherstone's
// branch-induced sparsity; SURVEY.md §8a "nnz(M)").
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/smplsim_hip.h"
#include "ss_hdr.h"

namespace ss {

struct HostModel {
  Hdr h{};
  std::vector<uint32_t> shared;   // tables copied into LDS once per workgroup
  std::vector<float> bodyc;       // [nb][kBodyC]: off3 ipos3 mass Ibody6 invw_tr  (loaded into registers)
  std::vector<float> candc;       // [ncand][kCandC]
  std::vector<int32_t> candb;     // [ncand] body of each candidate, bit 8: capsule
  std::vector<float> actc;        // [nv][4]: kp kd tlim (per dof, 0 for unactuated), [nv][2] scale offset -> in dofc
  std::vector<int32_t> dof_act;   // [nv] actuator index of a dof or -1
  std::vector<uint8_t> legal;     // [nb]
  std::vector<int> decode;        // [ne] (row_dof << 16 | col_dof) of every stored H entry (host-side, diagnostics)
  uint64_t illegal_mask = 0;      // bodies whose floor contact terminates the episode
  std::string error;
};

inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

inline void quat2mat(const double *q, double *m) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  double n = std::sqrt(w * w + x * x + y * y + z * z);
  w /= n; x /= n; y /= n; z /= n;
  m[0] = 1 - 2 * (y * y + z * z); m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = 1 - 2 * (x * x + z * z); m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = 1 - 2 * (x * x + y * y);
}

inline bool build_host_model(const ss_model_desc &d, HostModel &out) {
  Hdr &h = out.h;
  const int nb = d.nbody;
  if (nb < 1 || nb > 63) { out.error = "nbody must be in [1,63]"; return false; }
  if (d.body_parent[0] != -1) { out.error = "body 0 must be the root"; return false; }
  for (int b = 1; b < nb; b++)
    if (d.body_parent[b] < 0 || d.body_parent[b] >= b) { out.error = "parents must precede children"; return false; }
  h.nb = nb; h.nn = nb + 1; h.nv = 6 + 3 * (nb - 1); h.nq = h.nv + 1; h.nu = d.nu;
  const int nn = h.nn, nv = h.nv;

  // ---- node tree
  std::vector<int> nparent(nn), ndepth(nn), bdepth(nb);
  nparent[0] = -1; ndepth[0] = 0; nparent[1] = 0; ndepth[1] = 1; bdepth[0] = 0;
  for (int b = 1; b < nb; b++) {
    int p = d.body_parent[b];
    nparent[b + 1] = p + 1; ndepth[b + 1] = ndepth[p + 1] + 1; bdepth[b] = bdepth[p] + 1;
  }
  int nlev = 0, nblev = 0;
  for (int n = 0; n < nn; n++) nlev = std::max(nlev, ndepth[n] + 1);
  for (int b = 0; b < nb; b++) nblev = std::max(nblev, bdepth[b] + 1);
  h.nlev = nlev; h.nblev = nblev; h.maxD = 3 * (nlev - 1);
  // H storage: node n at depth d owns d+1 blocks (J = 0..d-1 couple to its ancestors, J = d is the diagonal
  // block), each 3 rows x 4 floats (one 16-byte row per ds_read_b128; the 4th float is padding)
  std::vector<int> nbase(nn);
  int ne = 0;
  for (int n = 0; n < nn; n++) { nbase[n] = ne; ne += 12 * (ndepth[n] + 1); }
  h.ne = ne;
  const int CW = h.maxD > 0 ? h.maxD : 1;                  // chain table row width (in dofs)
  const int CN = nlev;                                     // chain node table row width
  std::vector<int> chainnode(nn * CN, 0), chainrow(nn * CW, 0);
  for (int n = 0; n < nn; n++) {
    std::vector<int> anc;                                  // root -> parent
    for (int a = nparent[n]; a >= 0; a = nparent[a]) anc.insert(anc.begin(), a);
    for (size_t k = 0; k < anc.size(); k++) chainnode[n * CN + k] = anc[k];
    chainnode[n * CN + anc.size()] = n;                    // convenient: chain includes self at its depth
  }
  std::vector<int> decode(ne, -1);                         // (row_dof << 16 | col_dof) per stored float, -1 = padding
  for (int n = 0; n < nn; n++)
    for (int J = 0; J <= ndepth[n]; J++)
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
          decode[nbase[n] + 12 * J + 4 * r + c] = ((3 * n + r) << 16) | (3 * chainnode[n * CN + J] + c);
  std::vector<int> levstart(nlev + 1, 0), levnodes, blevstart(nblev + 1, 0), blevbodies;
  for (int L = 0; L < nlev; L++) {
    levstart[L] = (int)levnodes.size();
    for (int n = 0; n < nn; n++) if (ndepth[n] == L) levnodes.push_back(n);
  }
  levstart[nlev] = (int)levnodes.size();
  for (int L = 0; L < nblev; L++) {
    blevstart[L] = (int)blevbodies.size();
    for (int b = 0; b < nb; b++) if (bdepth[b] == L) blevbodies.push_back(b);
  }
  blevstart[nblev] = (int)blevbodies.size();

  // ---- dof constants: arm, lo, hi, limited, invw, kp, kd, tlim, ascale, aoffset, actuated, pad
  std::vector<float> dofc(nv * kDofC, 0.f);
  out.dof_act.assign(nv, -1);
  for (int i = 0; i < d.nu; i++) {
    int dof = d.actuator_dof[i];
    if (dof < 6 || dof >= nv || out.dof_act[dof] >= 0) { out.error = "bad actuator_dof"; return false; }
    out.dof_act[dof] = i;
  }
  for (int i = 0; i < nv; i++) {
    float *c = &dofc[i * kDofC];
    c[0] = (float)d.dof_armature[i];
    c[1] = (float)d.jnt_range[2 * i]; c[2] = (float)d.jnt_range[2 * i + 1];
    c[3] = (i >= 6 && d.jnt_limited[i]) ? 1.f : 0.f;
    c[4] = (float)d.dof_invweight0[i];
    int a = out.dof_act[i];
    if (a >= 0) {
      c[5] = (float)d.kp[a]; c[6] = (float)d.kd[a]; c[7] = (float)d.torque_lim[a];
      c[8] = (float)d.act_scale[a]; c[9] = (float)d.act_offset[a]; c[10] = 1.f; c[11] = (float)a;
    } else c[11] = -1.f;
  }

  // ---- body constants (registers): off3 ipos3 mass Ibody(xx xy xz yy yz zz) invw_tr
  out.bodyc.assign(nb * kBodyC, 0.f);
  for (int b = 0; b < nb; b++) {
    float *c = &out.bodyc[b * kBodyC];
    for (int k = 0; k < 3; k++) { c[k] = (float)d.body_pos[3 * b + k]; c[3 + k] = (float)d.body_ipos[3 * b + k]; }
    c[6] = (float)d.body_mass[b];
    double R[9]; quat2mat(d.body_iquat + 4 * b, R);
    double I[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
      double s = 0; for (int k = 0; k < 3; k++) s += R[3 * i + k] * d.body_inertia[3 * b + k] * R[3 * j + k];
      I[3 * i + j] = s;
    }
    c[7] = (float)I[0]; c[8] = (float)I[1]; c[9] = (float)I[2]; c[10] = (float)I[4]; c[11] = (float)I[5]; c[12] = (float)I[8];
    c[13] = (float)d.body_invweight0[2 * b];
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
      float c[kCandC] = {0};
      for (int r = 0; r < 3; r++) {
        c[r] = (float)(G[3 * r] * v[0] + G[3 * r + 1] * v[1] + G[3 * r + 2] * v[2]);   // corner vector (body frame)
        c[3 + r] = (float)d.geom_pos[3 * b + r];                                        // box centre (body frame)
      }
      c[7] = (float)d.body_invweight0[2 * b];
      out.candc.insert(out.candc.end(), c, c + kCandC); out.candb.push_back(b);
    }
  }
  for (int b = 0; b < nb; b++) {
    if (d.geom_type[b] != SS_GEOM_CAPSULE) continue;
    double G[9]; quat2mat(d.geom_quat + 4 * b, G);
    for (int s = 0; s < 2; s++) {
      double sg = s ? -1.0 : 1.0;
      float c[kCandC] = {0};
      for (int r = 0; r < 3; r++) {
        c[r] = (float)(d.geom_pos[3 * b + r] + sg * G[3 * r + 2] * d.geom_size[3 * b + 1]);  // end-sphere centre (body frame)
        c[3 + r] = (float)G[3 * r + 2];                                                      // capsule axis (body frame)
      }
      c[6] = (float)d.geom_size[3 * b];                                                      // radius
      c[7] = (float)d.body_invweight0[2 * b];
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
  if (nlev > 19 || nblev > 19) { out.error = "tree too deep"; return false; }
  for (int L = 0; L <= nlev; L++) h.levstart[L] = levstart[L];
  for (int L = 0; L <= nblev; L++) h.blevstart[L] = blevstart[L];
  // packed work-item records (one LDS read per item instead of a chain of dependent table reads); every block is
  // addressed by its float offset into H (or into the scratch buffer of parked P blocks, 12 floats per block):
  //  blk   (assemble)                 w0 = aJ | n<<8 | diag<<16            w1 = block(n,J)
  //  itemA (phase A, per level)       block(k,J)(13b) | n(6)<<13 | J(4)<<19 | kk(4)<<23
  //  itemB (phase B targets)          w0 = block(aI,J) | nsrc<<16          w1 = src_start
  //        fsrc                       scratch block (kk*L + I)*12 | block(k,J)<<16
  //  bsol  (fused leaves-to-root sweep targets)   w0 = aJ | nsrc<<8        w1 = src_start
  //        bsrc                       block(k,J) | n_k<<16
  std::vector<int> blk, itemA, itemB, fsrc, bsol, bsrc, accp, children;
  if (ne > 8191) { out.error = "H too large for the packed item tables"; return false; }
  for (int n = 0; n < nn; n++) for (int J = 0; J <= ndepth[n]; J++) {
    int aJ = chainnode[n * CN + J];
    blk.push_back(aJ | (n << 8) | ((aJ == n) << 16));
    blk.push_back(nbase[n] + 12 * J);
  }
  h.nblk = (int)blk.size() / 2;
  for (int L = 0; L < nlev; L++) {
    h.itemA[L] = (int)itemA.size(); h.itemB[L] = (int)itemB.size() / 2;
    int nk = levstart[L + 1] - levstart[L];
    if (nk > 16 || L > 15) { out.error = "model exceeds the packed item-table field widths"; return false; }
    for (int kk = 0; kk < nk; kk++) {
      int n = levnodes[levstart[L] + kk];
      for (int J = 0; J < std::max(L, 1); J++)
        itemA.push_back((nbase[n] + 12 * J) | (n << 13) | (J << 19) | (kk << 23));
    }
    for (int I = 0; I < L; I++) for (int J = 0; J <= I; J++) {
      std::vector<int> seen;
      for (int kk = 0; kk < nk; kk++) {
        int aI = chainnode[levnodes[levstart[L] + kk] * CN + I];
        bool dup = false; for (int a : seen) dup |= (a == aI);
        if (dup) continue;
        seen.push_back(aI);
        int start = (int)fsrc.size(), cnt = 0;
        for (int k2 = 0; k2 < nk; k2++) {
          int n2 = levnodes[levstart[L] + k2];
          if (chainnode[n2 * CN + I] != aI) continue;
          fsrc.push_back(((k2 * L + I) * 12) | ((nbase[n2] + 12 * J) << 16)); cnt++;
        }
        itemB.push_back((nbase[aI] + 12 * J) | (cnt << 16));
        itemB.push_back(start);
      }
    }
    h.bsol[L] = (int)bsol.size() / 2;
    for (int J = 0; J < L; J++) {
      std::vector<int> seen;
      for (int kk = 0; kk < nk; kk++) {
        int aJ = chainnode[levnodes[levstart[L] + kk] * CN + J];
        bool dup = false; for (int a : seen) dup |= (a == aJ);
        if (dup) continue;
        seen.push_back(aJ);
        int start = (int)bsrc.size(), cnt = 0;
        for (int k2 = 0; k2 < nk; k2++) {
          int n2 = levnodes[levstart[L] + k2];
          if (chainnode[n2 * CN + J] != aJ) continue;
          bsrc.push_back((nbase[n2] + 12 * J) | (n2 << 16)); cnt++;
        }
        bsol.push_back(aJ | (cnt << 8)); bsol.push_back(start);
      }
    }
  }
  h.itemA[nlev] = (int)itemA.size(); h.itemB[nlev] = (int)itemB.size() / 2;
  h.bsol[nlev] = (int)bsol.size() / 2;
  // tree accumulation (children -> parent), pull form, per body level: p | cstart<<8 | ccount<<20
  for (int L = 0; L < nblev; L++) {
    h.accp[L] = (int)accp.size();
    for (int b = 0; b < nb; b++) {
      if (bdepth[b] != L) continue;
      int start = (int)children.size(), cnt = 0;
      for (int c = 0; c < nb; c++) if (d.body_parent[c] == b) { children.push_back(c); cnt++; }
      if (cnt) accp.push_back(b | (start << 8) | (cnt << 20));
    }
  }
  h.accp[nblev] = (int)accp.size();
  h.itemA[nlev] = (int)itemA.size(); h.itemB[nlev] = (int)itemB.size() / 2;
  out.decode = decode;
  auto &S = out.shared; S.clear();
  auto push_i = [&](const std::vector<int> &v) { int o = (int)S.size(); for (int x : v) S.push_back((uint32_t)x); return o; };
  h.o_dofc = (int)S.size(); for (float f : dofc) S.push_back(f2u(f));
  h.o_chainnode = push_i(chainnode);
  h.o_nbase = push_i(nbase);
  h.o_ndepth = push_i(ndepth);
  h.o_levnodes = push_i(levnodes);
  std::vector<int> bpar(d.body_parent, d.body_parent + nb);
  h.o_bparent = push_i(bpar);
  h.o_blevbodies = push_i(blevbodies);
  h.o_blk = push_i(blk);
  h.o_itemA = push_i(itemA);
  h.o_itemB = push_i(itemB);
  h.o_fsrc = push_i(fsrc);
  h.o_accp = push_i(accp);
  h.o_children = push_i(children);
  h.o_bsol = push_i(bsol);
  h.o_bsrc = push_i(bsrc);
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
    h.o_subsize = push_i(subsize);
  }
  h.shared_words = (int)S.size();
  (void)chainrow; (void)nparent; (void)CW;

  // ---- per-env LDS layout (floats)
  int maxU = 0;                                            // U buffer: nodes-in-level * 3 * D
  for (int L = 1; L < nlev; L++) maxU = std::max(maxU, (levstart[L + 1] - levstart[L]) * 12 * L);
  if (nn > 64) { out.error = "too many nodes"; return false; }
  if (13 * h.nslot > ne) { out.error = "contact record buffer does not fit"; return false; }
  int o = 0;
  auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
  // Arrays with disjoint lifetimes share storage (LDS capacity sets the number of resident envs per CU):
  //   H region  : contact records (make_constraints) | per-body K inputs, subtree-summed Kc and Gb (forward_kin /
  //               newton_prepare, dead before assemble_H writes H) | the matrix itself
  //   G region  : R, r and Ad (forward pass .. constraints / row evaluation) | G_i = Hc S_i (assembly) | U buffers |
  //               scratch of body_accel / the forward triangular sweep
  //   grad,delta: V (body velocities, dead after make_constraints)
  h.l_H = take(ne);
  if (48 * nb > ne || 13 * h.nslot > ne) { out.error = "H region too small for its aliases"; return false; }
  h.l_K = h.l_H + 21 * nb;                                 // Kc (subtree sums); inputs live at H[0 .. 21 nb)
  h.l_Gb = h.l_H + 42 * nb;
  h.l_S = take(6 * nv);
  const int gneed = std::max(std::max(6 * nv, maxU), 18 * nb);
  h.l_G = take(gneed);                                     // also the scratch buffer of parked P blocks in the factorization
  h.maxU = maxU;
  h.l_R = h.l_G; h.l_r = h.l_G + 9 * nb;                   // R, r: first 12 nb floats of G
  h.l_Ad = h.l_G + gneed - 6 * nb;                         // Ad: last 6 nb floats of G (body_accel scratch uses G[0 .. 6 nn))
  if (12 * nb > gneed - 6 * nb || 6 * nn > gneed - 6 * nb) { out.error = "G region too small for its aliases"; return false; }
  h.l_Dinv = take(6 * nn);
  h.l_Ic = take(10 * nb);
  h.l_Ab = take(6 * nb);
  h.l_q = take(nv + 1); h.l_v = take(nv); h.l_a = take(nv); h.l_tau = take(nv);
  h.l_grad = take(nv); h.l_delta = take(nv);
  h.l_V = h.l_grad;                                        // V: 6 nb <= 2 (nv+pad) floats of grad + delta
  if (6 * nb > h.l_delta + nv - h.l_grad) { out.error = "V alias does not fit"; return false; }
  h.l_C = take(nv); h.l_diag = take(nv);
  h.l_misc = take(16);
  h.env_floats = o;

  h.dt = (float)d.timestep; h.grav = (float)d.gravity; h.margin = (float)d.margin; h.mu = (float)d.friction;
  for (int k = 0; k < 5; k++) h.solimp[k] = (float)d.solimp[k];
  double dmax = d.solimp[1];
  double tc = d.solref[0] < 2 * d.timestep ? 2 * d.timestep : d.solref[0];
  h.K = (float)(1.0 / (dmax * dmax * tc * tc * d.solref[1] * d.solref[1]));
  h.B = (float)(2.0 / (dmax * tc));
  for (int k = 0; k < 3; k++) h.qpos0_root[k] = (float)d.qpos0[k];
  if (d.impratio != 1.0) { out.error = "impratio != 1 is not supported"; return false; }
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
