/* oracle.c — CPU float64 restatement of the SMPLSim env-step hot path (see oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY — never linked or loaded by the product (smplsim_amd/).
 *
 * What each part follows:
 *   om_model_create      MuJoCo's XML compile for the MJCF subset the reference's humanoids use
 *                        (call site: reference smpl_sim/envs/base_env.py:139-142); [MJ-doc]
 *   om_kinematics        mj_kinematics (call site humanoid_env.py:389); [MJ-doc]
 *   om_forward/om_step   mj_forward / mj_step (call sites humanoid_env.py:450,484,491,507); [MJ-doc]
 *   om_spd_torque        StablePDController, reference smpl_sim/envs/controllers.py:116-190
 *   om_obs_v1/v2         reference smpl_sim/envs/humanoid_env.py:565-633 / :637-687 with the wxyz
 *                        quaternion helpers of smpl_sim/utils/np_transform_utils.py:16-146
 *   om_env_*             humanoid_env.py:439-512, humanoid_task.py, tasks/humanoid_speed.py,
 *                        tasks/humanoid_getup.py
 * [MJ-doc] = MuJoCo (>=3) is absent from /root/reference and the container; those parts restate
 * its documented pipeline and are "parity unpinned".
 *
 * Deliberately written in a different formulation from the HIP kernels (3-D Newton-Euler with
 * dense per-body Jacobians, dense Cholesky, explicit constraint Jacobian) so that agreement
 * between the two is evidence, not tautology.
 */
#include "oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define MAXSELF 256                     /* body-body contacts */
#define MAXCON (4 * OM_MAXB + MAXSELF)
#define MAXPAIR (OM_MAXB * (OM_MAXB - 1) / 2)
#define MINVAL 1e-15

/* ------------------------------------------------------------------ small math */
static inline void v3set(double *r, double a, double b, double c) { r[0] = a; r[1] = b; r[2] = c; }
static inline void v3cpy(double *r, const double *a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
static inline void v3add(double *r, const double *a, const double *b) { r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; }
static inline void v3sub(double *r, const double *a, const double *b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
static inline void v3addscl(double *r, const double *a, double s) { r[0] += a[0] * s; r[1] += a[1] * s; r[2] += a[2] * s; }
static inline double v3dot(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void v3cross(double *r, const double *a, const double *b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline double v3norm(const double *a) { return sqrt(v3dot(a, a)); }
/* mju_normalize3: returns the norm; tiny vectors become (1,0,0) */
static double v3normalize(double *a) {
  double n = v3norm(a);
  if (n < MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; }
  else { a[0] /= n; a[1] /= n; a[2] /= n; }
  return n;
}
static void m3mulv(double *r, const double *m, const double *v) { /* row-major 3x3 */
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  double y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  double z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void m3mul(double *r, const double *a, const double *b) {
  double t[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
    t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  memcpy(r, t, sizeof t);
}
static void qmul(double *r, const double *a, const double *b) { /* Hamilton, wxyz */
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void qnormalize(double *q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
  else { q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }
}
static void q2mat(double *m, const double *q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
static void axisangle2q(double *q, const double *axis, double angle) {
  double s = sin(angle * 0.5);
  q[0] = cos(angle * 0.5); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
static void qrotv(double *r, const double *q, const double *v) {
  double m[9]; q2mat(m, q); m3mulv(r, m, v);
}

/* dense Cholesky (lower) in place, n x n row-major with leading dimension n; returns 0 ok */
static int chol_factor(double *A, int n) {
  for (int j = 0; j < n; j++) {
    double s = A[j * n + j];
    for (int k = 0; k < j; k++) s -= A[j * n + k] * A[j * n + k];
    if (s <= 0) return -1;
    double l = sqrt(s);
    A[j * n + j] = l;
    for (int i = j + 1; i < n; i++) {
      double t = A[i * n + j];
      for (int k = 0; k < j; k++) t -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = t / l;
    }
  }
  return 0;
}
static void chol_solve(const double *L, int n, double *b) {
  for (int i = 0; i < n; i++) {
    double t = b[i];
    for (int k = 0; k < i; k++) t -= L[i * n + k] * b[k];
    b[i] = t / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double t = b[i];
    for (int k = i + 1; k < n; k++) t -= L[k * n + i] * b[k];
    b[i] = t / L[i * n + i];
  }
}

/* ------------------------------------------------------------------ model */
struct om_model {
  int nbody, nv, nq, nu;
  int parent[OM_MAXB];
  double body_pos[OM_MAXB][3];
  double mass[OM_MAXB], ipos[OM_MAXB][3], iquat[OM_MAXB][4], inertia[OM_MAXB][3], imat[OM_MAXB][9];
  int gtype[OM_MAXB];
  double gsize[OM_MAXB][3], gpos[OM_MAXB][3], gquat[OM_MAXB][4], gmat[OM_MAXB][9];
  double armature[OM_MAXV], range[OM_MAXV][2];
  int limited[OM_MAXV];
  int act_dof[OM_MAXV];
  double kp[OM_MAXV], kd[OM_MAXV], tlim[OM_MAXV], ascale[OM_MAXV], aoffset[OM_MAXV];
  int legal[OM_MAXB];
  double body_invw[OM_MAXB][2], dof_invw[OM_MAXV];
  double dt, grav, solref[2], solimp[5], margin, mu, impratio;
  int chain_len[OM_MAXB];
  int chain[OM_MAXB][OM_MAXV];
  /* body-body collision: candidate pairs after the contype/conaffinity test, the parent-child filter and the excludes;
   * b1 is the geom MuJoCo hands to the pair function first (lower geom type, then lower id): normals point b1 -> b2 */
  int self_collision, npair, max_self;
  int pair_b1[MAXPAIR], pair_b2[MAXPAIR];
  double brad[OM_MAXB];                       /* bounding-sphere radius of the body's geom about its centre */
  /* constraint solver settings (om_model_set_solver): mjModel.stat.meaninertia, mjOption.tolerance / iterations */
  double meaninertia, tolerance;
  int solver_mode, iterations;
  int ls_mode, ls_iterations;                  /* om_model_set_linesearch: OM_LS_EXACT | OM_LS_MUJOCO (mjOption.ls_tolerance / ls_iterations) */
  double ls_tolerance;
};

struct om_data {
  double qpos[OM_MAXV + 1], qvel[OM_MAXV], qacc[OM_MAXV], warm[OM_MAXV], ctrl[OM_MAXV];
  double qfrc_act[OM_MAXV], bias[OM_MAXV], qacc_smooth[OM_MAXV], qfrc_constraint[OM_MAXV];
  double *M;                                   /* nv x nv */
  double xpos[OM_MAXB][3], xquat[OM_MAXB][4], xmat[OM_MAXB][9], xipos[OM_MAXB][3], Iw[OM_MAXB][9];
  double axis[OM_MAXB][3][3];
  double linvel[OM_MAXB][3], angvel[OM_MAXB][3];
  int ncon, con_body[MAXCON], con_body1[MAXCON];   /* con_body1 = -1: floor contact */
  int nself, nself_dropped;
  double con_pos[MAXCON][3], con_frame[MAXCON][9], con_dist[MAXCON];
  int touch[OM_MAXB];
  int nefc, maxrow;
  double *J, *epos, *emargin, *ediag, *eD, *eR, *earef, *eforce, *ejar, *ejd;
  double *H, *work, *work2;                    /* nv x nv each */
  int solver_iter, nwarn;
  long ls_evals, ls_calls, ls_exhausted;         /* line-search statistics since creation (OM_D_LS_STATS) */
  double qpos_fwd[OM_MAXV + 1], qvel_fwd[OM_MAXV];   /* the state of the last om_forward: what the (stale) M and bias belong to */
  /* SimplePID state (reference controllers.py:193-262): per actuator integral and last error; pid_dt = the dt the env
   * hands the controller (timestep * control_freq_inv, humanoid_env.py:319) */
  double pid_i[OM_MAXV], pid_e[OM_MAXV], pid_dt;
  int pid_started;
};

static void dof_jac(const om_model *m_unused, const om_data *d, int dof, const double *P, double *jv, double *jw) {
  if (dof < 3) { v3set(jv, 0, 0, 0); jv[dof] = 1; v3set(jw, 0, 0, 0); return; }
  (void)m_unused; int b, j;
  if (dof < 6) { b = 0; j = dof - 3; } else { b = 1 + (dof - 6) / 3; j = (dof - 6) % 3; }
  const double *ax = d->axis[b][j];
  double r[3]; v3sub(r, P, d->xpos[b]);
  v3cpy(jw, ax); v3cross(jv, ax, r);
}

static void geom_mass_inertia(int type, const double *size, double density, double *mass, double *inertia) {
  if (type == OM_GEOM_BOX) {
    double a = size[0], b = size[1], c = size[2];
    double mm = density * 8.0 * a * b * c;
    *mass = mm;
    inertia[0] = mm / 3 * (b * b + c * c); inertia[1] = mm / 3 * (a * a + c * c); inertia[2] = mm / 3 * (a * a + b * b);
  } else {
    double r = size[0], h = 2 * size[1];
    double mc = density * M_PI * r * r * h, ms = density * 4.0 / 3.0 * M_PI * r * r * r;
    double ip = mc * (3 * r * r + h * h) / 12 + 0.4 * ms * r * r + ms * h * (3 * r + 2 * h) / 8;
    *mass = mc + ms;
    inertia[0] = inertia[1] = ip; inertia[2] = mc * r * r / 2 + 0.4 * ms * r * r;
  }
}

static void compute_M(const om_model *m, om_data *d);

om_model *om_model_create(const om_desc *ds) {
  if (ds->nbody < 1 || ds->nbody > OM_MAXB) return NULL;
  om_model *m = (om_model *)calloc(1, sizeof *m);
  int nb = ds->nbody;
  m->nbody = nb; m->nv = 6 + 3 * (nb - 1); m->nq = m->nv + 1; m->nu = ds->nu;
  for (int b = 0; b < nb; b++) {
    m->parent[b] = ds->parent[b];
    v3cpy(m->body_pos[b], ds->body_pos + 3 * b);
    const double *gp = ds->geom_params + 10 * b;
    m->gtype[b] = ds->geom_type[b];
    if (m->gtype[b] == OM_GEOM_BOX) {
      v3cpy(m->gpos[b], gp); v3cpy(m->gsize[b], gp + 3);
      memcpy(m->gquat[b], gp + 6, 4 * sizeof(double)); qnormalize(m->gquat[b]);
    } else if (m->gtype[b] == OM_GEOM_SPHERE) {               /* params: pos3 - - - radius; carried as a capsule of zero half length */
      v3cpy(m->gpos[b], gp); m->gsize[b][0] = gp[6]; m->gsize[b][1] = 0; m->gsize[b][2] = 0;
      m->gquat[b][0] = 1; m->gquat[b][1] = m->gquat[b][2] = m->gquat[b][3] = 0;
    } else {
      double vec[3]; v3sub(vec, gp + 3, gp);
      double len = v3norm(vec);
      m->gsize[b][0] = gp[6]; m->gsize[b][1] = 0.5 * len; m->gsize[b][2] = 0;
      for (int k = 0; k < 3; k++) m->gpos[b][k] = 0.5 * (gp[k] + gp[3 + k]);
      /* minimal rotation of +z onto vec */
      double u[3] = {vec[0] / len, vec[1] / len, vec[2] / len}, z[3] = {0, 0, 1}, ax[3];
      v3cross(ax, z, u);
      double s = v3norm(ax);
      if (s < 1e-10) v3set(ax, 1, 0, 0); else { ax[0] /= s; ax[1] /= s; ax[2] /= s; }
      axisangle2q(m->gquat[b], ax, atan2(s, u[2]));
    }
    q2mat(m->gmat[b], m->gquat[b]);
    geom_mass_inertia(m->gtype[b], m->gsize[b], ds->density[b], &m->mass[b], m->inertia[b]);
    v3cpy(m->ipos[b], m->gpos[b]);
    memcpy(m->iquat[b], m->gquat[b], 4 * sizeof(double));
    memcpy(m->imat[b], m->gmat[b], 9 * sizeof(double));
    m->legal[b] = ds->legal_contact ? ds->legal_contact[b] : 0;
    /* dof chain root -> b */
    if (b == 0) { m->chain_len[0] = 6; for (int k = 0; k < 6; k++) m->chain[0][k] = k; }
    else {
      int p = m->parent[b], n = m->chain_len[p];
      memcpy(m->chain[b], m->chain[p], n * sizeof(int));
      for (int k = 0; k < 3; k++) m->chain[b][n + k] = 6 + 3 * (b - 1) + k;
      m->chain_len[b] = n + 3;
    }
  }
  for (int i = 0; i < m->nv; i++) {
    m->armature[i] = ds->armature[i];
    m->limited[i] = ds->limited[i];
    m->range[i][0] = ds->range_deg[2 * i] * M_PI / 180.0;
    m->range[i][1] = ds->range_deg[2 * i + 1] * M_PI / 180.0;
  }
  for (int i = 0; i < m->nu; i++) {
    m->act_dof[i] = ds->act_dof[i];
    m->kp[i] = ds->kp[i]; m->kd[i] = ds->kd[i]; m->tlim[i] = ds->torque_lim[i];
    m->ascale[i] = ds->act_scale[i]; m->aoffset[i] = ds->act_offset[i];
  }
  /* candidate body pairs (mj_collision's static filters) */
  m->self_collision = ds->self_collision; m->max_self = ds->max_self_contacts; m->npair = 0;
  for (int b = 0; b < nb; b++) {
    const double *sz = m->gsize[b];
    m->brad[b] = m->gtype[b] == OM_GEOM_BOX ? sqrt(sz[0] * sz[0] + sz[1] * sz[1] + sz[2] * sz[2]) : sz[0] + sz[1];
  }
  if (ds->self_collision)
    for (int i = 0; i < nb; i++) for (int j = i + 1; j < nb; j++) {
      int ct1 = ds->contype ? ds->contype[i] : 1, ca1 = ds->conaffinity ? ds->conaffinity[i] : 1;
      int ct2 = ds->contype ? ds->contype[j] : 1, ca2 = ds->conaffinity ? ds->conaffinity[j] : 1;
      if (!((ct1 & ca2) || (ct2 & ca1))) continue;
      if (m->parent[j] == i || m->parent[i] == j) continue;             /* filterparent (the world is nobody's geom parent here) */
      int ex = 0;
      for (int e = 0; e < ds->nexclude; e++)
        if ((ds->exclude[2 * e] == i && ds->exclude[2 * e + 1] == j) || (ds->exclude[2 * e] == j && ds->exclude[2 * e + 1] == i)) ex = 1;
      if (ex) continue;
      /* MuJoCo hands the geom of the lower type id to the pair function first: sphere (2) < capsule (3) < box (6) */
      int ri = m->gtype[i] == OM_GEOM_SPHERE ? 0 : (m->gtype[i] == OM_GEOM_CAPSULE ? 1 : 2), rj = m->gtype[j] == OM_GEOM_SPHERE ? 0 : (m->gtype[j] == OM_GEOM_CAPSULE ? 1 : 2);
      int first = rj < ri ? j : i;
      m->pair_b1[m->npair] = first; m->pair_b2[m->npair] = first == i ? j : i; m->npair++;
    }
  m->dt = ds->timestep; m->grav = ds->gravity;
  memcpy(m->solref, ds->solref, sizeof m->solref); memcpy(m->solimp, ds->solimp, sizeof m->solimp);
  m->margin = ds->margin; m->mu = ds->mu; m->impratio = ds->impratio;

  /* invweight0 at qpos0 */
  om_data *d = om_data_create(m);
  memset(d->qpos, 0, sizeof d->qpos);
  v3cpy(d->qpos, m->body_pos[0]); d->qpos[3] = 1;
  om_kinematics(m, d);
  compute_M(m, d);
  int nv = m->nv;
  double *L = (double *)malloc(sizeof(double) * nv * nv);
  memcpy(L, d->M, sizeof(double) * nv * nv);
  chol_factor(L, nv);
  double *Minv = (double *)calloc((size_t)nv * nv, sizeof(double));
  double *col = (double *)malloc(sizeof(double) * nv);
  for (int j = 0; j < nv; j++) {
    memset(col, 0, sizeof(double) * nv); col[j] = 1;
    chol_solve(L, nv, col);
    for (int i = 0; i < nv; i++) Minv[i * nv + j] = col[i];
  }
  for (int b = 0; b < nb; b++) {
    int n = m->chain_len[b];
    double tr = 0, rot = 0;
    for (int k = 0; k < 3; k++) {                 /* row k of Jp and Jr */
      for (int which = 0; which < 2; which++) {
        double acc = 0;
        for (int a = 0; a < n; a++) {
          double jva[3], jwa[3]; dof_jac(m, d, m->chain[b][a], d->xipos[b], jva, jwa);
          double ja = which ? jwa[k] : jva[k];
          if (ja == 0) continue;
          for (int c = 0; c < n; c++) {
            double jvc[3], jwc[3]; dof_jac(m, d, m->chain[b][c], d->xipos[b], jvc, jwc);
            double jc = which ? jwc[k] : jvc[k];
            acc += ja * Minv[m->chain[b][a] * nv + m->chain[b][c]] * jc;
          }
        }
        if (which) rot += acc; else tr += acc;
      }
    }
    m->body_invw[b][0] = tr / 3; m->body_invw[b][1] = rot / 3;
  }
  for (int i = 0; i < nv; i++) m->dof_invw[i] = Minv[i * nv + i];
  /* mj_setConst: stat.meaninertia = mean diagonal of qM at qpos0; MuJoCo's solver defaults (the reference MJCF sets neither) */
  m->meaninertia = 0;
  for (int i = 0; i < nv; i++) m->meaninertia += d->M[i * nv + i];
  m->meaninertia /= nv > 1 ? nv : 1;
  m->solver_mode = OM_SOLVER_MUJOCO; m->tolerance = 1e-8; m->iterations = 100;
  m->ls_mode = OM_LS_EXACT; m->ls_tolerance = 0.01; m->ls_iterations = 50;
  for (int g = 0; g < 2; g++) {
    double a = (m->dof_invw[3 * g] + m->dof_invw[3 * g + 1] + m->dof_invw[3 * g + 2]) / 3;
    m->dof_invw[3 * g] = m->dof_invw[3 * g + 1] = m->dof_invw[3 * g + 2] = a;
  }
  free(L); free(Minv); free(col); om_data_destroy(d);
  return m;
}
void om_model_destroy(om_model *m) { free(m); }
void om_model_set_linesearch(om_model *m, int mode, double ls_tolerance, int ls_iterations) {
  m->ls_mode = mode;
  if (ls_tolerance > 0) m->ls_tolerance = ls_tolerance;
  if (ls_iterations > 0) m->ls_iterations = ls_iterations;
}
void om_model_set_solver(om_model *m, int mode, double tolerance, int iterations) {
  m->solver_mode = mode;
  if (tolerance > 0) m->tolerance = tolerance;
  if (iterations > 0) m->iterations = iterations;
}

int om_model_get(const om_model *m, int field, double *out) {
  int nb = m->nbody;
  switch (field) {
    case OM_M_MASS: memcpy(out, m->mass, nb * sizeof(double)); return nb;
    case OM_M_IPOS: for (int b = 0; b < nb; b++) v3cpy(out + 3 * b, m->ipos[b]); return 3 * nb;
    case OM_M_IQUAT: for (int b = 0; b < nb; b++) memcpy(out + 4 * b, m->iquat[b], 32); return 4 * nb;
    case OM_M_INERTIA: for (int b = 0; b < nb; b++) v3cpy(out + 3 * b, m->inertia[b]); return 3 * nb;
    case OM_M_GPOS: for (int b = 0; b < nb; b++) v3cpy(out + 3 * b, m->gpos[b]); return 3 * nb;
    case OM_M_GQUAT: for (int b = 0; b < nb; b++) memcpy(out + 4 * b, m->gquat[b], 32); return 4 * nb;
    case OM_M_GSIZE: for (int b = 0; b < nb; b++) v3cpy(out + 3 * b, m->gsize[b]); return 3 * nb;
    case OM_M_BODY_INVW: for (int b = 0; b < nb; b++) { out[2 * b] = m->body_invw[b][0]; out[2 * b + 1] = m->body_invw[b][1]; } return 2 * nb;
    case OM_M_DOF_INVW: memcpy(out, m->dof_invw, m->nv * sizeof(double)); return m->nv;
    case OM_M_RANGE: for (int i = 0; i < m->nv; i++) { out[2 * i] = m->range[i][0]; out[2 * i + 1] = m->range[i][1]; } return 2 * m->nv;
    case OM_M_MEANINERTIA: out[0] = m->meaninertia; return 1;
  }
  return -1;
}

/* ------------------------------------------------------------------ data */
om_data *om_data_create(const om_model *m) {
  om_data *d = (om_data *)calloc(1, sizeof *d);
  int nv = m->nv;
  d->maxrow = 4 * MAXCON + 2 * nv;
  d->M = (double *)calloc((size_t)nv * nv, sizeof(double));
  d->H = (double *)calloc((size_t)nv * nv, sizeof(double));
  d->work = (double *)calloc((size_t)nv * nv, sizeof(double));
  d->work2 = (double *)calloc((size_t)nv * nv, sizeof(double));
  d->J = (double *)calloc((size_t)d->maxrow * nv, sizeof(double));
  double **arrs[] = {&d->epos, &d->emargin, &d->ediag, &d->eD, &d->eR, &d->earef, &d->eforce, &d->ejar, &d->ejd};
  for (unsigned i = 0; i < sizeof arrs / sizeof arrs[0]; i++) *arrs[i] = (double *)calloc(d->maxrow, sizeof(double));
  d->qpos[3] = 1;
  return d;
}
void om_data_destroy(om_data *d) {
  if (!d) return;
  free(d->M); free(d->H); free(d->work); free(d->work2); free(d->J);
  free(d->epos); free(d->emargin); free(d->ediag); free(d->eD); free(d->eR); free(d->earef);
  free(d->eforce); free(d->ejar); free(d->ejd);
  free(d);
}

/* ------------------------------------------------------------------ kinematics */
void om_kinematics(const om_model *m, om_data *d) {
  static const double E[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  double q[4];
  memcpy(q, d->qpos + 3, sizeof q); qnormalize(q);
  v3cpy(d->xpos[0], d->qpos); memcpy(d->xquat[0], q, sizeof q); q2mat(d->xmat[0], q);
  for (int k = 0; k < 3; k++) v3set(d->axis[0][k], d->xmat[0][k], d->xmat[0][3 + k], d->xmat[0][6 + k]);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->parent[b];
    double off[3]; m3mulv(off, d->xmat[p], m->body_pos[b]);
    v3add(d->xpos[b], d->xpos[p], off);
    memcpy(q, d->xquat[p], sizeof q);
    for (int j = 0; j < 3; j++) {
      qrotv(d->axis[b][j], q, E[j]);                        /* axis before applying this joint */
      double ql[4]; axisangle2q(ql, E[j], d->qpos[7 + 3 * (b - 1) + j]);
      qmul(q, q, ql);
    }
    qnormalize(q);
    memcpy(d->xquat[b], q, sizeof q); q2mat(d->xmat[b], q);
  }
  for (int b = 0; b < m->nbody; b++) {
    double t[3]; m3mulv(t, d->xmat[b], m->ipos[b]); v3add(d->xipos[b], d->xpos[b], t);
    double xim[9]; m3mul(xim, d->xmat[b], m->imat[b]);
    /* Iw = xim diag(inertia) xim^T */
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += xim[3 * i + k] * m->inertia[b][k] * xim[3 * j + k];
      d->Iw[b][3 * i + j] = s;
    }
  }
}

/* joint-space inertia: M = sum_b Jp^T m Jp + Jr^T Iw Jr (+ armature) */
static void compute_M(const om_model *m, om_data *d) {
  int nv = m->nv;
  memset(d->M, 0, sizeof(double) * nv * nv);
  double ljv[OM_MAXV][3], ljw[OM_MAXV][3], lijw[OM_MAXV][3];
  for (int b = 0; b < m->nbody; b++) {
    int n = m->chain_len[b];
    for (int a = 0; a < n; a++) {
      dof_jac(m, d, m->chain[b][a], d->xipos[b], ljv[a], ljw[a]);
      m3mulv(lijw[a], d->Iw[b], ljw[a]);
    }
    for (int a = 0; a < n; a++) for (int c = 0; c <= a; c++) {
      double v = m->mass[b] * v3dot(ljv[a], ljv[c]) + v3dot(ljw[a], lijw[c]);
      int i = m->chain[b][a], j = m->chain[b][c];
      d->M[i * nv + j] += v;
      if (i != j) d->M[j * nv + i] += v;
    }
  }
  for (int i = 0; i < nv; i++) d->M[i * nv + i] += m->armature[i];
}

/* velocities (sensors) and bias force qfrc_bias = RNE(q, qd, 0) incl. gravity */
static void compute_bias(const om_model *m, om_data *d) {
  int nv = m->nv;
  double w[OM_MAXB][3], al[OM_MAXB][3], vo[OM_MAXB][3], ao[OM_MAXB][3];
  m3mulv(w[0], d->xmat[0], d->qvel + 3);
  v3cpy(vo[0], d->qvel); v3set(al[0], 0, 0, 0); v3set(ao[0], 0, 0, 0);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->parent[b];
    double r[3], t[3], t2[3];
    v3sub(r, d->xpos[b], d->xpos[p]);
    v3cross(t, w[p], r); v3add(vo[b], vo[p], t);
    v3cross(t2, w[p], t);                                   /* w x (w x r) */
    v3cross(t, al[p], r);
    for (int k = 0; k < 3; k++) ao[b][k] = ao[p][k] + t[k] + t2[k];
    v3cpy(w[b], w[p]); v3cpy(al[b], al[p]);
    for (int j = 0; j < 3; j++) {
      double qd = d->qvel[6 + 3 * (b - 1) + j];
      v3cross(t, w[b], d->axis[b][j]);
      v3addscl(al[b], t, qd);
      v3addscl(w[b], d->axis[b][j], qd);
    }
  }
  memset(d->bias, 0, sizeof(double) * nv);
  for (int b = 0; b < m->nbody; b++) {
    v3cpy(d->linvel[b], vo[b]); v3cpy(d->angvel[b], w[b]);
    double rc[3], t[3], t2[3], ac[3], F[3], N[3], Iw_w[3];
    v3sub(rc, d->xipos[b], d->xpos[b]);
    v3cross(t, w[b], rc); v3cross(t2, w[b], t);
    v3cross(t, al[b], rc);
    for (int k = 0; k < 3; k++) ac[k] = ao[b][k] + t[k] + t2[k];
    v3set(F, m->mass[b] * ac[0], m->mass[b] * ac[1], m->mass[b] * (ac[2] - m->grav));
    m3mulv(N, d->Iw[b], al[b]);
    m3mulv(Iw_w, d->Iw[b], w[b]);
    v3cross(t, w[b], Iw_w); v3add(N, N, t);
    for (int a = 0; a < m->chain_len[b]; a++) {
      double jv[3], jw[3]; dof_jac(m, d, m->chain[b][a], d->xipos[b], jv, jw);
      d->bias[m->chain[b][a]] += v3dot(jv, F) + v3dot(jw, N);
    }
  }
}

/* ------------------------------------------------------------------ collision (floor plane z = 0 only) */
static void make_frame(double *fr) { /* fr[0:3] normal, fr[3:6] hint or zero */
  v3normalize(fr);
  if (v3norm(fr + 3) < 0.5) {
    v3set(fr + 3, 0, 0, 0);
    if (fr[1] < 0.5 && fr[1] > -0.5) fr[4] = 1; else fr[5] = 1;
  }
  double dp = v3dot(fr, fr + 3);
  v3addscl(fr + 3, fr, -dp);
  v3normalize(fr + 3);
  v3cross(fr + 6, fr, fr + 3);
}

static void collide(const om_model *m, om_data *d) {
  d->ncon = 0;
  for (int b = 0; b < m->nbody; b++) {
    d->touch[b] = 0;
    double gp[3], gm[9], t[3];
    m3mulv(t, d->xmat[b], m->gpos[b]); v3add(gp, d->xpos[b], t);
    m3mul(gm, d->xmat[b], m->gmat[b]);
    if (m->gtype[b] == OM_GEOM_BOX) {
      int cnt = 0;
      for (int i = 0; i < 8 && cnt < 4; i++) {
        double vec[3] = {(i & 1) ? m->gsize[b][0] : -m->gsize[b][0], (i & 2) ? m->gsize[b][1] : -m->gsize[b][1],
                         (i & 4) ? m->gsize[b][2] : -m->gsize[b][2]};
        double corner[3]; m3mulv(corner, gm, vec);
        double ldist = corner[2];
        if (gp[2] + ldist > m->margin || ldist > 0) continue;
        int c = d->ncon++;
        d->con_body1[c] = -1;
        d->con_body[c] = b; d->con_dist[c] = gp[2] + ldist;
        v3set(d->con_pos[c], gp[0] + corner[0], gp[1] + corner[1], gp[2] + corner[2] - 0.5 * d->con_dist[c]);
        memset(d->con_frame[c], 0, 9 * sizeof(double)); d->con_frame[c][2] = 1;
        make_frame(d->con_frame[c]);
        cnt++;
      }
      d->touch[b] = cnt > 0;
    } else {
      double ax[3] = {gm[2], gm[5], gm[8]};
      const int sphere = m->gtype[b] == OM_GEOM_SPHERE;         /* mjc_PlaneSphere: one contact, no tangent hint */
      for (int s = 0; s < (sphere ? 1 : 2); s++) {
        double sg = s ? -1.0 : 1.0, c3[3];
        for (int k = 0; k < 3; k++) c3[k] = gp[k] + sg * ax[k] * m->gsize[b][1];
        double dist = c3[2] - m->gsize[b][0];
        if (dist > m->margin) continue;
        int c = d->ncon++;
        d->con_body1[c] = -1;
        d->con_body[c] = b; d->con_dist[c] = dist;
        v3set(d->con_pos[c], c3[0], c3[1], c3[2] - (m->gsize[b][0] + 0.5 * dist));
        memset(d->con_frame[c], 0, 9 * sizeof(double)); d->con_frame[c][2] = 1;
        if (!sphere) v3cpy(d->con_frame[c] + 3, ax);
        make_frame(d->con_frame[c]);
        d->touch[b] = 1;
      }
    }
  }
}

/* ------------------------------------------------------------------ body-body narrow phases
 * MuJoCo's pair functions for the geom types of the reference's humanoids (capsule, box), restated: [MJ-doc]
 *   capsule-capsule  mjc_CapsuleCapsule: closest points of the two segments, then a sphere-sphere test (two tests when the
 *                    axes are parallel)
 *   capsule-box      the point of the capsule's segment closest to the box (exact minimiser of the convex, piecewise
 *                    quadratic squared distance), a sphere-box test there (mjc_SphereBox) and one at the far end of the
 *                    segment: at most 2 contacts like mjc_CapsuleBox
 *   box-box          separating-axis test over the 15 axes; face axis: the vertices of either box within the margin of the
 *                    other's face and inside its rectangle (at most 8 contacts like mjc_BoxBox); edge-edge axis: one
 *                    contact at the closest points of the two supporting edges
 * "parity unpinned" like the rest of mj_step (oracle.h): the contact-selection details of mjc_CapsuleBox / mjc_BoxBox are
 * restated as rules, not line by line.  Normals point from the pair's first geom to its second. */
typedef struct { double pos[3], normal[3], dist; } ncon;

/* fb: direction to use when the centres (nearly) coincide — the separation vector is rounding noise then.  NULL = (1,0,0).
 * "Nearly" = closer than 1e-5 of the radii, far above float32 rounding of the positions: the float32 kernel and this code
 * take the same branch */
static int sphere_sphere(const double *p1, double r1, const double *p2, double r2, double margin, const double *fb, ncon *o) {
  double d[3]; v3sub(d, p2, p1);
  double len = v3norm(d), dist = len - (r1 + r2);
  if (dist > margin) return 0;
  if (len < 1e-5 * (r1 + r2)) { if (fb) v3cpy(o->normal, fb); else v3set(o->normal, 1, 0, 0); }
  else v3set(o->normal, d[0] / len, d[1] / len, d[2] / len);
  for (int k = 0; k < 3; k++) o->pos[k] = p1[k] + o->normal[k] * (r1 + 0.5 * dist);
  o->dist = dist;
  return 1;
}

static double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* geom = centre p, unit axis a (capsule z axis), radius r, half length h */
static int capsule_capsule(const double *p1, const double *a1, double r1, double h1, const double *p2, const double *a2, double r2,
                           double h2, double margin, ncon *o) {
  double dif[3]; v3sub(dif, p1, p2);
  double ma = v3dot(a1, a1), mb = -v3dot(a1, a2), mc = v3dot(a2, a2), u = -v3dot(a1, dif), v = v3dot(a2, dif);
  double det = ma * mc - mb * mb;
  double c1[3], c2[3];
  if (h1 == 0.0 || h2 == 0.0) {                              /* a sphere (mjc_SphereSphere / mjc_SphereCapsule): centre against the closest segment point */
    double x1 = h1 == 0.0 ? 0.0 : clampd(u / ma, -h1, h1), x2 = h2 == 0.0 ? 0.0 : clampd((v - mb * x1) / mc, -h2, h2);
    for (int k = 0; k < 3; k++) { c1[k] = p1[k] + a1[k] * x1; c2[k] = p2[k] + a2[k] * x2; }
    return sphere_sphere(c1, r1, c2, r2, margin, NULL, o);
  }
  if (fabs(det) >= MINVAL) {                                 /* general configuration */
    double x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
    if (x1 > h1) { x1 = h1; x2 = (v - mb * h1) / mc; }
    else if (x1 < -h1) { x1 = -h1; x2 = (v + mb * h1) / mc; }
    if (x2 > h2) { x2 = h2; x1 = clampd((u - mb * h2) / ma, -h1, h1); }
    else if (x2 < -h2) { x2 = -h2; x1 = clampd((u + mb * h2) / ma, -h1, h1); }
    for (int k = 0; k < 3; k++) { c1[k] = p1[k] + a1[k] * x1; c2[k] = p2[k] + a2[k] * x2; }
    /* axes that (nearly) intersect: push apart along their common perpendicular, on the side of geom 2's centre */
    double fb[3]; v3cross(fb, a1, a2); v3normalize(fb);
    if (v3dot(fb, dif) > 0) v3set(fb, -fb[0], -fb[1], -fb[2]);
    return sphere_sphere(c1, r1, c2, r2, margin, fb, o);
  }
  /* parallel axes: the ends of each segment against the other segment, at most two contacts */
  int n = 0;
  for (int e = 0; e < 4 && n < 2; e++) {
    double x1, x2;
    if (e < 2) { x1 = e == 0 ? h1 : -h1; x2 = (v - mb * x1) / mc; if (x2 > h2 || x2 < -h2) continue; }
    else { x2 = e == 2 ? h2 : -h2; x1 = (u - mb * x2) / ma; if (x1 > h1 || x1 < -h1) continue; }
    for (int k = 0; k < 3; k++) { c1[k] = p1[k] + a1[k] * x1; c2[k] = p2[k] + a2[k] * x2; }
    n += sphere_sphere(c1, r1, c2, r2, margin, NULL, o + n);
  }
  return n;
}

/* sphere (geom 1) against box (geom 2: centre bp, rotation bm row-major with the box axes as columns, half sizes bs) */
/* hint (box frame, may be NULL): which of two opposite faces a centre lying on the box's mid-plane is pushed out through
 * (capsule_box passes the capsule's centre: its mid-range rule puts the sphere exactly there when the axis skewers a thin box) */
static int sphere_box(const double *c, double r, const double *bp, const double *bm, const double *bs, double margin, const double *hint,
                      ncon *o) {
  double d[3], l[3], cl[3];
  v3sub(d, c, bp);
  for (int i = 0; i < 3; i++) { l[i] = bm[i] * d[0] + bm[3 + i] * d[1] + bm[6 + i] * d[2]; cl[i] = clampd(l[i], -bs[i], bs[i]); }
  double v[3] = {l[0] - cl[0], l[1] - cl[1], l[2] - cl[2]}, len = v3norm(v), nl[3], dist;
  if (len >= MINVAL) {                                       /* centre outside the box */
    dist = len - r;
    if (dist > margin) return 0;
    v3set(nl, v[0] / len, v[1] / len, v[2] / len);
  } else {                                                   /* centre inside: push out through the nearest face */
    int k = 0; double best = bs[0] - fabs(l[0]);
    for (int i = 1; i < 3; i++) if (bs[i] - fabs(l[i]) < best) { best = bs[i] - fabs(l[i]); k = i; }
    double sg = l[k] >= 0 ? 1.0 : -1.0;
    if (hint && fabs(l[k]) < 1e-5 * bs[k]) sg = hint[k] >= 0 ? 1.0 : -1.0;
    v3set(nl, 0, 0, 0); nl[k] = sg; cl[k] = sg * bs[k];
    dist = -best - r;
  }
  double nw[3], pw[3];                                       /* box -> sphere direction and closest box point, world */
  for (int i = 0; i < 3; i++) {
    nw[i] = bm[3 * i] * nl[0] + bm[3 * i + 1] * nl[1] + bm[3 * i + 2] * nl[2];
    pw[i] = bp[i] + bm[3 * i] * cl[0] + bm[3 * i + 1] * cl[1] + bm[3 * i + 2] * cl[2];
  }
  for (int i = 0; i < 3; i++) { o->normal[i] = -nw[i]; o->pos[i] = pw[i] + nw[i] * 0.5 * dist; }
  o->dist = dist;
  return 1;
}

/* derivative of the squared distance between the box [-s, s]^3 and the point p + t a (box frame), over 2 */
static double seg_box_dslope(const double *p, const double *a, const double *s, double t) {
  double g = 0;
  for (int i = 0; i < 3; i++) {
    double x = p[i] + t * a[i], e = fabs(x) - s[i];
    if (e > 0) g += a[i] * (x > 0 ? e : -e);
  }
  return g;
}

static int capsule_box(const double *cp, const double *ca, double r, double h, const double *bp, const double *bm, const double *bs,
                       double margin, ncon *o) {
  double d[3], p[3], a[3];
  v3sub(d, cp, bp);
  for (int i = 0; i < 3; i++) {
    p[i] = bm[i] * d[0] + bm[3 + i] * d[1] + bm[6 + i] * d[2];
    a[i] = bm[i] * ca[0] + bm[3 + i] * ca[1] + bm[6 + i] * ca[2];
  }
  /* breakpoints of the piecewise-linear slope: the segment crossing the six face planes; sorted with the two ends */
  double T[8]; int nt = 0;
  T[nt++] = -h;
  for (int i = 0; i < 3; i++) {
    if (fabs(a[i]) < MINVAL) continue;
    for (int sg = -1; sg <= 1; sg += 2) { double t = (sg * bs[i] - p[i]) / a[i]; if (t > -h && t < h) T[nt++] = t; }
  }
  T[nt++] = h;
  for (int i = 1; i < nt; i++) { double x = T[i]; int j = i; while (j > 0 && T[j - 1] > x) { T[j] = T[j - 1]; j--; } T[j] = x; }
  /* the slope g is continuous, piecewise linear and non-decreasing: the minimiser is where it crosses zero.  When it is zero
   * over a range (segment inside the box, or parallel to its nearest face) the middle of the range is taken; "zero" is
   * |g| <= tol with tol far above float32 rounding of g, so that the float32 kernel and this code take the same branch when
   * the exact slope at a breakpoint is zero (a segment entering the box has g = 0 at the entry breakpoint up to rounding) */
  double G[8], ts;
  const double bmax = bs[0] > bs[1] ? (bs[0] > bs[2] ? bs[0] : bs[2]) : (bs[1] > bs[2] ? bs[1] : bs[2]);
  const double tol = 1e-5 * (h + bmax);
  for (int k = 0; k < nt; k++) G[k] = seg_box_dslope(p, a, bs, T[k]);
  if (G[0] > tol) ts = T[0];
  else if (G[nt - 1] < -tol) ts = T[nt - 1];
  else {
    int i = 0;
    while (G[i] < -tol) i++;
    if (G[i] > tol) ts = T[i - 1] - G[i - 1] * (T[i] - T[i - 1]) / (G[i] - G[i - 1]);
    else { int e = i; while (e + 1 < nt && G[e + 1] <= tol) e++; ts = 0.5 * (T[i] + T[e]); }
  }
  int n = 0;
  double c[3];
  for (int i = 0; i < 3; i++) c[i] = cp[i] + ts * ca[i];
  n += sphere_box(c, r, bp, bm, bs, margin, p, o + n);
  double t2 = ts >= 0 ? -h : h;                              /* the far end of the segment */
  if (fabs(t2 - ts) > 1e-6 * (h > MINVAL ? h : 1.0)) {
    for (int i = 0; i < 3; i++) c[i] = cp[i] + t2 * ca[i];
    n += sphere_box(c, r, bp, bm, bs, margin, p, o + n);
  }
  return n;
}

static int box_box(const double *pa, const double *ma, const double *sa, const double *pb, const double *mb, const double *sb,
                   double margin, ncon *o) {
  double A[3][3], B[3][3], t[3], R[3][3], AR[3][3];
  for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) { A[i][k] = ma[3 * k + i]; B[i][k] = mb[3 * k + i]; }   /* axes */
  v3sub(t, pb, pa);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { R[i][j] = v3dot(A[i], B[j]); AR[i][j] = fabs(R[i][j]); }
  double best = -1e300; int bcode = -1; double bn[3] = {0, 0, 0};
  /* face axes of A (code 0-2) and B (3-5) */
  for (int i = 0; i < 3; i++) {
    double tl = v3dot(t, A[i]);
    double sep = fabs(tl) - (sa[i] + sb[0] * AR[i][0] + sb[1] * AR[i][1] + sb[2] * AR[i][2]);
    if (sep > best) { best = sep; bcode = i; for (int k = 0; k < 3; k++) bn[k] = (tl >= 0 ? 1.0 : -1.0) * A[i][k]; }
  }
  for (int j = 0; j < 3; j++) {
    double tl = v3dot(t, B[j]);
    double sep = fabs(tl) - (sa[0] * AR[0][j] + sa[1] * AR[1][j] + sa[2] * AR[2][j] + sb[j]);
    if (sep > best) { best = sep; bcode = 3 + j; for (int k = 0; k < 3; k++) bn[k] = (tl >= 0 ? 1.0 : -1.0) * B[j][k]; }
  }
  /* edge-edge axes A_i x B_j (code 6 + 3 i + j): chosen only when clearly less penetrating than the best face axis */
  double ebest = -1e300; int ecode = -1; double en[3] = {0, 0, 0};
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double L[3]; v3cross(L, A[i], B[j]);
    double ln = v3norm(L);
    if (ln < 1e-6) continue;
    v3set(L, L[0] / ln, L[1] / ln, L[2] / ln);
    double tl = v3dot(t, L), ra = 0, rb = 0;
    for (int k = 0; k < 3; k++) { ra += sa[k] * fabs(v3dot(A[k], L)); rb += sb[k] * fabs(v3dot(B[k], L)); }
    double sep = fabs(tl) - (ra + rb);
    if (sep > ebest) { ebest = sep; ecode = 6 + 3 * i + j; for (int k = 0; k < 3; k++) en[k] = (tl >= 0 ? 1.0 : -1.0) * L[k]; }
  }
  if (best > margin || ebest > margin) return 0;             /* a separating axis */
  int n = 0;
  if (ecode >= 0 && ebest > best + 1e-4 + 0.05 * fabs(best)) {
    /* edge-edge: the supporting edge of A (direction A_i, through its vertex farthest along +n) and of B (along -n) */
    int i = (ecode - 6) / 3, j = (ecode - 6) % 3;
    double qa[3], qb[3];
    v3cpy(qa, pa); v3cpy(qb, pb);
    for (int k = 0; k < 3; k++) {
      if (k != i) v3addscl(qa, A[k], (v3dot(en, A[k]) > 0 ? 1.0 : -1.0) * sa[k]);
      if (k != j) v3addscl(qb, B[k], (v3dot(en, B[k]) > 0 ? -1.0 : 1.0) * sb[k]);
    }
    /* closest points of the two lines qa + x A_i, qb + y B_j */
    double w[3]; v3sub(w, qa, qb);
    double b_ = R[i][j], d_ = v3dot(A[i], w), e_ = v3dot(B[j], w), den = 1.0 - b_ * b_;
    double x = den > 1e-12 ? (b_ * e_ - d_) / den : 0.0, y = den > 1e-12 ? (e_ - b_ * d_) / den : 0.0;
    x = clampd(x, -sa[i], sa[i]); y = clampd(y, -sb[j], sb[j]);
    for (int k = 0; k < 3; k++) {
      double ca_ = qa[k] + x * A[i][k], cb_ = qb[k] + y * B[j][k];
      o[0].pos[k] = 0.5 * (ca_ + cb_); o[0].normal[k] = en[k];
    }
    o[0].dist = ebest;
    return 1;
  }
  /* face contact (the manifold of ODE's dBoxBox, from which mjc_BoxBox descends): reference face = the face of the box owning
   * the best axis that looks at the other box; incident face = the other box's face most anti-parallel to it; the incident
   * rectangle is clipped (Sutherland-Hodgman) against the reference face's four side planes; the clipped polygon's vertices
   * within the margin of the reference plane are the contacts — half way between vertex and plane, normal = the axis */
  {
    const int ref_a = bcode < 3, ir = ref_a ? bcode : bcode - 3, i1 = (ir + 1) % 3, i2 = (ir + 2) % 3;
    const double (*Rf)[3] = ref_a ? A : B;
    const double (*Xi)[3] = ref_a ? B : A;
    const double *pr = ref_a ? pa : pb, *sr = ref_a ? sa : sb, *pi = ref_a ? pb : pa, *si = ref_a ? sb : sa;
    double nout[3]; for (int k = 0; k < 3; k++) nout[k] = ref_a ? bn[k] : -bn[k];
    int j = 0; double bd = -1;
    for (int k = 0; k < 3; k++) { double dk = fabs(v3dot(nout, Xi[k])); if (dk > bd) { bd = dk; j = k; } }
    const double sgn = v3dot(nout, Xi[j]) > 0 ? -1.0 : 1.0;
    const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    double poly[8][3], tmp[8][3];
    int np = 4;
    for (int v = 0; v < 4; v++) {
      const double su = (v == 0 || v == 3) ? si[j1] : -si[j1], sv = v < 2 ? si[j2] : -si[j2];
      for (int k = 0; k < 3; k++) poly[v][k] = pi[k] + sgn * si[j] * Xi[j][k] + su * Xi[j1][k] + sv * Xi[j2][k];
    }
    const double clip_tol = 1e-5 * (sr[i1] + sr[i2]);
    for (int e = 0; e < 4 && np > 0; e++) {
      const int tt = e < 2 ? i1 : i2;
      const double sg = (e & 1) ? -1.0 : 1.0;
      double rel[3]; v3sub(rel, poly[np - 1], pr);
      double fprev = sg * v3dot(rel, Rf[tt]) - sr[tt];
      int nq = 0;
      for (int v = 0; v < np; v++) {
        const double *x = poly[v], *xp = poly[v == 0 ? np - 1 : v - 1];
        v3sub(rel, x, pr);
        const double f = sg * v3dot(rel, Rf[tt]) - sr[tt];
        const int in = f <= clip_tol, inp = fprev <= clip_tol;
        if (in != inp && nq < 8) { const double u = fprev / (fprev - f); for (int k = 0; k < 3; k++) tmp[nq][k] = xp[k] + u * (x[k] - xp[k]); nq++; }
        if (in && nq < 8) { v3cpy(tmp[nq], x); nq++; }
        fprev = f;
      }
      np = nq;
      memcpy(poly, tmp, sizeof poly);
    }
    double kept[8][3];
    for (int v = 0; v < np && n < 8; v++) {
      double rel[3]; v3sub(rel, poly[v], pr);
      const double dist = v3dot(rel, nout) - sr[ir];
      if (dist > margin) continue;
      int dup = 0;
      for (int q = 0; q < n; q++) {
        double e[3]; v3sub(e, kept[q], poly[v]);
        const double en_ = v3dot(e, nout);
        v3addscl(e, nout, -en_);
        if (v3dot(e, e) <= clip_tol * clip_tol) dup = 1;
      }
      if (dup) continue;
      v3cpy(kept[n], poly[v]);
      for (int k = 0; k < 3; k++) { o[n].pos[k] = poly[v][k] - 0.5 * dist * nout[k]; o[n].normal[k] = bn[k]; }
      o[n].dist = dist;
      n++;
    }
  }
  return n;
}

/* test hook: one pair function on raw geometry.  kind 0 capsule-capsule, 1 capsule-box, 2 box-box.  in: geom 1 then geom 2
 * (capsule: centre3 axis3 radius halflength; box: centre3 rowmajor-rotation9 halfsizes3), then the margin.
 * out: [n, then n x (pos3 normal3 dist)] */
int om_narrow_phase(int kind, const double *in, double *out) {
  ncon c[8]; int n = 0;
  if (kind == 0) n = capsule_capsule(in, in + 3, in[6], in[7], in + 8, in + 11, in[14], in[15], in[16], c);
  else if (kind == 1) n = capsule_box(in, in + 3, in[6], in[7], in + 8, in + 11, in + 20, in[23], c);
  else n = box_box(in, in + 3, in + 12, in + 15, in + 18, in + 27, in[30], c);
  out[0] = n;
  for (int i = 0; i < n; i++) { memcpy(out + 1 + 7 * i, c[i].pos, 24); memcpy(out + 4 + 7 * i, c[i].normal, 24); out[7 + 7 * i] = c[i].dist; }
  return n;
}

static void geom_world(const om_model *m, const om_data *d, int b, double *gp, double *gm) {
  double t[3]; m3mulv(t, d->xmat[b], m->gpos[b]); v3add(gp, d->xpos[b], t);
  m3mul(gm, d->xmat[b], m->gmat[b]);
}

/* body-body contacts appended behind the floor contacts; touch[] (the env's termination rule) stays floor-only */
static void collide_bodies(const om_model *m, om_data *d) {
  d->nself = 0; d->nself_dropped = 0;
  if (!m->self_collision) return;
  int first = d->ncon;
  for (int q = 0; q < m->npair; q++) {
    int b1 = m->pair_b1[q], b2 = m->pair_b2[q];
    double p1[3], m1[9], p2[3], m2[9], dc[3];
    geom_world(m, d, b1, p1, m1); geom_world(m, d, b2, p2, m2);
    v3sub(dc, p2, p1);
    if (v3norm(dc) > m->brad[b1] + m->brad[b2] + m->margin) continue;     /* bounding spheres */
    ncon out[8]; int n = 0;
    double a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]};
    if (m->gtype[b1] != OM_GEOM_BOX && m->gtype[b2] != OM_GEOM_BOX)     /* capsules and spheres (zero half length) */
      n = capsule_capsule(p1, a1, m->gsize[b1][0], m->gsize[b1][1], p2, a2, m->gsize[b2][0], m->gsize[b2][1], m->margin, out);
    else if (m->gtype[b1] != OM_GEOM_BOX)
      n = capsule_box(p1, a1, m->gsize[b1][0], m->gsize[b1][1], p2, m2, m->gsize[b2], m->margin, out);
    else
      n = box_box(p1, m1, m->gsize[b1], p2, m2, m->gsize[b2], m->margin, out);
    for (int i = 0; i < n && d->ncon < MAXCON; i++) {
      int c = d->ncon++;
      d->con_body1[c] = b1; d->con_body[c] = b2; d->con_dist[c] = out[i].dist;
      v3cpy(d->con_pos[c], out[i].pos);
      memset(d->con_frame[c], 0, 9 * sizeof(double)); v3cpy(d->con_frame[c], out[i].normal);
      make_frame(d->con_frame[c]);
    }
  }
  d->nself = d->ncon - first;
  if (m->max_self > 0 && d->nself > m->max_self) {           /* keep the deepest max_self (stable: ties keep pair order) */
    int keep = m->max_self, idx[MAXCON], ns = d->nself;
    for (int i = 0; i < ns; i++) idx[i] = first + i;
    for (int i = 1; i < ns; i++) { int x = idx[i], j = i; while (j > 0 && d->con_dist[idx[j - 1]] > d->con_dist[x]) { idx[j] = idx[j - 1]; j--; } idx[j] = x; }
    for (int i = 1; i < keep; i++) { int x = idx[i], j = i; while (j > 0 && idx[j - 1] > x) { idx[j] = idx[j - 1]; j--; } idx[j] = x; }   /* back in pair order */
    for (int i = 0; i < keep; i++) {
      int sidx = idx[i], c = first + i;
      d->con_body1[c] = d->con_body1[sidx]; d->con_body[c] = d->con_body[sidx]; d->con_dist[c] = d->con_dist[sidx];
      v3cpy(d->con_pos[c], d->con_pos[sidx]); memcpy(d->con_frame[c], d->con_frame[sidx], 9 * sizeof(double));
    }
    d->nself_dropped = ns - keep; d->nself = keep; d->ncon = first + keep;
  }
}

/* ------------------------------------------------------------------ constraints */
static double impedance(const double *si, double pos, double margin) {
  if (si[0] == si[1] || si[2] <= MINVAL) return 0.5 * (si[0] + si[1]);
  double x = (pos - margin) / si[2];
  if (x < 0) x = -x;
  if (x >= 1 || x <= 0) return x >= 1 ? si[1] : si[0];
  double y;
  if (si[4] == 1) y = x;
  else if (x <= si[3]) y = pow(x, si[4]) / pow(si[3], si[4] - 1);
  else y = 1 - pow(1 - x, si[4]) / pow(1 - si[3], si[4] - 1);
  return si[0] + y * (si[1] - si[0]);
}

static void make_constraints(const om_model *m, om_data *d) {
  int nv = m->nv, r = 0;
  /* joint limits (hinges) */
  for (int i = 6; i < nv; i++) {
    if (!m->limited[i]) continue;
    double q = d->qpos[i + 1];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side < 0 ? q - m->range[i][0] : m->range[i][1] - q;
      if (dist < 0) {
        memset(d->J + (size_t)r * nv, 0, sizeof(double) * nv);
        d->J[(size_t)r * nv + i] = -side;
        d->epos[r] = dist; d->emargin[r] = 0; d->ediag[r] = m->dof_invw[i];
        r++;
      }
    }
  }
  int first_contact_row = r;
  for (int c = 0; c < d->ncon; c++) {
    /* relative velocity of the contact point: body 2 minus body 1 (floor contacts: body 1 = world, no Jacobian, weight 0) */
    double fn[OM_MAXV], ft1[OM_MAXV], ft2[OM_MAXV];
    memset(fn, 0, sizeof(double) * nv); memset(ft1, 0, sizeof(double) * nv); memset(ft2, 0, sizeof(double) * nv);
    double w = 0;
    for (int side = 0; side < 2; side++) {
      int b = side ? d->con_body[c] : d->con_body1[c];
      if (b < 0) continue;
      double sgn = side ? 1.0 : -1.0;
      for (int a = 0; a < m->chain_len[b]; a++) {
        double jv[3], jw[3]; dof_jac(m, d, m->chain[b][a], d->con_pos[c], jv, jw);
        int dof = m->chain[b][a];
        fn[dof] += sgn * v3dot(d->con_frame[c], jv); ft1[dof] += sgn * v3dot(d->con_frame[c] + 3, jv); ft2[dof] += sgn * v3dot(d->con_frame[c] + 6, jv);
      }
      w += m->body_invw[b][0];
    }
    for (int k = 0; k < 4; k++) {
      double *row = d->J + (size_t)r * nv;
      const double *jt = k < 2 ? ft1 : ft2;
      double sg = (k & 1) ? -m->mu : m->mu;
      for (int a = 0; a < nv; a++) row[a] = fn[a] + sg * jt[a];
      d->epos[r] = d->con_dist[c]; d->emargin[r] = m->margin;
      d->ediag[r] = w * (1 + m->mu * m->mu);
      r++;
    }
  }
  d->nefc = r;
  /* impedance, R, D, aref */
  double dmax = m->solimp[1];
  double tc = m->solref[0] < 2 * m->dt ? 2 * m->dt : m->solref[0];   /* refsafe */
  double K = 1.0 / (dmax * dmax * tc * tc * m->solref[1] * m->solref[1]);
  double B = 2.0 / (dmax * tc);
  for (int i = 0; i < r; i++) {
    double imp = impedance(m->solimp, d->epos[i], d->emargin[i]);
    double R = (1 - imp) * d->ediag[i] / imp;
    d->eR[i] = R < MINVAL ? MINVAL : R;
    double vel = 0;
    const double *row = d->J + (size_t)i * nv;
    for (int k = 0; k < nv; k++) vel += row[k] * d->qvel[k];
    d->earef[i] = -B * vel - K * imp * (d->epos[i] - d->emargin[i]);
  }
  for (int i = first_contact_row; i < r; i += 4) {
    double R1 = d->eR[i] / (m->impratio < MINVAL ? MINVAL : m->impratio);
    double mu_reg = m->mu * sqrt(R1 / d->eR[i]);
    double Rpy = 2 * mu_reg * mu_reg * d->eR[i];
    for (int k = 0; k < 4; k++) d->eR[i + k] = Rpy;
  }
  for (int i = 0; i < r; i++) d->eD[i] = 1.0 / d->eR[i];
}

/* cost, gradient pieces at acceleration a: jar = J a - aref */
static double eval_cost(const om_model *m, om_data *d, const double *a, double *jar) {
  int nv = m->nv;
  double cost = 0;
  /* Gauss: 0.5 (a - as)^T M (a - as) */
  for (int i = 0; i < nv; i++) {
    double s = 0, di = a[i] - d->qacc_smooth[i];
    for (int k = 0; k < nv; k++) s += d->M[i * nv + k] * (a[k] - d->qacc_smooth[k]);
    cost += 0.5 * di * s;
  }
  for (int r = 0; r < d->nefc; r++) {
    const double *row = d->J + (size_t)r * nv;
    double s = -d->earef[r];
    for (int k = 0; k < nv; k++) s += row[k] * a[k];
    jar[r] = s;
    if (s < 0) cost += 0.5 * d->eD[r] * s * s;
  }
  return cost;
}

/* ---- MuJoCo's line search (OM_LS_MUJOCO): a restatement of PrimalSearch of MuJoCo's engine_solver.c (the Newton / CG solvers' 1-D
 * search; reached from the reference through mujoco.mj_step, humanoid_env.py:450; MuJoCo 3.x; "parity unpinned" like the rest of
 * mj_step: MJ-(V9b) of oracle.h).  phi(alpha) = cost(a + alpha dir) is convex and piecewise quadratic; a point carries alpha, cost and
 * the first two derivatives.
 *   gtol = tolerance * ls_tolerance * |dir| / scale           (scale = 1 / (meaninertia max(1, nv)): the solver's own scaling)
 *   p0 = phi at 0;  p1 = phi at the Newton point of p0;  p1 = p0 if that raised the cost;  done if |p1'| < gtol
 *   phase 1 (one-sided): while the slope keeps its sign (p1' dir <= -gtol): p2 = p1, p1 = Newton point of p1; done if |p1'| < gtol
 *   phase 2 (bracket p1 | p2, slopes of opposite sign): candidates = the Newton points of both ends and the midpoint; any candidate with
 *            |slope| < gtol ends the search; otherwise each end moves to the candidate with the slope of its own sign that is closest
 *            to zero; no end moved -> the midpoint is returned
 *   every evaluation counts against ls_iterations (50); when they run out: the lower end of the bracket if it improves on p0, else 0.
 * alpha = 0 ends the solver's iteration ("no improvement").  An inexact search: the Newton iterates differ from the exact search's by
 * up to gtol in the slope — 1e-10 |dir| meaninertia nv — which is why the two searches give the same solver output to ~1e-9
 * (DESIGN.md 4d has the measured table). */
typedef struct { double alpha, cost, d1, d2; } ls_pnt;
typedef struct { const om_data *d; int ne; double q1, q2; int evals; } ls_ctx;
static void ls_point(ls_ctx *c, double al, ls_pnt *p) {
  const om_data *d = c->d;
  double cost = al * c->q1 + 0.5 * al * al * c->q2, d1 = c->q1 + al * c->q2, d2 = c->q2;
  for (int r = 0; r < c->ne; r++) {
    const double x = d->ejar[r] + al * d->ejd[r];
    if (x < 0) { cost += 0.5 * d->eD[r] * x * x; d1 += d->eD[r] * x * d->ejd[r]; d2 += d->eD[r] * d->ejd[r] * d->ejd[r]; }
  }
  p->alpha = al; p->cost = cost; p->d1 = d1; p->d2 = d2; c->evals++;
}
static int ls_update_bracket(ls_ctx *c, ls_pnt *p, const ls_pnt cand[3], ls_pnt *pnext) {
  int flag = 0;
  for (int i = 0; i < 3; i++) {
    if (p->d1 < 0 && cand[i].d1 < 0 && p->d1 < cand[i].d1) { *p = cand[i]; flag = 1; }
    else if (p->d1 > 0 && cand[i].d1 > 0 && p->d1 > cand[i].d1) { *p = cand[i]; flag = 2; }
  }
  if (flag) ls_point(c, p->alpha - p->d1 / p->d2, pnext);
  return flag;
}
static double mujoco_linesearch(ls_ctx *c, double snorm, double gtol, int ls_iterations, int *exhausted) {
  ls_pnt p0, p1, p2, pmid, p1next, p2next;
  *exhausted = 0;
  if (snorm < 1e-15) return 0;                               /* mjMINVAL: search vector too small */
  ls_point(c, 0, &p0);
  ls_point(c, p0.alpha - p0.d1 / p0.d2, &p1);
  if (p0.cost < p1.cost) p1 = p0;
  if (fabs(p1.d1) < gtol) return p1.alpha;
  const int dir = p1.d1 < 0 ? 1 : -1;
  int p2update = 0;
  p2 = p1;
  while (p1.d1 * dir <= -gtol && c->evals < ls_iterations) {
    p2 = p1; p2update = 1;
    ls_point(c, p1.alpha - p1.d1 / p1.d2, &p1);
    if (fabs(p1.d1) < gtol) return p1.alpha;
  }
  if (c->evals >= ls_iterations) { *exhausted = 1; return p1.alpha; }
  if (!p2update) return p1.alpha;
  p2next = p1;
  ls_point(c, p1.alpha - p1.d1 / p1.d2, &p1next);
  while (c->evals < ls_iterations) {
    ls_point(c, 0.5 * (p1.alpha + p2.alpha), &pmid);
    const ls_pnt cand[3] = {p1next, p2next, pmid};
    for (int i = 0; i < 3; i++) if (fabs(cand[i].d1) < gtol) return cand[i].alpha;
    const int b1 = ls_update_bracket(c, &p1, cand, &p1next), b2 = ls_update_bracket(c, &p2, cand, &p2next);
    if (!b1 && !b2) return pmid.alpha;
  }
  *exhausted = 1;
  if (p1.cost <= p2.cost && p1.cost < p0.cost) return p1.alpha;
  if (p2.cost <= p1.cost && p2.cost < p0.cost) return p2.alpha;
  return 0;
}

static void solve_constraints(const om_model *m, om_data *d) {
  int nv = m->nv, ne = d->nefc;
  double a[OM_MAXV], grad[OM_MAXV], dir[OM_MAXV], Ma[OM_MAXV];
  d->solver_iter = 0;
  if (ne == 0) {
    memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
    memset(d->qfrc_constraint, 0, sizeof(double) * nv);
    return;
  }
  /* warm start: the better of qacc_warmstart and qacc_smooth */
  double cw = eval_cost(m, d, d->warm, d->ejar), cs = eval_cost(m, d, d->qacc_smooth, d->ejar);
  memcpy(a, cw < cs ? d->warm : d->qacc_smooth, sizeof(double) * nv);
  /* Newton with exact line search.  Two termination rules (om_model_set_solver):
   *   OM_SOLVER_MUJOCO     mj_solPrimal's (MuJoCo engine_solver.c; reached from the reference through mujoco.mj_step,
   *                        humanoid_env.py:450): after every iteration
   *                          improvement = scale (cost_before - cost_after), gradient = scale |grad(a_after)|,
   *                          scale = 1 / (meaninertia max(1, nv));  stop when either is < opt.tolerance (1e-8), or after
   *                          opt.iterations (100) iterations, or when the line search finds no step (alpha = 0)
   *   OM_SOLVER_CONVERGED  to the rounding level of the forces (the triage reference: what any faithful solver converges to) */
  const int mj = m->solver_mode == OM_SOLVER_MUJOCO;
  const double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  const int maxit = mj ? m->iterations : 100;
  double cost = cw < cs ? cw : cs, improvement = 0;
  for (int it = 0; it < maxit; it++) {
    eval_cost(m, d, a, d->ejar);
    /* gradient = M (a - as) + J^T (D jar)_- */
    for (int i = 0; i < nv; i++) {
      double s = 0;
      for (int k = 0; k < nv; k++) s += d->M[i * nv + k] * (a[k] - d->qacc_smooth[k]);
      grad[i] = s;
    }
    for (int r = 0; r < ne; r++) {
      if (d->ejar[r] >= 0) { d->eforce[r] = 0; continue; }
      const double *row = d->J + (size_t)r * nv;
      double Dj = d->eD[r] * d->ejar[r];
      d->eforce[r] = -Dj;
      for (int i = 0; i < nv; i++) grad[i] += row[i] * Dj;
    }
    double gn = 0, fn = 0;
    for (int i = 0; i < nv; i++) {
      double fs = d->qfrc_act[i] - d->bias[i];
      gn += grad[i] * grad[i]; fn += fs * fs;
    }
    for (int r = 0; r < ne; r++) fn += d->eforce[r] * d->eforce[r];
    if (mj) { if (it > 0 && (improvement < m->tolerance || scale * sqrt(gn) < m->tolerance)) break; }
    else if (sqrt(gn) <= 1e-13 * (1.0 + sqrt(fn))) break;   /* gradient at rounding level of the forces */
    /* Hessian = M + J^T D_active J */
    memcpy(d->H, d->M, sizeof(double) * nv * nv);
    for (int r = 0; r < ne; r++) {
      if (d->ejar[r] >= 0) continue;
      const double *row = d->J + (size_t)r * nv;
      for (int i = 0; i < nv; i++) {
        if (row[i] == 0) continue;
        double ri = d->eD[r] * row[i];
        for (int k = 0; k <= i; k++) d->H[i * nv + k] += ri * row[k];
      }
    }
    for (int i = 0; i < nv; i++) for (int k = i + 1; k < nv; k++) d->H[i * nv + k] = d->H[k * nv + i];
    if (chol_factor(d->H, nv)) break;
    for (int i = 0; i < nv; i++) dir[i] = -grad[i];
    chol_solve(d->H, nv, dir);
    /* exact line search on the convex piecewise-quadratic phi(alpha) */
    for (int i = 0; i < nv; i++) {
      double s = 0;
      for (int k = 0; k < nv; k++) s += d->M[i * nv + k] * dir[k];
      Ma[i] = s;
    }
    double q1 = 0, q2 = 0;                                  /* phi_gauss' (0) and phi_gauss'' */
    for (int i = 0; i < nv; i++) {
      double s = 0;
      for (int k = 0; k < nv; k++) s += d->M[i * nv + k] * (a[k] - d->qacc_smooth[k]);
      q1 += dir[i] * s; q2 += dir[i] * Ma[i];
    }
    for (int r = 0; r < ne; r++) {
      const double *row = d->J + (size_t)r * nv;
      double s = 0;
      for (int k = 0; k < nv; k++) s += row[k] * dir[k];
      d->ejd[r] = s;
    }
#define DPHI(al, d1, d2) do { d1 = q1 + (al) * q2; d2 = q2; \
      for (int r_ = 0; r_ < ne; r_++) { double x_ = d->ejar[r_] + (al) * d->ejd[r_]; \
        if (x_ < 0) { d1 += d->eD[r_] * x_ * d->ejd[r_]; d2 += d->eD[r_] * d->ejd[r_] * d->ejd[r_]; } } } while (0)
    double lo = 0, hi = 1, d1, d2, dlo;
    DPHI(0.0, dlo, d2);
    if (dlo >= 0) break;                                    /* not a descent direction: converged (MuJoCo: alpha = 0) */
    d->solver_iter = it + 1;
    if (m->ls_mode == OM_LS_MUJOCO) {
      ls_ctx lc = {d, ne, q1, q2, 0};
      double sn = 0;
      for (int i = 0; i < nv; i++) sn += dir[i] * dir[i];
      int exhausted = 0;
      const double al_mj = mujoco_linesearch(&lc, sqrt(sn), m->tolerance * m->ls_tolerance * sqrt(sn) / scale, m->ls_iterations, &exhausted);
      d->ls_evals += lc.evals; d->ls_calls++; d->ls_exhausted += exhausted;
      if (al_mj == 0) { d->solver_iter = it; break; }       /* "no improvement": mj_solPrimal leaves its loop */
      for (int i = 0; i < nv; i++) a[i] += al_mj * dir[i];
      const double newcost = eval_cost(m, d, a, d->ejar);
      improvement = scale * (cost - newcost);
      cost = newcost;
      continue;
    }
    DPHI(hi, d1, d2);
    int guard = 0;
    while (d1 < 0 && guard++ < 60) { lo = hi; hi *= 2; DPHI(hi, d1, d2); }
    double al = hi;
    if (d1 != 0) {
      al = 1.0 <= hi && 1.0 >= lo ? 1.0 : 0.5 * (lo + hi);
      for (int ls = 0; ls < 100; ls++) {
        DPHI(al, d1, d2);
        if (fabs(d1) < 1e-14 * (fabs(q1) + 1e-300)) break;
        if (d1 < 0) lo = al; else hi = al;
        double nx = al - d1 / d2;
        if (!(nx > lo && nx < hi)) nx = 0.5 * (lo + hi);
        if (nx == al || hi - lo < 1e-16 * hi) break;
        al = nx;
      }
    }
    double step = 0;
    for (int i = 0; i < nv; i++) { a[i] += al * dir[i]; step += al * dir[i] * al * dir[i]; }
    if (mj) {
      const double newcost = eval_cost(m, d, a, d->ejar);
      improvement = scale * (cost - newcost);
      cost = newcost;
    } else if (sqrt(step) < 1e-15) break;
  }
  eval_cost(m, d, a, d->ejar);
  memcpy(d->qacc, a, sizeof(double) * nv);
  memset(d->qfrc_constraint, 0, sizeof(double) * nv);
  for (int r = 0; r < ne; r++) {
    d->eforce[r] = d->ejar[r] < 0 ? -d->eD[r] * d->ejar[r] : 0;
    if (d->eforce[r] == 0) continue;
    const double *row = d->J + (size_t)r * nv;
    for (int k = 0; k < nv; k++) d->qfrc_constraint[k] += row[k] * d->eforce[r];
  }
}

void om_forward(const om_model *m, om_data *d) {
  int nv = m->nv;
  memcpy(d->qpos_fwd, d->qpos, sizeof d->qpos_fwd); memcpy(d->qvel_fwd, d->qvel, sizeof d->qvel_fwd);
  om_kinematics(m, d);
  compute_M(m, d);
  collide(m, d);
  collide_bodies(m, d);
  compute_bias(m, d);                                       /* velocity stage: sensors + bias */
  make_constraints(m, d);
  memset(d->qfrc_act, 0, sizeof(double) * nv);
  for (int i = 0; i < m->nu; i++) d->qfrc_act[m->act_dof[i]] = d->ctrl[i];   /* gear 1, unclamped */
  memcpy(d->work, d->M, sizeof(double) * nv * nv);
  chol_factor(d->work, nv);
  for (int i = 0; i < nv; i++) d->qacc_smooth[i] = d->qfrc_act[i] - d->bias[i];
  chol_solve(d->work, nv, d->qacc_smooth);
  solve_constraints(m, d);
}

/* mj_checkPos / mj_checkVel / mj_checkAcc: a NaN or |x| > mjMAXVAL (1e10) triggers MuJoCo's
 * autoreset (mj_resetData: qpos = qpos0, everything else zero) and bumps a warning counter */
static int is_bad(const double *x, int n) {
  for (int i = 0; i < n; i++) if (x[i] != x[i] || x[i] > 1e10 || x[i] < -1e10) return 1;
  return 0;
}
static void reset_data(const om_model *m, om_data *d) {
  memset(d->qpos, 0, sizeof d->qpos); memset(d->qvel, 0, sizeof d->qvel);
  memset(d->qacc, 0, sizeof d->qacc); memset(d->warm, 0, sizeof d->warm); memset(d->ctrl, 0, sizeof d->ctrl);
  v3cpy(d->qpos, m->body_pos[0]); d->qpos[3] = 1;
  d->nwarn++;
}

void om_step(const om_model *m, om_data *d) {
  int nv = m->nv;
  if (is_bad(d->qpos, m->nq)) reset_data(m, d);
  if (is_bad(d->qvel, nv)) reset_data(m, d);
  om_forward(m, d);
  if (is_bad(d->qacc, nv)) { reset_data(m, d); om_forward(m, d); }
  /* semi-implicit Euler (no joint damping => no implicit solve) */
  for (int i = 0; i < nv; i++) d->qvel[i] += m->dt * d->qacc[i];
  for (int k = 0; k < 3; k++) d->qpos[k] += m->dt * d->qvel[k];
  {
    double ax[3] = {d->qvel[3], d->qvel[4], d->qvel[5]}, qr[4];
    double ang = m->dt * v3normalize(ax);
    axisangle2q(qr, ax, ang);
    qnormalize(d->qpos + 3);
    qmul(d->qpos + 3, d->qpos + 3, qr);
  }
  for (int i = 6; i < nv; i++) d->qpos[i + 1] += m->dt * d->qvel[i];
  memcpy(d->warm, d->qacc, sizeof(double) * nv);
}

/* ------------------------------------------------------------------ controllers */
void om_spd_torque(const om_model *m, const om_data *d, const double *action, double *tau) {
  int nv = m->nv, nu = m->nu;
  double dt = m->dt;
  double *A = d->work2;                                   /* per-data scratch (no malloc in the hot loop) */
  double kp[OM_MAXV] = {0}, kd[OM_MAXV] = {0}, perr[OM_MAXV] = {0}, rhs[OM_MAXV];
  memcpy(A, d->M, sizeof(double) * nv * nv);
  for (int i = 0; i < nu; i++) {
    int dof = m->act_dof[i];
    kp[dof] = m->kp[i]; kd[dof] = m->kd[i];
    double target = action[i] * m->ascale[i] + m->aoffset[i];
    perr[dof] = d->qpos[dof + 1] + d->qvel[dof] * dt - target;
  }
  for (int i = 0; i < nv; i++) {
    A[i * nv + i] += kd[i] * dt;
    rhs[i] = -d->bias[i] - kp[i] * perr[i] - kd[i] * d->qvel[i];
  }
  chol_factor(A, nv);
  chol_solve(A, nv, rhs);                                   /* rhs = q_accel */
  for (int i = 0; i < nu; i++) {
    int dof = m->act_dof[i];
    double t = -m->kp[i] * perr[dof] - m->kd[i] * (d->qvel[dof] + rhs[dof] * dt);
    tau[i] = t > m->tlim[i] ? m->tlim[i] : (t < -m->tlim[i] ? -m->tlim[i] : t);
  }
}

void om_ctrl_torque(const om_model *m, const om_data *d, int mode, double power_scale, const double *action, double *tau) {
  if (mode == 0) { om_spd_torque(m, d, action, tau); return; }
  om_data *ds = (om_data *)d;                               /* simple_pid keeps controller state in the data */
  const double dtp = d->pid_dt > 0 ? d->pid_dt : 15.0 * m->dt;
  for (int i = 0; i < m->nu; i++) {
    int dof = m->act_dof[i];
    double t, lim = m->tlim[i];
    if (mode == 4) { tau[i] = action[i]; continue; }        /* `default`: ctrl = action (humanoid_env.py:409-410) */
    if (mode == 1) {                                        /* PIDController with zero integral gain */
      double target = action[i] * m->ascale[i] + m->aoffset[i];
      t = -m->kp[i] * (d->qpos[dof + 1] - target) - m->kd[i] * d->qvel[dof];
    } else if (mode == 3) {                                 /* SimplePID: gains kp, kd as passed (the env passes jkp/10, jkd/10), ki = 1 */
      double err = action[i] * m->ascale[i] + m->aoffset[i] - d->qpos[dof + 1];
      double derr = ds->pid_started ? err - ds->pid_e[i] : 0.0;
      double in = ds->pid_i[i] + 1.0 * err * dtp;
      in = in > lim ? lim : (in < -lim ? -lim : in);
      t = m->kp[i] * err + in + m->kd[i] * derr / dtp;
      ds->pid_i[i] = in; ds->pid_e[i] = err;
    } else {                                                /* SimpleTorqueController */
      t = action[i] * power_scale * lim;
    }
    tau[i] = t > lim ? lim : (t < -lim ? -lim : t);
  }
  if (mode == 3) ds->pid_started = 1;
}
void om_set_pid_dt(om_data *d, double dt) { d->pid_dt = dt; }

/* ------------------------------------------------------------------ observations */
/* wxyz helpers of np_transform_utils.py */
static void npt_quat_rotate(const double *q, const double *v, double *r) {
  double qw = q[0]; const double *qv = q + 1;
  double a = 2.0 * qw * qw - 1.0, cr[3]; v3cross(cr, qv, v);
  double dt = v3dot(qv, v);
  for (int k = 0; k < 3; k++) r[k] = v[k] * a + cr[k] * qw * 2.0 + qv[k] * dt * 2.0;
}
static void npt_quat_mul(const double *a, const double *b, double *r) {
  double w1 = a[0], x1 = a[1], y1 = a[2], z1 = a[3], w2 = b[0], x2 = b[1], y2 = b[2], z2 = b[3];
  double ww = (z1 + x1) * (x2 + y2), yy = (w1 - y1) * (w2 + z2), zz = (w1 + y1) * (w2 - z2);
  double xx = ww + yy + zz, qq = 0.5 * (xx + (z1 - x1) * (x2 - y2));
  r[0] = qq - ww + (z1 - y1) * (y2 - z2); r[1] = qq - xx + (x1 + w1) * (x2 + w2);
  r[2] = qq - yy + (w1 - x1) * (y2 + z2); r[3] = qq - zz + (z1 + y1) * (w2 - x2);
}
static void heading_quat_inv(const double *root_quat, double *hq) {
  static const double base_conj[4] = {0.5, -0.5, -0.5, -0.5};
  double q[4], ex[3] = {1, 0, 0}, rd[3];
  npt_quat_mul(root_quat, base_conj, q);                    /* remove_base_rot */
  npt_quat_rotate(q, ex, rd);
  double heading = atan2(rd[1], rd[0]);
  double th = -heading / 2;
  hq[0] = cos(th); hq[1] = 0; hq[2] = 0; hq[3] = sin(th);
  double n = sqrt(hq[0] * hq[0] + hq[3] * hq[3]); if (n < 1e-9) n = 1e-9;
  hq[0] /= n; hq[3] /= n;
}
static int obs_common(int nb, const double *xpos, const double *xquat, int root_h, const double *hq, float *obs) {
  int o = 0;
  if (root_h) obs[o++] = (float)xpos[2];
  for (int b = 1; b < nb; b++) {
    double l[3], r[3]; v3sub(l, xpos + 3 * b, xpos); npt_quat_rotate(hq, l, r);
    for (int k = 0; k < 3; k++) obs[o++] = (float)r[k];
  }
  for (int b = 0; b < nb; b++) {
    double lq[4], t[3], n[3], ex[3] = {1, 0, 0}, ez[3] = {0, 0, 1};
    npt_quat_mul(hq, xquat + 4 * b, lq);
    npt_quat_rotate(lq, ex, t); npt_quat_rotate(lq, ez, n);
    for (int k = 0; k < 3; k++) obs[o++] = (float)t[k];
    for (int k = 0; k < 3; k++) obs[o++] = (float)n[k];
  }
  return o;
}
void om_obs_v1(int nb, const double *qpos, const double *qvel, const double *xpos, const double *xquat, int root_h, float *obs) {
  (void)qpos;
  double hq[4], r[3];
  heading_quat_inv(xquat, hq);
  int o = obs_common(nb, xpos, xquat, root_h, hq, obs);
  npt_quat_rotate(hq, qvel, r); for (int k = 0; k < 3; k++) obs[o++] = (float)r[k];
  npt_quat_rotate(hq, qvel + 3, r); for (int k = 0; k < 3; k++) obs[o++] = (float)r[k];
  for (int i = 6; i < 6 + 3 * (nb - 1); i++) obs[o++] = (float)qvel[i];
}
void om_obs_v2(int nb, const double *xpos, const double *xquat, const double *linvel, const double *angvel, int root_h, float *obs) {
  double hq[4], r[3];
  heading_quat_inv(xquat, hq);
  int o = obs_common(nb, xpos, xquat, root_h, hq, obs);
  for (int b = 0; b < nb; b++) { npt_quat_rotate(hq, linvel + 3 * b, r); for (int k = 0; k < 3; k++) obs[o++] = (float)r[k]; }
  for (int b = 0; b < nb; b++) { npt_quat_rotate(hq, angvel + 3 * b, r); for (int k = 0; k < 3; k++) obs[o++] = (float)r[k]; }
}

/* ------------------------------------------------------------------ env layer */
struct om_env {
  const om_model *m;
  om_data *d;
  om_env_cfg cfg;
  int cur_t;
  double tar;            /* tar_speed or tar_height (reach: tar_pos x) */
  double tar_yz[2];      /* reach: tar_pos y, z */
  double change_steps;
  int recovery_counter;
  double prev_root_pos[3];
};

om_env *om_env_create(const om_model *m, const om_env_cfg *cfg) {
  om_env *e = (om_env *)calloc(1, sizeof *e);
  e->m = m; e->d = om_data_create(m); e->cfg = *cfg;
  e->d->pid_dt = m->dt * cfg->control_freq_inv;
  return e;
}
void om_env_destroy(om_env *e) { if (e) { om_data_destroy(e->d); free(e); } }
om_data *om_env_data(om_env *e) { return e->d; }
int om_env_obs_size(const om_env *e) {
  int nb = e->m->nbody, nd = 3 * (nb - 1);
  int n = (e->cfg.root_height_obs ? 1 : 0) + nd + (e->cfg.self_obs_v == 1 ? nb * 6 + 6 + nd : nb * 12);
  if (e->cfg.task == OM_TASK_SPEED || e->cfg.task == OM_TASK_REACH) n += 3;
  if (e->cfg.task == OM_TASK_GETUP) n += 1;
  return n;
}
void om_env_get_task(const om_env *e, double *o) {
  o[0] = e->cur_t; o[1] = e->tar; o[2] = e->change_steps; o[3] = e->recovery_counter;
  v3cpy(o + 4, e->prev_root_pos); o[7] = e->tar_yz[0]; o[8] = e->tar_yz[1];
}
void om_env_set_task(om_env *e, const double *i) {
  e->cur_t = (int)i[0]; e->tar = i[1]; e->change_steps = i[2]; e->recovery_counter = (int)i[3];
  v3cpy(e->prev_root_pos, i + 4); e->tar_yz[0] = i[7]; e->tar_yz[1] = i[8];
}

static void env_reset_task(om_env *e, const double *u) {
  const om_env_cfg *c = &e->cfg;
  if (c->task == OM_TASK_SPEED) {
    e->tar = (c->tar_speed_max - c->tar_speed_min) * u[0] + c->tar_speed_min;
    int ch = c->speed_change_min + (int)floor(u[1] * (c->speed_change_max - c->speed_change_min));
    e->change_steps = e->cur_t + ch;
  } else if (c->task == OM_TASK_GETUP) {
    e->tar = (c->tar_height_max - c->tar_height_min) * u[0] + c->tar_height_min;
    int ch = c->height_change_min + (int)floor(u[1] * (c->height_change_max - c->height_change_min));
    e->change_steps = e->cur_t + ch;
  } else if (c->task == OM_TASK_REACH) {                     /* humanoid_reach.py:81-92 */
    e->tar = c->tar_dist_max * (2.0 * u[0] - 1.0); e->tar_yz[0] = c->tar_dist_max * (2.0 * u[1] - 1.0);
    e->tar_yz[1] = (c->tar_height_max - c->tar_height_min) * u[2] + c->tar_height_min;
    int ch = c->height_change_min + (int)floor(u[3] * (c->height_change_max - c->height_change_min));
    e->change_steps = e->cur_t + ch;
  }
}

static void env_obs(om_env *e, float *obs) {
  const om_model *m = e->m; om_data *d = e->d;
  om_kinematics(m, d);
  int n;
  if (e->cfg.self_obs_v == 1) {
    om_obs_v1(m->nbody, d->qpos, d->qvel, &d->xpos[0][0], &d->xquat[0][0], e->cfg.root_height_obs, obs);
    n = (e->cfg.root_height_obs ? 1 : 0) + 3 * (m->nbody - 1) * 2 + m->nbody * 6 + 6;
  } else {
    om_obs_v2(m->nbody, &d->xpos[0][0], &d->xquat[0][0], &d->linvel[0][0], &d->angvel[0][0], e->cfg.root_height_obs, obs);
    n = (e->cfg.root_height_obs ? 1 : 0) + 3 * (m->nbody - 1) + m->nbody * 12;
  }
  if (e->cfg.task == OM_TASK_SPEED) {
    double hq[4], ex[3] = {1, 0, 0}, r[3];
    heading_quat_inv(d->qpos + 3, hq);
    npt_quat_rotate(hq, ex, r);
    obs[n++] = (float)r[0]; obs[n++] = (float)r[1]; obs[n++] = (float)e->tar;
  } else if (e->cfg.task == OM_TASK_GETUP) {
    obs[n++] = (float)e->tar;
  } else if (e->cfg.task == OM_TASK_REACH) {                 /* compute_location_observations, humanoid_reach.py:21-30 */
    double hq[4], l[3] = {e->tar - d->qpos[0], e->tar_yz[0] - d->qpos[1], e->tar_yz[1] - d->qpos[2]}, r[3];
    heading_quat_inv(d->qpos + 3, hq);
    npt_quat_rotate(hq, l, r);
    obs[n++] = (float)r[0]; obs[n++] = (float)r[1]; obs[n++] = (float)r[2];
  }
}

static void env_substeps(om_env *e, const double *action, int n) {
  double tau[OM_MAXV];
  for (int i = 0; i < n; i++) {
    om_ctrl_torque(e->m, e->d, e->cfg.control_mode, e->cfg.power_scale, action, tau);
    memcpy(e->d->ctrl, tau, sizeof(double) * e->m->nu);
    om_step(e->m, e->d);
  }
}

void om_env_obs(om_env *e, float *obs) { env_obs(e, obs); }

/* test hooks for the np_transform_utils restatements: op 0 quat_mul(a,b), 1 quat_rotate(a, b[0:3]),
 * 2 calc_heading_quat_inv(remove_base_rot(a)) */
void om_quat_op(int op, const double *a, const double *b, double *out) {
  if (op == 0) npt_quat_mul(a, b, out);
  else if (op == 1) npt_quat_rotate(a, b, out);
  else heading_quat_inv(a, out);
}

void om_env_reset(om_env *e, const double *fall_actions, const double *task_rand, float *obs) {
  const om_model *m = e->m; om_data *d = e->d;
  static const double zero2[4] = {0, 0, 0, 0};
  if (e->cfg.task == OM_TASK_GETUP) e->recovery_counter = e->cfg.recovery_steps;
  if (e->cfg.task != OM_TASK_BASE) env_reset_task(e, task_rand ? task_rand : zero2);  /* uses the OLD cur_t */
  if (e->cfg.state_init != OM_INIT_EXTERNAL) { memset(d->qpos, 0, sizeof d->qpos); memset(d->qvel, 0, sizeof d->qvel); }
  if (e->cfg.state_init == OM_INIT_EXTERNAL) {
    /* init_humanoid with a caller-provided state (reference-state init): only the mj_forward below */
  } else if (e->cfg.state_init == OM_INIT_DEFAULT) {
    d->qpos[2] = 0.94; d->qpos[3] = d->qpos[4] = d->qpos[5] = d->qpos[6] = 0.5;
  } else {
    d->qpos[2] = 0.3; d->qpos[3] = 1;
    om_forward(m, d);
    for (int k = 0; k < 3; k++) {
      double act[OM_MAXV];
      for (int i = 0; i < m->nu; i++) act[i] = (fall_actions ? fall_actions[k * m->nu + i] : 0.5) - 0.5;
      env_substeps(e, act, e->cfg.control_freq_inv);
    }
  }
  om_forward(m, d);                                         /* reset_sim */
  e->cur_t = 0;
  env_obs(e, obs);
}

static int legal_contacts(const om_env *e) {
  for (int b = 0; b < e->m->nbody; b++) if (e->d->touch[b] && !e->m->legal[b]) return 0;
  return 1;
}

void om_env_step(om_env *e, const double *action, const double *task_rand, float *obs, double *reward,
                 int *terminated, int *truncated) {
  const om_env_cfg *c = &e->cfg; om_data *d = e->d;
  static const double zero2[4] = {0, 0, 0, 0};
  /* pre_physics_step */
  if (c->task != OM_TASK_BASE && e->cur_t >= e->change_steps) env_reset_task(e, task_rand ? task_rand : zero2);
  if (c->task == OM_TASK_SPEED) v3cpy(e->prev_root_pos, d->xpos[0]);
  env_substeps(e, action, c->control_freq_inv);
  e->cur_t += 1;
  env_obs(e, obs);                                          /* refreshes xpos/xquat via mj_kinematics */
  double rew = 0;
  int term = 0, trunc = e->cur_t > c->episode_length;
  if (c->task == OM_TASK_SPEED) {
    double dtc = c->control_freq_inv * e->m->dt;
    double vx = (d->xpos[0][0] - e->prev_root_pos[0]) / dtc, vy = (d->xpos[0][1] - e->prev_root_pos[1]) / dtc;
    double err = e->tar - vx;
    rew = exp(-0.25 * (err * err + 0.1 * vy * vy));
    term = !legal_contacts(e);
  } else if (c->task == OM_TASK_GETUP) {
    double diff = e->tar - d->xpos[0][2];
    rew = exp(-4.0 * diff * diff);
    if (e->recovery_counter > 0) { e->recovery_counter -= 1; term = 0; trunc = 0; }
    else term = !legal_contacts(e);
  } else if (c->task == OM_TASK_REACH) {                     /* reach_reward, humanoid_reach.py:10-19 */
    const double *p = d->xpos[c->reach_body];
    double dx = e->tar - p[0], dy = e->tar_yz[0] - p[1], dz = e->tar_yz[1] - p[2];
    rew = exp(-4.0 * (dx * dx + dy * dy + dz * dz));
    term = !legal_contacts(e);
  }
  *reward = rew; *terminated = term; *truncated = trunc;
}

/* ------------------------------------------------------------------ getters */
int om_get(const om_model *m, const om_data *d, int f, double *out) {
  int nv = m->nv, nb = m->nbody;
  switch (f) {
    case OM_D_QPOS: memcpy(out, d->qpos, sizeof(double) * m->nq); return m->nq;
    case OM_D_QVEL: memcpy(out, d->qvel, sizeof(double) * nv); return nv;
    case OM_D_QACC: memcpy(out, d->qacc, sizeof(double) * nv); return nv;
    case OM_D_WARM: memcpy(out, d->warm, sizeof(double) * nv); return nv;
    case OM_D_CTRL: memcpy(out, d->ctrl, sizeof(double) * m->nu); return m->nu;
    case OM_D_M: memcpy(out, d->M, sizeof(double) * nv * nv); return nv * nv;
    case OM_D_BIAS: memcpy(out, d->bias, sizeof(double) * nv); return nv;
    case OM_D_XPOS: memcpy(out, d->xpos, sizeof(double) * 3 * nb); return 3 * nb;
    case OM_D_XIPOS: memcpy(out, d->xipos, sizeof(double) * 3 * nb); return 3 * nb;
    case OM_D_XQUAT: memcpy(out, d->xquat, sizeof(double) * 4 * nb); return 4 * nb;
    case OM_D_LINVEL: memcpy(out, d->linvel, sizeof(double) * 3 * nb); return 3 * nb;
    case OM_D_ANGVEL: memcpy(out, d->angvel, sizeof(double) * 3 * nb); return 3 * nb;
    case OM_D_TOUCH: for (int b = 0; b < nb; b++) out[b] = d->touch[b]; return nb;
    case OM_D_NCON: out[0] = d->ncon; return 1;
    case OM_D_CON_POS: memcpy(out, d->con_pos, sizeof(double) * 3 * d->ncon); return 3 * d->ncon;
    case OM_D_CON_FRAME: memcpy(out, d->con_frame, sizeof(double) * 9 * d->ncon); return 9 * d->ncon;
    case OM_D_CON_DIST: memcpy(out, d->con_dist, sizeof(double) * d->ncon); return d->ncon;
    case OM_D_CON_BODY: for (int c = 0; c < d->ncon; c++) out[c] = d->con_body[c]; return d->ncon;
    case OM_D_CON_BODY1: for (int c = 0; c < d->ncon; c++) out[c] = d->con_body1[c]; return d->ncon;
    case OM_D_NSELF: out[0] = d->nself; out[1] = m->npair; out[2] = d->nself_dropped; return 3;
    case OM_D_QACC_SMOOTH: memcpy(out, d->qacc_smooth, sizeof(double) * nv); return nv;
    case OM_D_QFRC_CONSTRAINT: memcpy(out, d->qfrc_constraint, sizeof(double) * nv); return nv;
    case OM_D_NEFC: out[0] = d->nefc; return 1;
    case OM_D_EFC_FORCE: memcpy(out, d->eforce, sizeof(double) * d->nefc); return d->nefc;
    case OM_D_SOLVER_ITER: out[0] = d->solver_iter; out[1] = d->nwarn; return 2;
    case OM_D_LS_STATS: out[0] = (double)d->ls_evals; out[1] = (double)d->ls_calls; out[2] = (double)d->ls_exhausted; return 3;
    case OM_D_QPOS_FWD: memcpy(out, d->qpos_fwd, sizeof(double) * m->nq); return m->nq;
    case OM_D_QVEL_FWD: memcpy(out, d->qvel_fwd, sizeof(double) * nv); return nv;
    case OM_D_ENERGY: {
      double ke = 0, pe = 0;
      for (int i = 0; i < nv; i++) { double s = 0; for (int k = 0; k < nv; k++) s += d->M[i * nv + k] * d->qvel[k]; ke += 0.5 * d->qvel[i] * s; }
      for (int b = 0; b < nb; b++) pe += -m->mass[b] * m->grav * d->xipos[b][2];
      out[0] = ke; out[1] = pe; return 2;
    }
  }
  return -1;
}
int om_set(const om_model *m, om_data *d, int f, const double *in) {
  switch (f) {
    case OM_D_QPOS: memcpy(d->qpos, in, sizeof(double) * m->nq); return 0;
    case OM_D_QVEL: memcpy(d->qvel, in, sizeof(double) * m->nv); return 0;
    case OM_D_WARM: memcpy(d->warm, in, sizeof(double) * m->nv); return 0;
    case OM_D_CTRL: memcpy(d->ctrl, in, sizeof(double) * m->nu); return 0;
    case OM_D_M: memcpy(d->M, in, sizeof(double) * m->nv * m->nv); return 0;
    case OM_D_BIAS: memcpy(d->bias, in, sizeof(double) * m->nv); return 0;
  }
  return -1;
}

/* ------------------------------------------------------------------ threaded rollout (cpu_baseline) */
typedef struct { om_env **envs; int lo, hi, nenv, nsteps; const double *actions; long done; } roll_arg;
static void *roll_worker(void *p) {
  roll_arg *a = (roll_arg *)p;
  float obs[OM_MAXB * 12 + OM_MAXV + 16];
  double rew; int term, trunc;
  for (int s = 0; s < a->nsteps; s++)
    for (int i = a->lo; i < a->hi; i++) {
      int nu = a->envs[i]->m->nu;
      om_env_step(a->envs[i], a->actions + ((size_t)s * a->nenv + i) * nu, NULL, obs, &rew, &term, &trunc);
      if (term || trunc) om_env_reset(a->envs[i], NULL, NULL, obs);
      a->done++;
    }
  return NULL;
}
long om_batch_rollout(om_env **envs, int nenv, int nsteps, const double *actions, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > nenv) nthreads = nenv;
  pthread_t th[256]; roll_arg args[256];
  if (nthreads > 256) nthreads = 256;
  for (int t = 0; t < nthreads; t++) {
    args[t] = (roll_arg){envs, (int)((long)nenv * t / nthreads), (int)((long)nenv * (t + 1) / nthreads), nenv, nsteps, actions, 0};
    pthread_create(&th[t], NULL, roll_worker, &args[t]);
  }
  long tot = 0;
  for (int t = 0; t < nthreads; t++) { pthread_join(th[t], NULL); tot += args[t].done; }
  return tot;
}
