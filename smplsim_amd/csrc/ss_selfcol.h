// ss_selfcol.h — pair functions of the body-body contacts: capsule-capsule, capsule-box, box-box.
//
// What mj_step's collision stage does for the geom types of the reference's humanoids
// (reference smpl_sim/data/assets/mjcf/smpl_humanoid.xml:5,24 — boxes contype 7, capsules contype 1, conaffinity 1: every
// pair collides; :231-242 excludes), restated as rules [MJ-doc] (the float64 twin and its known-answer tests:
// oracle/oracle.c, tests/test_oracle_selfcollision.py):
//   capsule-capsule  closest points of the two segments, then sphere-sphere (two tests when the axes are parallel)
//   capsule-box      the segment point closest to the box — exact minimiser of the convex piecewise-quadratic squared
//                    distance, found from the zero of its piecewise-linear slope — a sphere-box test there and one at the
//                    far end of the segment (at most 2 contacts)
//   box-box          separating-axis test over the 15 axes; face axis: the face of the other box that faces the selected
//                    (reference) face is clipped against the reference face's side planes (Sutherland-Hodgman, the face
//                    manifold of ODE's dBoxBox that mjc_BoxBox descends from) and the clipped polygon's vertices within the
//                    margin of the reference face are the contacts (at most 8); edge-edge axis: one contact at the closest
//                    points of the supporting edges
// One lane runs one pair; plain scalar code in the kernel's scalar type.  Normals point from the pair's first geom to its second.
#pragma once
#include "ss_hdr.h"

#ifndef SS_DEV
#if defined(__HIPCC__)
#define SS_DEV __device__ __forceinline__
#else
#define SS_DEV inline
#endif
#endif

namespace ss {
namespace sc {

struct NCon { real pos[3], n[3], dist; };
constexpr real kMin = real(1e-15);

SS_DEV real dot3(const real *a, const real *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
SS_DEV real clampr(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }

// fb: direction used when the centres (nearly) coincide — closer than 1e-5 of the radii, where the separation vector is rounding
// noise (far above float32 rounding of the positions, so the float64 twin takes the same branch); null = (1,0,0)
SS_DEV int sphere_sphere(const real *p1, real r1, const real *p2, real r2, real margin, const real *fb, NCon *o) {
  const real d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  const real len = SS_M(sqrt)(dot3(d, d)), dist = len - (r1 + r2);
  if (dist > margin) return 0;
  if (len < real(1e-5) * (r1 + r2)) { o->n[0] = fb ? fb[0] : real(1); o->n[1] = fb ? fb[1] : real(0); o->n[2] = fb ? fb[2] : real(0); }
  else { const real il = real(1) / len; o->n[0] = d[0] * il; o->n[1] = d[1] * il; o->n[2] = d[2] * il; }
  for (int k = 0; k < 3; k++) o->pos[k] = p1[k] + o->n[k] * (r1 + real(0.5) * dist);
  o->dist = dist;
  return 1;
}

// capsule = centre p, unit axis a, radius r, half length h
SS_DEV int capsule_capsule(const real *p1, const real *a1, real r1, real h1, const real *p2, const real *a2, real r2, real h2,
                           real margin, NCon *o) {
  const real dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  const real ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2), u = -dot3(a1, dif), v = dot3(a2, dif);
  const real det = ma * mc - mb * mb;
  real c1[3], c2[3];
  if (SS_M(fabs)(det) >= kMin) {
    real x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
    if (x1 > h1) { x1 = h1; x2 = (v - mb * h1) / mc; }
    else if (x1 < -h1) { x1 = -h1; x2 = (v + mb * h1) / mc; }
    if (x2 > h2) { x2 = h2; x1 = clampr((u - mb * h2) / ma, -h1, h1); }
    else if (x2 < -h2) { x2 = -h2; x1 = clampr((u + mb * h2) / ma, -h1, h1); }
    for (int k = 0; k < 3; k++) { c1[k] = p1[k] + a1[k] * x1; c2[k] = p2[k] + a2[k] * x2; }
    // axes that (nearly) intersect: push apart along their common perpendicular, on the side of geom 2's centre
    real fb[3] = {a1[1] * a2[2] - a1[2] * a2[1], a1[2] * a2[0] - a1[0] * a2[2], a1[0] * a2[1] - a1[1] * a2[0]};
    const real fl = SS_M(sqrt)(dot3(fb, fb));
    if (fl < kMin) { fb[0] = 1; fb[1] = 0; fb[2] = 0; } else { const real il = real(1) / fl; fb[0] *= il; fb[1] *= il; fb[2] *= il; }
    if (dot3(fb, dif) > 0) { fb[0] = -fb[0]; fb[1] = -fb[1]; fb[2] = -fb[2]; }
    return sphere_sphere(c1, r1, c2, r2, margin, fb, o);
  }
  int n = 0;                                                 // parallel axes: segment ends against the other segment
  for (int e = 0; e < 4 && n < 2; e++) {
    real x1, x2;
    if (e < 2) { x1 = e == 0 ? h1 : -h1; x2 = (v - mb * x1) / mc; if (x2 > h2 || x2 < -h2) continue; }
    else { x2 = e == 2 ? h2 : -h2; x1 = (u - mb * x2) / ma; if (x1 > h1 || x1 < -h1) continue; }
    for (int k = 0; k < 3; k++) { c1[k] = p1[k] + a1[k] * x1; c2[k] = p2[k] + a2[k] * x2; }
    n += sphere_sphere(c1, r1, c2, r2, margin, nullptr, o + n);
  }
  return n;
}

// sphere (first geom) against box (second geom: centre bp, rotation bm row-major with the box axes as columns, half sizes bs)
// hint (box frame, may be null): which of two opposite faces a centre lying on the box's mid-plane is pushed out through
// (capsule_box passes the capsule's centre: its mid-range rule puts the sphere exactly there when the axis skewers a thin box)
SS_DEV int sphere_box(const real *c, real r, const real *bp, const real *bm, const real *bs, real margin, const real *hint, NCon *o) {
  const real d[3] = {c[0] - bp[0], c[1] - bp[1], c[2] - bp[2]};
  real l[3], cl[3];
  for (int i = 0; i < 3; i++) { l[i] = bm[i] * d[0] + bm[3 + i] * d[1] + bm[6 + i] * d[2]; cl[i] = clampr(l[i], -bs[i], bs[i]); }
  const real v[3] = {l[0] - cl[0], l[1] - cl[1], l[2] - cl[2]};
  const real len = SS_M(sqrt)(dot3(v, v));
  real nl[3], dist;
  if (len >= kMin) {
    dist = len - r;
    if (dist > margin) return 0;
    const real il = real(1) / len;
    nl[0] = v[0] * il; nl[1] = v[1] * il; nl[2] = v[2] * il;
  } else {                                                   // centre inside the box: out through the nearest face
    int k = 0; real best = bs[0] - SS_M(fabs)(l[0]);
    for (int i = 1; i < 3; i++) { const real e = bs[i] - SS_M(fabs)(l[i]); if (e < best) { best = e; k = i; } }
    real sg = l[k] >= 0 ? real(1) : real(-1);
    if (hint && SS_M(fabs)(l[k]) < real(1e-5) * bs[k]) sg = hint[k] >= 0 ? real(1) : real(-1);
    nl[0] = nl[1] = nl[2] = 0;
    if (k == 0) { nl[0] = sg; cl[0] = sg * bs[0]; } else if (k == 1) { nl[1] = sg; cl[1] = sg * bs[1]; } else { nl[2] = sg; cl[2] = sg * bs[2]; }
    dist = -best - r;
  }
  for (int i = 0; i < 3; i++) {
    const real nw = bm[3 * i] * nl[0] + bm[3 * i + 1] * nl[1] + bm[3 * i + 2] * nl[2];
    const real pw = bp[i] + bm[3 * i] * cl[0] + bm[3 * i + 1] * cl[1] + bm[3 * i + 2] * cl[2];
    o->n[i] = -nw; o->pos[i] = pw + nw * real(0.5) * dist;
  }
  o->dist = dist;
  return 1;
}

// slope / 2 of the squared distance between the box [-s, s]^3 and the point p + t a (box frame)
SS_DEV real seg_box_slope(const real *p, const real *a, const real *s, real t) {
  real g = 0;
  for (int i = 0; i < 3; i++) {
    const real x = p[i] + t * a[i], e = SS_M(fabs)(x) - s[i];
    if (e > 0) g += a[i] * (x > 0 ? e : -e);
  }
  return g;
}

SS_DEV int capsule_box(const real *cp, const real *ca, real r, real h, const real *bp, const real *bm, const real *bs, real margin,
                       NCon *o) {
  const real d[3] = {cp[0] - bp[0], cp[1] - bp[1], cp[2] - bp[2]};
  real p[3], a[3];
  for (int i = 0; i < 3; i++) {
    p[i] = bm[i] * d[0] + bm[3 + i] * d[1] + bm[6 + i] * d[2];
    a[i] = bm[i] * ca[0] + bm[3 + i] * ca[1] + bm[6 + i] * ca[2];
  }
  real T[8], G[8]; int nt = 0;
  T[nt++] = -h;
  for (int i = 0; i < 3; i++) {
    if (SS_M(fabs)(a[i]) < kMin) continue;
    for (int sg = -1; sg <= 1; sg += 2) { const real t = ((real)sg * bs[i] - p[i]) / a[i]; if (t > -h && t < h) T[nt++] = t; }
  }
  T[nt++] = h;
  for (int i = 1; i < nt; i++) { const real x = T[i]; int j = i; while (j > 0 && T[j - 1] > x) { T[j] = T[j - 1]; j--; } T[j] = x; }
  for (int k = 0; k < nt; k++) G[k] = seg_box_slope(p, a, bs, T[k]);
  // zero of the non-decreasing piecewise-linear slope; |g| <= tol counts as zero (middle of the zero range), tol far above the
  // float32 rounding of g: same branch as the float64 twin when the exact slope at a breakpoint is zero
  const real bmax = bs[0] > bs[1] ? (bs[0] > bs[2] ? bs[0] : bs[2]) : (bs[1] > bs[2] ? bs[1] : bs[2]);
  const real tol = real(1e-5) * (h + bmax);
  real ts;
  if (G[0] > tol) ts = T[0];
  else if (G[nt - 1] < -tol) ts = T[nt - 1];
  else {
    int i = 0;
    while (G[i] < -tol) i++;
    if (G[i] > tol) ts = T[i - 1] - G[i - 1] * (T[i] - T[i - 1]) / (G[i] - G[i - 1]);
    else { int e = i; while (e + 1 < nt && G[e + 1] <= tol) e++; ts = real(0.5) * (T[i] + T[e]); }
  }
  int n = 0;
  real c[3];
  for (int i = 0; i < 3; i++) c[i] = cp[i] + ts * ca[i];
  n += sphere_box(c, r, bp, bm, bs, margin, p, o + n);
  const real t2 = ts >= 0 ? -h : h;
  if (SS_M(fabs)(t2 - ts) > real(1e-6) * (h > kMin ? h : real(1))) {
    for (int i = 0; i < 3; i++) c[i] = cp[i] + t2 * ca[i];
    n += sphere_box(c, r, bp, bm, bs, margin, p, o + n);
  }
  return n;
}

SS_DEV int box_box(const real *pa, const real *ma, const real *sa, const real *pb, const real *mb, const real *sb, real margin, NCon *o) {
  real A[3][3], B[3][3], R[3][3], AR[3][3];
  for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) { A[i][k] = ma[3 * k + i]; B[i][k] = mb[3 * k + i]; }
  const real t[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { R[i][j] = dot3(A[i], B[j]); AR[i][j] = SS_M(fabs)(R[i][j]); }
  real best = real(-1e30), bn[3] = {0, 0, 0};
  int bcode = 0;                                             // reference face: axis i of A (0..2) or axis j of B (3..5)
  for (int i = 0; i < 3; i++) {
    const real tl = dot3(t, A[i]);
    const real sep = SS_M(fabs)(tl) - (sa[i] + sb[0] * AR[i][0] + sb[1] * AR[i][1] + sb[2] * AR[i][2]);
    if (sep > best) { best = sep; bcode = i; const real sg = tl >= 0 ? real(1) : real(-1); for (int k = 0; k < 3; k++) bn[k] = sg * A[i][k]; }
  }
  for (int j = 0; j < 3; j++) {
    const real tl = dot3(t, B[j]);
    const real sep = SS_M(fabs)(tl) - (sa[0] * AR[0][j] + sa[1] * AR[1][j] + sa[2] * AR[2][j] + sb[j]);
    if (sep > best) { best = sep; bcode = 3 + j; const real sg = tl >= 0 ? real(1) : real(-1); for (int k = 0; k < 3; k++) bn[k] = sg * B[j][k]; }
  }
  real ebest = real(-1e30), en[3] = {0, 0, 0}; int ecode = -1;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    real L[3] = {A[i][1] * B[j][2] - A[i][2] * B[j][1], A[i][2] * B[j][0] - A[i][0] * B[j][2], A[i][0] * B[j][1] - A[i][1] * B[j][0]};
    const real ln = SS_M(sqrt)(dot3(L, L));
    if (ln < real(1e-6)) continue;
    const real il = real(1) / ln;
    L[0] *= il; L[1] *= il; L[2] *= il;
    const real tl = dot3(t, L);
    real ra = 0, rb = 0;
    for (int k = 0; k < 3; k++) { ra += sa[k] * SS_M(fabs)(dot3(A[k], L)); rb += sb[k] * SS_M(fabs)(dot3(B[k], L)); }
    const real sep = SS_M(fabs)(tl) - (ra + rb);
    if (sep > ebest) { ebest = sep; ecode = 3 * i + j; const real sg = tl >= 0 ? real(1) : real(-1); for (int k = 0; k < 3; k++) en[k] = sg * L[k]; }
  }
  if (best > margin || ebest > margin) return 0;
  if (ecode >= 0 && ebest > best + real(1e-4) + real(0.05) * SS_M(fabs)(best)) {
    const int i = ecode / 3, j = ecode - 3 * i;
    real qa[3] = {pa[0], pa[1], pa[2]}, qb[3] = {pb[0], pb[1], pb[2]};
    for (int k = 0; k < 3; k++) {
      if (k != i) { const real s_ = (dot3(en, A[k]) > 0 ? real(1) : real(-1)) * sa[k]; for (int c = 0; c < 3; c++) qa[c] += A[k][c] * s_; }
      if (k != j) { const real s_ = (dot3(en, B[k]) > 0 ? real(-1) : real(1)) * sb[k]; for (int c = 0; c < 3; c++) qb[c] += B[k][c] * s_; }
    }
    const real w[3] = {qa[0] - qb[0], qa[1] - qb[1], qa[2] - qb[2]};
    const real b_ = R[i][j], d_ = dot3(A[i], w), e_ = dot3(B[j], w), den = real(1) - b_ * b_;
    real x = den > real(1e-12) ? (b_ * e_ - d_) / den : real(0), y = den > real(1e-12) ? (e_ - b_ * d_) / den : real(0);
    x = clampr(x, -sa[i], sa[i]); y = clampr(y, -sb[j], sb[j]);
    for (int k = 0; k < 3; k++) { o[0].pos[k] = real(0.5) * ((qa[k] + x * A[i][k]) + (qb[k] + y * B[j][k])); o[0].n[k] = en[k]; }
    o[0].dist = ebest;
    return 1;
  }
  // ---- face contact.  Reference face = the face of the box that owns the best axis, on the side of the other box; incident
  // face = the face of the other box most anti-parallel to it.  The incident rectangle is clipped against the four side planes of
  // the reference face; every vertex of the clipped polygon within the margin of the reference plane is a contact, placed half
  // way between the vertex and the plane, normal = the axis (first geom -> second geom).  A vertex within clip_tol of a side plane
  // counts as inside (clip_tol far above float32 rounding: the float64 twin classifies alike), coincident output points are merged.
  const bool refA = bcode < 3;
  const int ir = refA ? bcode : bcode - 3, i1 = (ir + 1) % 3, i2 = (ir + 2) % 3;
  const real(*Rf)[3] = refA ? A : B;
  const real(*Xi)[3] = refA ? B : A;
  const real *pr = refA ? pa : pb, *sr = refA ? sa : sb, *pi_ = refA ? pb : pa, *si = refA ? sb : sa;
  const real nout[3] = {refA ? bn[0] : -bn[0], refA ? bn[1] : -bn[1], refA ? bn[2] : -bn[2]};   // outward normal of the reference face
  int j = 0; real bd = real(-1);
  for (int k = 0; k < 3; k++) { const real d_ = SS_M(fabs)(dot3(nout, Xi[k])); if (d_ > bd) { bd = d_; j = k; } }
  const real sgn = dot3(nout, Xi[j]) > 0 ? real(-1) : real(1);
  const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  real poly[8][3], tmp[8][3];
  int np = 4;
  for (int v = 0; v < 4; v++) {
    const real su = (v == 0 || v == 3) ? si[j1] : -si[j1], sv = v < 2 ? si[j2] : -si[j2];
    for (int k = 0; k < 3; k++) poly[v][k] = pi_[k] + sgn * si[j] * Xi[j][k] + su * Xi[j1][k] + sv * Xi[j2][k];
  }
  const real clip_tol = real(1e-5) * (sr[i1] + sr[i2]);
  for (int e = 0; e < 4 && np > 0; e++) {
    const int tt = e < 2 ? i1 : i2;
    const real sg = (e & 1) ? real(-1) : real(1);
    real fprev;
    { const real *x = poly[np - 1]; fprev = sg * ((x[0] - pr[0]) * Rf[tt][0] + (x[1] - pr[1]) * Rf[tt][1] + (x[2] - pr[2]) * Rf[tt][2]) - sr[tt]; }
    int nq = 0;
    for (int v = 0; v < np; v++) {
      const real *x = poly[v], *xp = poly[v == 0 ? np - 1 : v - 1];
      const real f = sg * ((x[0] - pr[0]) * Rf[tt][0] + (x[1] - pr[1]) * Rf[tt][1] + (x[2] - pr[2]) * Rf[tt][2]) - sr[tt];
      const bool in = f <= clip_tol, inp = fprev <= clip_tol;
      if (in != inp && nq < 8) {                              // the edge crosses the plane
        const real u = fprev / (fprev - f);
        for (int k = 0; k < 3; k++) tmp[nq][k] = xp[k] + u * (x[k] - xp[k]);
        nq++;
      }
      if (in && nq < 8) { for (int k = 0; k < 3; k++) tmp[nq][k] = x[k]; nq++; }
      fprev = f;
    }
    np = nq;
    for (int v = 0; v < np; v++) for (int k = 0; k < 3; k++) poly[v][k] = tmp[v][k];
  }
  int n = 0;
  const real merge2 = clip_tol * clip_tol;
  for (int v = 0; v < np && n < 8; v++) {
    const real *x = poly[v];
    const real dist = (x[0] - pr[0]) * nout[0] + (x[1] - pr[1]) * nout[1] + (x[2] - pr[2]) * nout[2] - sr[ir];
    if (dist > margin) continue;
    bool dup = false;
    for (int q_ = 0; q_ < n; q_++) {                          // tmp is free again: it keeps the vertices already emitted
      const real ex = tmp[q_][0] - x[0], ey = tmp[q_][1] - x[1], ez = tmp[q_][2] - x[2];
      const real en_ = ex * nout[0] + ey * nout[1] + ez * nout[2];
      const real tx = ex - en_ * nout[0], ty = ey - en_ * nout[1], tz = ez - en_ * nout[2];
      dup |= tx * tx + ty * ty + tz * tz <= merge2;
    }
    if (dup) continue;
    for (int k = 0; k < 3; k++) { tmp[n][k] = x[k]; o[n].pos[k] = x[k] - real(0.5) * dist * nout[k]; o[n].n[k] = bn[k]; }
    o[n].dist = dist;
    n++;
  }
  return n;
}

}  // namespace sc
}  // namespace ss
