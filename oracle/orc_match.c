/*
 * orc_match.c -- oracle restatement of the brute-force Hamming matchers and
 * their FP64 geometric gates (all of this IS in the reference tree):
 *   brisk::Hamming::PopcntofXORed(a, b, 3)   call sites Frontend.cpp:341,1580,1661,1846,2024
 *   triangulation::triangulateFast            okvis_frontend/src/stereo_triangulation.cpp:50-132
 *   Frontend::matchStereo inner loops         okvis_frontend/src/Frontend.cpp:2016-2076
 *   Frontend::matchMotionStereo inner loops   okvis_frontend/src/Frontend.cpp:1812-1905
 *   Frontend::verifyRecognisedPlace matching  okvis_frontend/src/Frontend.cpp:330-355
 *
 * TEST INFRASTRUCTURE ONLY (see okvfe_oracle.h).  Vector arithmetic is written
 * out component-wise, sums left to right, no FMA (build -ffp-contract=off).
 * A pose is (C, r) with p_W = C p_C + r; its inverse is (C^T, -(C^T r)) as in
 * okvis_kinematics/include/okvis/kinematics/implementation/Transformation.hpp:207-209.
 */
#include "okvfe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

uint32_t orc_popcnt_xor(const uint8_t* a, const uint8_t* b, int n128) {
  uint32_t c = 0;
  for (int i = 0; i < 16 * n128; ++i) {
    uint8_t x = (uint8_t)(a[i] ^ b[i]);
    while (x) {
      c += x & 1u;
      x >>= 1;
    }
  }
  return c;
}

static inline uint32_t popc48(const uint8_t* a, const uint8_t* b) {
  uint64_t x[6], y[6];
  memcpy(x, a, 48);
  memcpy(y, b, 48);
  uint32_t c = 0;
  for (int i = 0; i < 6; ++i) c += (uint32_t)__builtin_popcountll(x[i] ^ y[i]);
  return c;
}

/* Order of every 3-term FP64 sum of the gate chain (dot products, norms, 3x3 * 3 products).  The reference
 * evaluates them through Eigen (stereo_triangulation.cpp:62-76, Frontend.cpp:2027-2073): a fixed-size
 * Vector3d reduction is not packet-aligned, so a stock build takes Eigen's unrolled non-vectorised
 * redux (Redux.h, redux_novec_unroller: split at Length / 2), which evaluates x0 + (x1 + x2).
 * tree = 1 (default): that order; tree = 0: left to right, (x0 + x1) + x2 -- what rounds 1-4 assumed.
 * Neither can be confirmed here (no Eigen in the image); tools/ref_compare picks the winner in one run
 * on a machine that has the reference built. */
static int g_reduction_tree = 1;
void orc_set_reduction(int tree) { g_reduction_tree = tree != 0; }
int orc_get_reduction(void) { return g_reduction_tree; }
static inline double sum3(double p0, double p1, double p2) {
  if (g_reduction_tree) {
    const double t = p1 + p2;
    return p0 + t;
  }
  const double s = p0 + p1;
  return s + p2;
}
static inline double dot3(const double a[3], const double b[3]) {
  const double p0 = a[0] * b[0];
  const double p1 = a[1] * b[1];
  const double p2 = a[2] * b[2];
  return sum3(p0, p1, p2);
}
static inline void normalize3(const double v[3], double out[3]) {
  const double n = sqrt(dot3(v, v));
  out[0] = v[0] / n;
  out[1] = v[1] / n;
  out[2] = v[2] / n;
}
static inline void rot(const double C[9], const double v[3], double out[3]) {
  out[0] = dot3(C, v);
  out[1] = dot3(C + 3, v);
  out[2] = dot3(C + 6, v);
}
static inline void rot_t(const double C[9], const double v[3], double out[3]) {
  for (int i = 0; i < 3; ++i) {
    const double p0 = C[i] * v[0];
    const double p1 = C[3 + i] * v[1];
    const double p2 = C[6 + i] * v[2];
    out[i] = sum3(p0, p1, p2);
  }
}
/* hp_C = T^-1 * hp_W  (Transformation::operator*(Vector4d), Transformation.hpp:271-278) */
static inline void inv_transform_h(const orc_pose* T, const double hp[4], double out[4]) {
  double ri[3], cr[3], h[3];
  rot_t(T->C, T->r, cr);
  ri[0] = -cr[0]; ri[1] = -cr[1]; ri[2] = -cr[2];
  rot_t(T->C, hp, h);
  const double s = hp[3];
  out[0] = h[0] + ri[0] * s;
  out[1] = h[1] + ri[1] * s;
  out[2] = h[2] + ri[2] * s;
  out[3] = s;
}

static void midpoint_parallel(const double p1[3], const double e1[3], const double p2[3],
                              const double e2[3], const double t12[3], double sigma, double hp[4],
                              int* is_valid) {
  *is_valid = 1;
  double m[3], mid[3], d[3], dn[3];
  const double tn = sqrt(dot3(t12, t12));
  const double f = 40.0 * (0.01 > tn ? 0.01 : tn);
  for (int i = 0; i < 3; ++i) {
    m[i] = p1[i] + 0.5 * t12[i];
    mid[i] = m[i] + f * (e1[i] + e2[i]);
  }
  hp[0] = mid[0]; hp[1] = mid[1]; hp[2] = mid[2]; hp[3] = 1.0;
  const double c26 = cos(2.6 * sigma);
  for (int i = 0; i < 3; ++i) d[i] = mid[i] - p1[i];
  normalize3(d, dn);
  if (dot3(e1, dn) < c26) *is_valid = 0;
  for (int i = 0; i < 3; ++i) d[i] = mid[i] - p2[i];
  normalize3(d, dn);
  if (dot3(e2, dn) < c26) *is_valid = 0;
}

void orc_triangulate_fast(const double p1[3], const double e1[3], const double p2[3],
                          const double e2[3], double sigma, double hp[4], int* is_valid,
                          int* is_parallel) {
  *is_parallel = 0;
  *is_valid = 1;
  double t12[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  const double b0 = dot3(t12, e1), b1 = dot3(t12, e2);
  const double a00 = dot3(e1, e1);
  const double a10 = dot3(e1, e2);
  const double a01 = -a10;
  const double a11 = -dot3(e2, e2);
  /* computeInverseWithCheck(A_inverse, invertible, 1e-12) for a 2x2 */
  const double det = a00 * a11 - a01 * a10;
  const int invertible = fabs(det) > 1.0e-12;
  if (!invertible) {
    *is_parallel = 1;
    midpoint_parallel(p1, e1, p2, e2, t12, sigma, hp, is_valid);
    return;
  }
  const double invdet = 1.0 / det;
  const double i00 = a11 * invdet, i10 = -a10 * invdet, i01 = -a01 * invdet, i11 = a00 * invdet;
  const double l0 = i00 * b0 + i01 * b1;
  const double l1 = i10 * b0 + i11 * b1;
  if (l0 < 0.01 || l1 < 0.01) {
    *is_parallel = 1;
    midpoint_parallel(p1, e1, p2, e2, t12, sigma, hp, is_valid);
    return;
  }
  double mid[3], d1[3], d2[3], n1[3], n2[3];
  for (int i = 0; i < 3; ++i) {
    const double xm = l0 * e1[i] + p1[i];
    const double xn = l1 * e2[i] + p2[i];
    mid[i] = (xm + xn) / 2.0;
    d1[i] = mid[i] - p1[i];
    d2[i] = mid[i] - p2[i];
  }
  normalize3(d1, n1);
  normalize3(d2, n2);
  const double c26 = cos(2.6 * sigma);
  if (dot3(e1, n1) < c26) *is_valid = 0;
  if (dot3(e2, n2) < c26) *is_valid = 0;
  if (dot3(n2, n1) > cos(6.0 * sigma)) *is_parallel = 1;
  hp[0] = mid[0]; hp[1] = mid[1]; hp[2] = mid[2]; hp[3] = 1.0;
}

/* ---- matchStereo (Frontend.cpp:2016-2076): k0 ascending, k1 ascending, strict < ------------ */
void orc_match_stereo(const uint8_t* desc0, const orc_keypoint* kp0, const double* bp0,
                      const uint8_t* bpv0, int n0, const uint8_t* desc1, const orc_keypoint* kp1,
                      const double* bp1, const uint8_t* bpv1, int n1, const orc_pose* T_WC0,
                      const orc_pose* T_WC1, double f0, double f1, double threshold,
                      orc_stereo_match* out) {
  for (int k0 = 0; k0 < n0; ++k0) {
    double distances = threshold;
    int initialisable = 0;
    double hps_W[4] = {0, 0, 0, 0};
    int k1_match = 0;
    for (int k1 = 0; k1 < n1; ++k1) {
      const uint32_t dist = popc48(desc0 + 48 * (size_t)k0, desc1 + 48 * (size_t)k1);
      if ((double)dist < distances) {
        const double size0 = (double)kp0[k0].size, size1 = (double)kp1[k1].size;
        const double s0 = size0 / f0, s1 = size1 / f1;
        const double sigma = (s0 > s1 ? s0 : s1) * 0.125; /* std::max(a,b): a<b ? b : a */
        int is_valid = 0, is_parallel = 0;
        if (!bpv0[k0]) continue;
        if (!bpv1[k1]) continue;
        double v[3], e0_W[3], e1_W[3], hp_W[4], hp_C0[4], hp_C1[4];
        rot(T_WC0->C, bp0 + 3 * (size_t)k0, v);
        normalize3(v, e0_W);
        rot(T_WC1->C, bp1 + 3 * (size_t)k1, v);
        normalize3(v, e1_W);
        orc_triangulate_fast(T_WC0->r, e0_W, T_WC1->r, e1_W, sigma, hp_W, &is_valid,
                             &is_parallel);
        inv_transform_h(T_WC0, hp_W, hp_C0);
        inv_transform_h(T_WC1, hp_W, hp_C1);
        if (!is_parallel) {
          const double w4 = hp_W[3];
          hp_W[0] /= w4; hp_W[1] /= w4; hp_W[2] /= w4; hp_W[3] /= w4;
          if (hp_C0[2] / hp_C0[3] < 0.05) is_valid = 0;
          if (hp_C1[2] / hp_C1[3] < 0.05) is_valid = 0;
          if (dot3(e0_W, e1_W) < 0.8) is_valid = 0;
        }
        if (is_valid) {
          distances = (double)dist;
          memcpy(hps_W, hp_W, sizeof(hps_W));
          k1_match = k1;
          initialisable = !is_parallel;
        }
      }
    }
    orc_stereo_match* o = &out[k0];
    memset(o, 0, sizeof(*o));
    if (distances < threshold) {
      o->k1 = k1_match;
      o->dist = (int32_t)distances;
      o->initialisable = initialisable;
      memcpy(o->hp_W, hps_W, sizeof(hps_W));
    } else {
      o->k1 = -1;
      o->dist = (int32_t)threshold;
    }
  }
}

/* ---- matchMotionStereo (Frontend.cpp:1789-1905) --------------------------------------------
 * Frame 0 = older frame, frame 1 = current frame, same camera `cam`.
 * skip0[k0] != 0 stands for the estimator-state tests at :1814-1841 (landmark
 * already initialised / keypoint already observed): such k0 are not matched.
 * matched1[k1] != 0 = current keypoint already carries a landmark (:1795-1798)
 * and is left out of the packed candidate set (the packed order is k1
 * ascending, so iterating k1 ascending and skipping is the same loop). */
void orc_match_motion_stereo(const uint8_t* desc0, const orc_keypoint* kp0, const double* bp0,
                             const uint8_t* bpv0, const uint8_t* skip0, int n0,
                             const uint8_t* desc1, const orc_keypoint* kp1, const double* bp1,
                             const uint8_t* bpv1, const uint8_t* matched1, int n1,
                             const orc_pose* T_WC0, const orc_pose* T_WC1, const orc_camera* cam,
                             uint32_t threshold, orc_motion_match* out) {
  const double f0 = 0.5 * (cam->fu + cam->fv);
  for (int k0 = 0; k0 < n0; ++k0) {
    orc_motion_match* o = &out[k0];
    memset(o, 0, sizeof(*o));
    o->k1 = -1;
    o->dist = (int32_t)threshold;
    if (skip0 && skip0[k0]) continue;
    uint32_t distances = threshold;
    int initialisable = 0;
    double quality = 0.0;
    double hps_W[4] = {0, 0, 0, 0};
    int k1_max = 1000;
    if (!bpv0[k0]) continue;
    double v[3], e0_W[3];
    rot(T_WC0->C, bp0 + 3 * (size_t)k0, v);
    normalize3(v, e0_W);
    const double sigma = (double)kp0[k0].size / f0 * 0.125;
    for (int k1 = 0; k1 < n1; ++k1) {
      if (matched1 && matched1[k1]) continue;
      const uint32_t dist = popc48(desc0 + 48 * (size_t)k0, desc1 + 48 * (size_t)k1);
      if (dist < distances) {
        int is_valid = 0, is_parallel = 0;
        if (!bpv1[k1]) continue;
        double e1_W[3], hp_W[4], hp_C0[4], hp_C1[4];
        rot(T_WC1->C, bp1 + 3 * (size_t)k1, v);
        normalize3(v, e1_W);
        const double ee = dot3(e0_W, e1_W);
        if (ee < 0.5) continue;
        orc_triangulate_fast(T_WC0->r, e0_W, T_WC1->r, e1_W, sigma, hp_W, &is_valid,
                             &is_parallel);
        if (!is_valid) continue;
        inv_transform_h(T_WC0, hp_W, hp_C0);
        inv_transform_h(T_WC1, hp_W, hp_C1);
        if (ee < 0.8) is_valid = 0;
        if (!is_parallel) {
          const double w4 = hp_W[3];
          hp_W[0] /= w4; hp_W[1] /= w4; hp_W[2] /= w4; hp_W[3] /= w4;
          if (hp_C0[2] / hp_C0[3] < 0.2) is_valid = 0;
          if (hp_C1[2] / hp_C1[3] < 0.2) is_valid = 0;
        }
        if (is_valid) {
          k1_max = k1;
          distances = dist;
          double a[3], b[3], an[3], bn[3];
          for (int i = 0; i < 3; ++i) {
            a[i] = hp_W[i] - T_WC0->r[i];
            b[i] = hp_W[i] - T_WC1->r[i];
          }
          normalize3(a, an);
          normalize3(b, bn);
          quality = acos(dot3(an, bn));
          memcpy(hps_W, hp_W, sizeof(hps_W));
          initialisable = !is_parallel;
        }
      }
    }
    if (distances < threshold) {
      o->k1 = k1_max;
      o->dist = (int32_t)distances;
      o->initialisable = initialisable;
      o->quality = quality;
      memcpy(o->hp_W, hps_W, sizeof(hps_W));
      /* 4 px reprojection check of the winner (:1897-1905) */
      double hp_C1[4], head[3], pt1p[2];
      inv_transform_h(T_WC1, hps_W, hp_C1);
      if (hp_C1[3] < 0) {
        head[0] = -hp_C1[0]; head[1] = -hp_C1[1]; head[2] = -hp_C1[2];
      } else {
        head[0] = hp_C1[0]; head[1] = hp_C1[1]; head[2] = hp_C1[2];
      }
      const int status = orc_cam_project(cam, head, pt1p, NULL);
      const double ex = (double)kp1[k1_max].x - pt1p[0], ey = (double)kp1[k1_max].y - pt1p[1];
      o->accepted = (status == 0 && sqrt(ex * ex + ey * ey) < 4.0) ? 1 : 0;
    }
  }
}

/* ---- candidate list and ungated arg-min ----------------------------------------------------- */
int orc_hamming_candidates(const uint8_t* A, int nA, const uint8_t* B, int nB, int thr,
                           orc_cand* out, int cap) {
  int n = 0;
  for (int i = 0; i < nA; ++i)
    for (int j = 0; j < nB; ++j) {
      const int d = (int)popc48(A + 48 * (size_t)i, B + 48 * (size_t)j);
      if (d < thr) {
        if (n < cap) {
          out[n].i = i;
          out[n].j = j;
          out[n].dist = d;
        }
        ++n;
      }
    }
  return n;
}

/* For every row i of A: first-lowest j with minimal distance below thr
 * (Frontend.cpp:337-346: uint32 distMin = threshold; strict <). */
void orc_hamming_argmin(const uint8_t* A, int nA, const uint8_t* B, int nB, uint32_t thr,
                        int32_t* best_j, uint32_t* best_d) {
  for (int i = 0; i < nA; ++i) {
    uint32_t dmin = thr;
    int32_t jmin = -1;
    for (int j = 0; j < nB; ++j) {
      const uint32_t d = popc48(A + 48 * (size_t)i, B + 48 * (size_t)j);
      if (d < dmin) {
        dmin = d;
        jmin = j;
      }
    }
    best_j[i] = jmin;
    best_d[i] = dmin;
  }
}

/* ---- matchToMapByThread, 3-D landmarks (Frontend.cpp:1552-1589) ----------------------------------
 * Landmarks in the given order (the reference iterates a std::map by ascending LandmarkId);
 * landmark l owns pool rows desc_begin[l] .. desc_begin[l+1]-1.  distances[k] starts at the
 * threshold (double, fresh vector at Frontend.cpp:1365) and is updated with strict '<'. */
void orc_match_to_map(const uint8_t* desc, const orc_keypoint* kps, const uint8_t* use, int n_k,
                      const double* proj, const int32_t* desc_begin, int n_lm, const uint8_t* pool,
                      double reprojection_threshold, double threshold, int32_t* best_lm,
                      int32_t* best_d) {
  const double thr_sq = reprojection_threshold * reprojection_threshold;
  for (int k = 0; k < n_k; ++k) {
    best_lm[k] = -1;
    best_d[k] = (int32_t)threshold;
  }
  for (int l = 0; l < n_lm; ++l) {
    for (int k = 0; k < n_k; ++k) {
      if (!use[k]) continue;
      const double dx = proj[2 * l] - (double)kps[k].x, dy = proj[2 * l + 1] - (double)kps[k].y;
      if (dx * dx + dy * dy > thr_sq) continue;
      for (int d = desc_begin[l]; d < desc_begin[l + 1]; ++d) {
        const double dist = (double)popc48(desc + 48 * (size_t)k, pool + 48 * (size_t)d);
        if (dist < (double)best_d[k]) {
          best_d[k] = (int32_t)dist;
          best_lm[k] = l;
        }
      }
    }
  }
}

/* ---- matchToMapByThreadUnitialised (Frontend.cpp:1616-1719) -------------------------------------
 * Landmarks that are not yet 3-D: every pooled descriptor d carries the observing ray e0_W[d]
 * and camera centre r0_W[d].  Per keypoint k (use[k] != 0, Frontend.cpp:1621-1635):
 * e1_W = C (e1_C / |e1_C|); dist < distances[k] -> epipolar-plane and divergence tests unless the
 * rays are nearly parallel -> triangulateFast(sigma = 1/f) valid -> not closer than 0.2 m to
 * either centre -> if the landmark is the one the keypoint already carries: count and leave this
 * landmark's descriptor loop; else update distances[k], landmark, and hp (only when not parallel).
 * previous[k] = index of the landmark keypoint k already carries, or -1. */
void orc_match_to_map_uninit(const uint8_t* desc, const double* bp, const uint8_t* use,
                             const int32_t* previous, int n_k, const int32_t* desc_begin, int n_lm,
                             const uint8_t* pool, const double* e0_W, const double* r0_W,
                             const orc_pose* T_WC1, double focal, double threshold,
                             int32_t* best_lm, int32_t* best_d, double* hps_W, uint8_t* hp_set,
                             int32_t* ctr_out) {
  const double sigma = 1.0 / focal;
  const double cos6 = cos(6.0 * sigma);
  int ctr = 0;
  for (int k = 0; k < n_k; ++k) {
    best_lm[k] = -1;
    best_d[k] = (int32_t)threshold;
    hp_set[k] = 0;
    hps_W[4 * k] = hps_W[4 * k + 1] = hps_W[4 * k + 2] = hps_W[4 * k + 3] = 0.0;
  }
  for (int l = 0; l < n_lm; ++l) {
    for (int k = 0; k < n_k; ++k) {
      if (!use[k]) continue;
      double en[3], e1_W[3];
      normalize3(bp + 3 * (size_t)k, en);
      rot(T_WC1->C, en, e1_W);
      for (int d = desc_begin[l]; d < desc_begin[l + 1]; ++d) {
        const double dist = (double)popc48(desc + 48 * (size_t)k, pool + 48 * (size_t)d);
        if (dist < (double)best_d[k]) {
          const double* e0 = e0_W + 3 * (size_t)d;
          const double* r0 = r0_W + 3 * (size_t)d;
          if (dot3(e0, e1_W) < cos6) {
            double t[3], et[3], c0[3], c1[3], n0[3], n1[3], cx[3], nn[3], nnn[3];
            for (int i = 0; i < 3; ++i) t[i] = T_WC1->r[i] - r0[i];
            normalize3(t, et);
            c0[0] = e0[1] * et[2] - e0[2] * et[1];
            c0[1] = e0[2] * et[0] - e0[0] * et[2];
            c0[2] = e0[0] * et[1] - e0[1] * et[0];
            normalize3(c0, n0);
            c1[0] = e1_W[1] * et[2] - e1_W[2] * et[1];
            c1[1] = e1_W[2] * et[0] - e1_W[0] * et[2];
            c1[2] = e1_W[0] * et[1] - e1_W[1] * et[0];
            normalize3(c1, n1);
            if (dot3(n0, n1) < cos6) continue; /* not in epipolar plane */
            cx[0] = e0[1] * e1_W[2] - e0[2] * e1_W[1];
            cx[1] = e0[2] * e1_W[0] - e0[0] * e1_W[2];
            cx[2] = e0[0] * e1_W[1] - e0[1] * e1_W[0];
            for (int i = 0; i < 3; ++i) nn[i] = n0[i] + n0[i];
            normalize3(nn, nnn);
            if (dot3(cx, nnn) > 0.0) continue; /* divergent rays */
          }
          double hp[4];
          int is_valid = 0, is_parallel = 0;
          orc_triangulate_fast(r0, e0, T_WC1->r, e1_W, sigma, hp, &is_valid, &is_parallel);
          if (!is_valid) continue;
          if (!is_parallel) {
            double p[3], a[3], b[3];
            for (int i = 0; i < 3; ++i) {
              p[i] = hp[i] / hp[3];
              a[i] = p[i] - r0[i];
              b[i] = p[i] - T_WC1->r[i];
            }
            if (sqrt(dot3(a, a)) < 0.2) is_valid = 0;
            if (sqrt(dot3(b, b)) < 0.2) is_valid = 0;
          }
          if (!is_valid) continue;
          if (l == previous[k]) {
            ++ctr;
            break;
          }
          best_d[k] = (int32_t)dist;
          best_lm[k] = l;
          if (!is_parallel) {
            memcpy(hps_W + 4 * (size_t)k, hp, 4 * sizeof(double));
            hp_set[k] = 1;
          }
        }
      }
    }
  }
  *ctr_out = ctr;
}

/* ---- verifyRecognisedPlace, all landmarks of one camera (Frontend.cpp:330-355) ------------------
 * Landmark l owns pool rows desc_begin[l] .. desc_begin[l+1]-1 (its descriptors in insertion
 * order, :318-326).  Running minimum with strict <, descriptors outer, k inner; k_min stays 0 and
 * dist_min = threshold when nothing is below the threshold (the caller then ignores k_min). */
void orc_verify_place(const uint8_t* pool, const int32_t* desc_begin, int n_landmarks,
                      const uint8_t* frame_desc, int K, uint32_t threshold, int32_t* k_min,
                      uint32_t* dist_min) {
  for (int l = 0; l < n_landmarks; ++l) {
    uint32_t dmin = threshold;
    int32_t kmin = 0;
    for (int d = desc_begin[l]; d < desc_begin[l + 1]; ++d)
      for (int k = 0; k < K; ++k) {
        const uint32_t dist = popc48(frame_desc + 48 * (size_t)k, pool + 48 * (size_t)d);
        if (dist < dmin) {
          dmin = dist;
          kmin = k;
        }
      }
    k_min[l] = kmin;
    dist_min[l] = dmin;
  }
}

/* ---- DBoW2 vocabulary descent with the FBrisk trait --------------------------------------------
 * DBoW2 (external/DBoW2, un-vendored) TemplatedVocabulary<FBrisk::TDescriptor, FBrisk>::transform
 * as published: from the root, at every level take the child with the smallest
 * FBrisk::distance (= Hamming over 48 bytes, okvis_frontend/src/FBrisk.cpp:64-67), the FIRST child
 * winning ties (strict <, children in stored order), until a leaf; the leaf's word id is the
 * feature's word (features quantised at Frontend.cpp:756-766 via dBow_->database).
 * Node n's children are child_index[child_begin[n] .. child_begin[n+1]).  word[n] < 0 for inner
 * nodes. */
void orc_voc_transform(const uint8_t* desc, int n, const uint8_t* node_desc, const int32_t* child_begin,
                       const int32_t* child_index, const int32_t* word, int32_t* word_out,
                       int32_t* node_out) {
  for (int i = 0; i < n; ++i) {
    int node = 0;
    while (child_begin[node + 1] > child_begin[node]) {
      int best = child_index[child_begin[node]];
      uint32_t best_d = popc48(desc + 48 * (size_t)i, node_desc + 48 * (size_t)best);
      for (int c = child_begin[node] + 1; c < child_begin[node + 1]; ++c) {
        const int id = child_index[c];
        const uint32_t d = popc48(desc + 48 * (size_t)i, node_desc + 48 * (size_t)id);
        if (d < best_d) {
          best_d = d;
          best = id;
        }
      }
      node = best;
    }
    word_out[i] = word[node];
    node_out[i] = node;
  }
}

/* ---- matchToMap: landmark projection + descriptor-view pooling (Frontend.cpp:1219-1359) ---------
 * Per landmark (caller's order = ascending LandmarkId): FoV check by projecting the homogeneous
 * point into the current camera (:1232-1256), then the observations in the reference's iteration
 * order (observations.rbegin() .. rend(), i.e. the caller flattens them that way): 3-D test
 * (:1284-1290), view-point (> 0.6 rad) and scale (> 50 %) pruning unless `exclusive`
 * (loopClosureLandmarksToUseExclusively, :1293-1303), score, and the three-slot "keep the best"
 * buffer EXACTLY as written at :1305-1340 -- including its quirks: the descriptor is written at row
 * `o` (the largest slot index replaced so far), not at the replaced slot, and only the first `o`
 * rows survive the crop at :1344, so a landmark with a single accepted observation ends with zero
 * rows and is skipped (:1351-1354) and at most 2 rows are ever kept.
 * acos: orc_acos_fixed (fixed IEEE sequence, <= 1 ulp from libm; the score only ranks views).
 * Outputs per landmark: status 0 = not matched against (outside the FoV / no rows), 1 = 3-D,
 * 2 = not 3-D yet; n_desc = rows kept; obs[r] = observation whose descriptor sits in row r;
 * projection; e_W / r_W (3 doubles per kept row: observing ray and camera centre, :1326-1330). */
double orc_acos_fixed(double x) {
  static const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
                      pi = 3.14159265358979311600e+00;
  static const double pS[6] = {1.66666666666666657415e-01, -3.25565818622400915405e-01,
                               2.01212532134862925881e-01, -4.00555345006794114027e-02,
                               7.91534994289814532176e-04, 3.47933107596021167570e-05};
  static const double qS[5] = {1.0, -2.40339491173441421878e+00, 2.02094576023350569471e+00,
                               -6.88283971605453293030e-01, 7.70381505559019352791e-02};
  double ax, z, p, q, r, sq, w, df, c;
  if (x != x) return x;
  ax = x < 0.0 ? -x : x;
  if (ax >= 1.0) {
    if (x == 1.0) return 0.0;
    if (x == -1.0) return pi + 2.0 * pio2_lo;
    return (x - x) / (x - x);
  }
  if (ax < 0.5) {
    if (ax < 6.938893903907228e-18) return pio2_hi + pio2_lo;
    z = x * x;
  } else if (x < 0.0) {
    z = (1.0 + x) * 0.5;
  } else {
    z = (1.0 - x) * 0.5;
  }
  p = z * (pS[0] + z * (pS[1] + z * (pS[2] + z * (pS[3] + z * (pS[4] + z * pS[5])))));
  q = 1.0 + z * (qS[1] + z * (qS[2] + z * (qS[3] + z * qS[4])));
  r = p / q;
  if (ax < 0.5) return pio2_hi - (x - (pio2_lo - x * r));
  sq = sqrt(z);
  if (x < 0.0) {
    w = r * sq - pio2_lo;
    return pi - 2.0 * (sq + w);
  }
  {
    uint64_t bits;
    memcpy(&bits, &sq, 8);
    bits &= 0xFFFFFFFF00000000ull;
    memcpy(&df, &bits, 8);
  }
  c = (z - df * df) / (sq + df);
  w = r * sq + c;
  return 2.0 * (df + w);
}

void orc_prepare_landmarks(const double* hp_W, const double* quality, const int32_t* obs_begin,
                           int n_landmarks, const int32_t* obs_pose, const double* obs_bp,
                           const orc_pose* poses_old, const orc_pose* T_WC1, const orc_camera* cam,
                           double repr_threshold, int exclusive, int32_t* status, int32_t* n_desc,
                           int32_t* obs_rows /* n*3 */, double* projection /* n*2 */,
                           double* e_W /* n*2*3 */, double* r_W /* n*2*3 */) {
  const double maxU = (double)cam->w + repr_threshold, maxV = (double)cam->h + repr_threshold;
  const double focal = cam->fu + cam->fv; /* sum, as at Frontend.cpp:1213-1215 */
  const double cos10 = cos(10.0 / focal), cos06 = cos(0.6);
  for (int l = 0; l < n_landmarks; ++l) {
    status[l] = 0;
    n_desc[l] = 0;
    obs_rows[3 * l] = obs_rows[3 * l + 1] = obs_rows[3 * l + 2] = -1;
    projection[2 * l] = projection[2 * l + 1] = 0.0;
    for (int i = 0; i < 6; ++i) e_W[6 * l + i] = r_W[6 * l + i] = 0.0;
    const double* hp = hp_W + 4 * (size_t)l;
    const double p_W[3] = {hp[0] / hp[3], hp[1] / hp[3], hp[2] / hp[3]};
    const double r_Wv[3] = {p_W[0] - T_WC1->r[0], p_W[1] - T_WC1->r[1], p_W[2] - T_WC1->r[2]};
    double e_Wv[3];
    normalize3(r_Wv, e_Wv);
    const double rn = sqrt(dot3(r_Wv, r_Wv));
    const double r = 0.01 > rn ? 0.01 : rn;
    double hp_C[4], head[3], kp[2];
    inv_transform_h(T_WC1, hp, hp_C);
    if (hp_C[3] < 0) {
      head[0] = -hp_C[0]; head[1] = -hp_C[1]; head[2] = -hp_C[2];
    } else {
      head[0] = hp_C[0]; head[1] = hp_C[1]; head[2] = hp_C[2];
    }
    const int st = orc_cam_project(cam, head, kp, NULL);
    if (st == 4 || st == 3) continue; /* Invalid, Behind */
    if (kp[0] < -repr_threshold || kp[1] < -repr_threshold || kp[0] > maxU || kp[1] > maxV) continue;
    projection[2 * l] = kp[0];
    projection[2 * l + 1] = kp[1];
    int is3d = 0, o = 0, rows[3] = {-1, -1, -1};
    double best[3] = {1.0, 1.0, 1.0}, ew[3][3], rw[3][3];
    for (int ob = obs_begin[l]; ob < obs_begin[l + 1]; ++ob) {
      const orc_pose* To = poses_old + obs_pose[ob];
      const double r_old[3] = {p_W[0] - To->r[0], p_W[1] - To->r[1], p_W[2] - To->r[2]};
      if (!is3d) {
        const double f = 0.2 / focal / quality[l];
        const double rc[3] = {r_Wv[0] - f * r_old[0], r_Wv[1] - f * r_old[1], r_Wv[2] - f * r_old[2]};
        double a[3], b[3];
        normalize3(r_Wv, a);
        normalize3(rc, b);
        if (dot3(a, b) > cos10) is3d = 1;
      }
      double eo[3];
      normalize3(r_old, eo);
      const double cosVC = dot3(e_Wv, eo);
      if (cosVC < cos06 && !exclusive) continue;
      const double scaleChange = fabs(r - sqrt(dot3(r_old, r_old))) / r;
      if (scaleChange > 0.5 && !exclusive) continue;
      const double score = 0.5 * ((orc_get_libm() ? acos(cosVC) : orc_acos_fixed(cosVC)) / 0.6 + scaleChange / 0.5);
      double worst = 0.0;
      int wi = 0;
      for (int n = 0; n < 3; ++n)
        if (best[n] > worst) {
          worst = best[n];
          wi = n;
        }
      if (score < best[wi]) {
        rows[o] = ob;
        double en[3];
        normalize3(obs_bp + 3 * (size_t)ob, en);
        rot(To->C, en, ew[o]);
        rw[o][0] = To->r[0]; rw[o][1] = To->r[1]; rw[o][2] = To->r[2];
        o = o > wi ? o : wi;
        best[wi] = score;
      }
    }
    if (o == 0) continue; /* "no observations -- weird" */
    status[l] = is3d ? 1 : 2;
    n_desc[l] = o;
    for (int k = 0; k < 3; ++k) obs_rows[3 * l + k] = rows[k];
    for (int k = 0; k < o && k < 2; ++k)
      for (int i = 0; i < 3; ++i) {
        e_W[6 * l + 3 * k + i] = ew[k][i];
        r_W[6 * l + 3 * k + i] = rw[k][i];
      }
  }
}


/* ---- DBoW2 bag-of-words vector and L1 database query (behind dBow_->database.add / query,
 * Frontend.cpp:756-766).  DBoW2 is an un-vendored submodule of the reference (external/DBoW2,
 * .gitmodules); this follows its published TemplatedVocabulary::transform(features, BowVector) and
 * TemplatedDatabase::queryL1: addWeight / addIfNotExist per feature in order, division by the
 * number of distinct words when the scoring does not normalise, L1 normalisation in ascending word
 * order; the query walks the query words in ascending order and, per word, the inverted-file row in
 * ascending entry order: value[entry] += |q - d| - |q| - |d|; score = -value / 2.  Entries without
 * a common word are not listed (-1 here). */
int orc_bow_vector(const int32_t* word_ids, int n_features, const double* word_weight, int n_words, int weighting,
                   int normalise_l1, int32_t* ids_out, double* values_out) {
  double* acc = (double*)calloc((size_t)n_words, sizeof(double));
  unsigned char* seen = (unsigned char*)calloc((size_t)n_words, 1);
  const int sums = weighting == 0 || weighting == 1;
  for (int i = 0; i < n_features; ++i) {
    const int id = word_ids[i];
    const double w = word_weight[id];
    if (!(w > 0)) continue;
    if (!seen[id]) { seen[id] = 1; acc[id] = w; }
    else if (sums) acc[id] = acc[id] + w;
  }
  int n = 0;
  for (int id = 0; id < n_words; ++id)
    if (seen[id]) { ids_out[n] = id; values_out[n] = acc[id]; ++n; }
  if (normalise_l1) {
    double norm = 0.0;
    for (int i = 0; i < n; ++i) norm = norm + fabs(values_out[i]);
    if (norm > 0.0) for (int i = 0; i < n; ++i) values_out[i] = values_out[i] / norm;
  } else if (sums && n > 0) {
    const double nd = (double)n;
    for (int i = 0; i < n; ++i) values_out[i] = values_out[i] / nd;
  }
  free(acc);
  free(seen);
  return n;
}

void orc_bow_query_l1(const int32_t* db_begin, const int32_t* db_ids, const double* db_values, int n_entries,
                      const int32_t* q_ids, const double* q_values, int n_q, int n_words, double* scores) {
  /* the inverted file: row of word w = (entry, value) pairs in ascending entry order */
  int32_t* row_begin = (int32_t*)calloc((size_t)n_words + 1, sizeof(int32_t));
  const int m = db_begin[n_entries];
  for (int i = 0; i < m; ++i) row_begin[db_ids[i] + 1]++;
  for (int w = 0; w < n_words; ++w) row_begin[w + 1] += row_begin[w];
  int32_t* fill = (int32_t*)malloc((size_t)n_words * sizeof(int32_t));
  memcpy(fill, row_begin, (size_t)n_words * sizeof(int32_t));
  int32_t* row_entry = (int32_t*)malloc((size_t)(m > 0 ? m : 1) * sizeof(int32_t));
  double* row_value = (double*)malloc((size_t)(m > 0 ? m : 1) * sizeof(double));
  for (int e = 0; e < n_entries; ++e)
    for (int i = db_begin[e]; i < db_begin[e + 1]; ++i) {
      const int pos = fill[db_ids[i]]++;
      row_entry[pos] = e;
      row_value[pos] = db_values[i];
    }
  double* value = (double*)calloc((size_t)(n_entries > 0 ? n_entries : 1), sizeof(double));
  unsigned char* hit = (unsigned char*)calloc((size_t)(n_entries > 0 ? n_entries : 1), 1);
  for (int j = 0; j < n_q; ++j) {
    const double q = q_values[j];
    for (int r = row_begin[q_ids[j]]; r < row_begin[q_ids[j] + 1]; ++r) {
      const double d = row_value[r];
      double t = fabs(q - d);
      t = t - fabs(q);
      t = t - fabs(d);
      value[row_entry[r]] = value[row_entry[r]] + t;
      hit[row_entry[r]] = 1;
    }
  }
  for (int e = 0; e < n_entries; ++e) scores[e] = hit[e] ? -value[e] / 2.0 : -1.0;
  free(row_begin); free(fill); free(row_entry); free(row_value); free(value); free(hit);
}
