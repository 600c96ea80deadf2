/*
 * orc_camera.c -- oracle restatement of the pinhole camera model used on the
 * front-end path (all of it IS in the reference tree):
 *   project / Jacobian   okvis_cv/include/okvis/cameras/implementation/PinholeCamera.hpp:241-374
 *   backProject          .../PinholeCamera.hpp:574-593
 *   awareness maps       .../PinholeCamera.hpp:180-208
 *   radial-tangential    .../RadialTangentialDistortion.hpp:90-135 (distort), :214-252 (undistort)
 *   equidistant          .../EquidistantDistortion.hpp:87-171 (distort), :319-351 (undistort)
 *   image bounds         .../CameraBase.hpp:97-106
 *   Frame::computeBackProjections  okvis_cv/include/okvis/implementation/Frame.hpp:178-193
 *
 * TEST INFRASTRUCTURE ONLY (see okvfe_oracle.h).  FP64 throughout, evaluated
 * left to right with explicit temporaries (build with -ffp-contract=off).
 * Pinned against the tolerances of okvis_cv/test/TestPinholeCamera.cpp:52-140
 * (back-project/project round trip < 0.01 px, analytic vs numeric Jacobian
 * < 1e-4) in tests/test_oracle_pins.py.
 *
 * atan: the reference calls libm's atan (EquidistantDistortion.hpp:98,138).  A libm atan is
 * not reproducible across libm versions or on the GPU, so this oracle evaluates orc_atan_fixed
 * -- its own copy of the published fdlibm reduction + degree-23 odd polynomial, a fixed
 * sequence of IEEE operations -- which stays within 1 ulp of glibc's atan
 * (tests/test_oracle_pins.py::test_fixed_atan_within_one_ulp_of_libm).  The product carries the
 * same sequence (okvis2_amd/csrc/atan_fixed.h), which is what makes equidistant back-projections
 * comparable as bit patterns.
 *
 * Attribution: orc_cam_distort keeps the reference's operation order and is a close
 * transcription of RadialTangentialDistortion.hpp:111-135 / EquidistantDistortion.hpp:128-171,
 * Copyright (c) 2015 Autonomous Systems Lab / ETH Zurich, (c) 2020 Smart Robotics Lab / Imperial
 * College London, (c) 2024 Smart Robotics Lab / Technical University of Munich, BSD 3-Clause
 * (licence text in the header of those files; its conditions apply to that fragment).
 */
#include "okvfe_oracle.h"

#include <math.h>
#include <string.h>

/* fdlibm-style atan (K.C. Ng's published scheme): break points 7/16, 11/16, 19/16, 39/16 */
/* Transcendental back end of the oracle: 0 (default) = the fixed IEEE sequences the device evaluates
 * (bit-exact against the GPU), 1 = libm's atan / acos, i.e. what the REFERENCE calls
 * (EquidistantDistortion.hpp:98,138; Frontend.cpp:1312).  tests/test_oracle_libm_variant.py measures
 * what the <= 1 ulp between the two changes downstream (back-projections, gates, match rows). */
static int g_orc_use_libm = 0;
void orc_set_libm(int on) { g_orc_use_libm = on != 0; }
int orc_get_libm(void) { return g_orc_use_libm; }
double orc_atan_eval(double x) { return g_orc_use_libm ? atan(x) : orc_atan_fixed(x); }

double orc_atan_fixed(double x) {
  static const double hi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01,
                               9.82793723247329054082e-01, 1.57079632679489655800e+00};
  static const double lo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17,
                               1.39033110312309984516e-17, 6.12323399573676603587e-17};
  static const double aT[11] = {
      3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01,
      -1.11111104054623557880e-01, 9.09088713343650656196e-02, -7.69187620504482999495e-02,
      6.66107313738753120669e-02, -5.83357013379057348645e-02, 4.97687799461593236017e-02,
      -3.65315727442169155270e-02, 1.62858201153657823623e-02};
  double ax, t, z, w, s1, s2, r;
  int id, neg;
  if (x != x) return x;
  neg = x < 0.0;
  ax = neg ? -x : x;
  if (ax >= 7.378697629483821e19) { /* 2^66 */
    r = hi[3] + lo[3];
    return neg ? -r : r;
  }
  if (ax < 0.4375) {
    if (ax < 1.862645149230957e-09) return x; /* 2^-29 */
    id = -1;
    t = ax;
  } else if (ax < 1.1875) {
    if (ax < 0.6875) {
      id = 0;
      t = (2.0 * ax - 1.0) / (2.0 + ax);
    } else {
      id = 1;
      t = (ax - 1.0) / (ax + 1.0);
    }
  } else if (ax < 2.4375) {
    id = 2;
    t = (ax - 1.5) / (1.0 + 1.5 * ax);
  } else {
    id = 3;
    t = -1.0 / ax;
  }
  z = t * t;
  w = z * z;
  s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  if (id < 0)
    r = t - t * (s1 + s2);
  else
    r = hi[id] - ((t * (s1 + s2) - lo[id]) - t);
  return neg ? -r : r;
}

/* distort u -> out, with the 2x2 Jacobian J (row-major) when J != NULL */
int orc_cam_distort(const orc_camera* c, const double u[2], double out[2], double J[4]) {
  const double u0 = u[0], u1 = u[1];
  if (c->dist_type == ORC_DIST_NONE) {
    out[0] = u0;
    out[1] = u1;
    if (J) {
      J[0] = 1.0; J[1] = 0.0; J[2] = 0.0; J[3] = 1.0;
    }
    return 1;
  }
  if (c->dist_type == ORC_DIST_RADTAN) {
    const double k1 = c->d[0], k2 = c->d[1], p1 = c->d[2], p2 = c->d[3];
    const double mx_u = u0 * u0;
    const double my_u = u1 * u1;
    const double mxy_u = u0 * u1;
    const double rho_u = mx_u + my_u;
    const double rad_dist_u = k1 * rho_u + k2 * rho_u * rho_u;
    out[0] = u0 + u0 * rad_dist_u + 2.0 * p1 * mxy_u + p2 * (rho_u + 2.0 * mx_u);
    out[1] = u1 + u1 * rad_dist_u + 2.0 * p2 * mxy_u + p1 * (rho_u + 2.0 * my_u);
    if (J) {
      J[0] = 1 + rad_dist_u + k1 * 2.0 * mx_u + k2 * rho_u * 4 * mx_u + 2.0 * p1 * u1 +
             6 * p2 * u0;
      J[2] = k1 * 2.0 * u0 * u1 + k2 * 4 * rho_u * u0 * u1 + p1 * 2.0 * u0 + 2.0 * p2 * u1;
      J[1] = J[2];
      J[3] = 1 + rad_dist_u + k1 * 2.0 * my_u + k2 * rho_u * 4 * my_u + 6 * p1 * u1 +
             2.0 * p2 * u0;
    }
    return 1;
  }
  /* equidistant */
  {
    const double k1 = c->d[0], k2 = c->d[1], k3 = c->d[2], k4 = c->d[3];
    const double r = sqrt(u0 * u0 + u1 * u1);
    const double theta = orc_atan_eval(r);
    const double theta2 = theta * theta;
    const double theta4 = theta2 * theta2;
    const double theta6 = theta4 * theta2;
    const double theta8 = theta4 * theta4;
    const double thetad = theta * (1.0 + k1 * theta2 + k2 * theta4 + k3 * theta6 + k4 * theta8);
    const double scaling = (r > 1e-8) ? thetad / r : 1.0;
    out[0] = scaling * u0;
    out[1] = scaling * u1;
    if (J) {
      if (r > 1e-8) {
        double t2, t3, t4, t6, t7, t8, t9, t11, t17, t18, t19, t20, t25;
        t2 = u0 * u0;
        t3 = u1 * u1;
        t4 = t2 + t3;
        t6 = orc_atan_eval(sqrt(t4));
        t7 = t6 * t6;
        t8 = 1.0 / sqrt(t4);
        t9 = t7 * t7;
        t11 = 1.0 / ((t2 + t3) + 1.0);
        t17 = (((k1 * t7 + k2 * t9) + k3 * t7 * t9) + k4 * (t9 * t9)) + 1.0;
        t18 = 1.0 / t4;
        t19 = 1.0 / sqrt(t4 * t4 * t4);
        t20 = t6 * t8 * t17;
        t25 = ((k2 * t6 * t7 * t8 * t11 * u1 * 4.0 + k3 * t6 * t8 * t9 * t11 * u1 * 6.0) +
               k4 * t6 * t7 * t8 * t9 * t11 * u1 * 8.0) +
              k1 * t6 * t8 * t11 * u1 * 2.0;
        t4 = ((k2 * t6 * t7 * t8 * t11 * u0 * 4.0 + k3 * t6 * t8 * t9 * t11 * u0 * 6.0) +
              k4 * t6 * t7 * t8 * t9 * t11 * u0 * 8.0) +
             k1 * t6 * t8 * t11 * u0 * 2.0;
        t7 = t11 * t17 * t18 * u0 * u1;
        J[1] = (t7 + t6 * t8 * t25 * u0) - t6 * t17 * t19 * u0 * u1;
        J[3] = ((t20 - t3 * t6 * t17 * t19) + t3 * t11 * t17 * t18) + t6 * t8 * t25 * u1;
        J[0] = ((t20 - t2 * t6 * t17 * t19) + t2 * t11 * t17 * t18) + t6 * t8 * t4 * u0;
        J[2] = (t7 + t6 * t8 * t4 * u1) - t6 * t17 * t19 * u0 * u1;
      } else {
        J[0] = 1.0; J[1] = 0.0; J[2] = 0.0; J[3] = 1.0;
      }
    }
    return 1;
  }
}

/* Gauss-Newton undistortion: at most 5 (radtan) / 20 (equidistant) iterations,
 * success once chi2 < 1e-6, early exit at chi2 < 1e-15. */
int orc_cam_undistort(const orc_camera* c, const double pd[2], double out[2]) {
  if (c->dist_type == ORC_DIST_NONE) {
    out[0] = pd[0];
    out[1] = pd[1];
    return 1;
  }
  const int n = (c->dist_type == ORC_DIST_RADTAN) ? 5 : 20;
  double x_bar[2] = {pd[0], pd[1]};
  int success = 0;
  for (int i = 0; i < n; ++i) {
    double x_tmp[2], E[4];
    orc_cam_distort(c, x_bar, x_tmp, E);
    const double e0 = pd[0] - x_tmp[0], e1 = pd[1] - x_tmp[1];
    /* E2 = E^T E */
    const double a = E[0] * E[0] + E[2] * E[2];
    const double b = E[0] * E[1] + E[2] * E[3];
    const double cc = E[1] * E[0] + E[3] * E[2];
    const double d = E[1] * E[1] + E[3] * E[3];
    const double det = a * d - b * cc;
    const double invdet = 1.0 / det;
    const double i00 = d * invdet, i01 = -b * invdet, i10 = -cc * invdet, i11 = a * invdet;
    /* B = inv(E2) * E^T */
    const double b00 = i00 * E[0] + i01 * E[1];
    const double b01 = i00 * E[2] + i01 * E[3];
    const double b10 = i10 * E[0] + i11 * E[1];
    const double b11 = i10 * E[2] + i11 * E[3];
    const double du0 = b00 * e0 + b01 * e1;
    const double du1 = b10 * e0 + b11 * e1;
    x_bar[0] += du0;
    x_bar[1] += du1;
    const double chi2 = e0 * e0 + e1 * e1;
    if (chi2 < 1e-6) success = 1;
    if (chi2 < 1e-15) {
      success = 1;
      break;
    }
  }
  out[0] = x_bar[0];
  out[1] = x_bar[1];
  return success;
}

int orc_cam_backproject(const orc_camera* c, const double pt[2], double dir[3]) {
  const double one_over_fu = 1.0 / c->fu, one_over_fv = 1.0 / c->fv;
  double p2[2], und[2];
  p2[0] = (pt[0] - c->cu) * one_over_fu;
  p2[1] = (pt[1] - c->cv) * one_over_fv;
  const int success = orc_cam_undistort(c, p2, und);
  dir[0] = und[0];
  dir[1] = und[1];
  dir[2] = 1.0;
  return success;
}

int orc_cam_project(const orc_camera* c, const double p[3], double pt[2], double J23[6]) {
  if (fabs(p[2]) < 1.0e-12) return 4;
  const double rz = 1.0 / p[2];
  const double rz2 = rz * rz;
  double und[2] = {p[0] * rz, p[1] * rz};
  double dist[2], D[4];
  orc_cam_distort(c, und, dist, J23 ? D : NULL);
  if (J23) {
    J23[0] = c->fu * D[0] * rz;
    J23[1] = c->fu * D[1] * rz;
    J23[2] = -c->fu * (p[0] * D[0] + p[1] * D[1]) * rz2;
    J23[3] = c->fv * D[2] * rz;
    J23[4] = c->fv * D[3] * rz;
    J23[5] = -c->fv * (p[0] * D[2] + p[1] * D[3]) * rz2;
  }
  pt[0] = c->fu * dist[0] + c->cu;
  pt[1] = c->fv * dist[1] + c->cv;
  if (pt[0] < 0.0 || pt[1] < 0.0) return 1;
  if (pt[0] >= (double)c->w || pt[1] >= (double)c->h) return 1;
  if (p[2] > 0.0) return 0;
  return 3;
}

/* rays: h*w*3 f32 (normalised back-projection, zero when it failed);
 * jac: h*w*6 f32 (2x3 point Jacobian of project(ray), row-major) written only
 * when project() is Successful -- the reference leaves the other entries
 * uninitialised (cv::Mat is not zero-filled); the oracle defines them as 0. */
void orc_cam_awareness_maps(const orc_camera* c, float* rays, float* jac) {
  for (int v = 0; v < c->h; ++v) {
    for (int u = 0; u < c->w; ++u) {
      double ray[3];
      const double pt[2] = {(double)u, (double)v};
      if (orc_cam_backproject(c, pt, ray)) {
        const double n = sqrt(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
        ray[0] /= n; ray[1] /= n; ray[2] /= n;
      } else {
        ray[0] = ray[1] = ray[2] = 0.0;
      }
      float* r = rays + ((size_t)v * c->w + u) * 3;
      r[0] = (float)ray[0]; r[1] = (float)ray[1]; r[2] = (float)ray[2];
      float* j = jac + ((size_t)v * c->w + u) * 6;
      double p2[2], J[6];
      if (orc_cam_project(c, ray, p2, J) == 0) {
        for (int i = 0; i < 6; ++i) j[i] = (float)J[i];
      } else {
        for (int i = 0; i < 6; ++i) j[i] = 0.0f;
      }
    }
  }
}

int orc_backproject_keypoints(const orc_camera* c, const orc_keypoint* kps, int n, double* dirs,
                              uint8_t* valid) {
  int ctr = 0;
  for (int k = 0; k < n; ++k) {
    const double pt[2] = {(double)kps[k].x, (double)kps[k].y};
    const int ok = orc_cam_backproject(c, pt, dirs + 3 * (size_t)k);
    valid[k] = (uint8_t)ok;
    ctr += ok;
  }
  return ctr;
}

/* Field-of-view overlap of camera `c` as seen by camera `o` (NCameraSystem::computeOverlaps,
 * okvis_cv/src/NCameraSystem.cpp:48-119): every pixel of c is back-projected, rotated by
 * R = C(T_Cother_C) (points at infinity), projected into o, and counted when the projection is
 * Successful and back-projects onto the same direction (|cos - 1| < 1e-10).  mask (h*w of c) may
 * be NULL.  Returns hasOverlap. */
int orc_cam_overlap(const orc_camera* c, const orc_camera* o, const double R[9], uint8_t* mask) {
  int has = 0;
  for (int u = 0; u < c->w; ++u) {
    for (int v = 0; v < c->h; ++v) {
      double ray[3], ro[3], pt[2], ver[3];
      const double p[2] = {(double)u, (double)v};
      orc_cam_backproject(c, p, ray);
      for (int i = 0; i < 3; ++i) ro[i] = R[3 * i] * ray[0] + R[3 * i + 1] * ray[1] + R[3 * i + 2] * ray[2];
      int hit = 0;
      if (orc_cam_project(o, ro, pt, NULL) == 0) {
        orc_cam_backproject(o, pt, ver);
        const double na = sqrt(ro[0] * ro[0] + ro[1] * ro[1] + ro[2] * ro[2]);
        const double nb = sqrt(ver[0] * ver[0] + ver[1] * ver[1] + ver[2] * ver[2]);
        const double dot = (ro[0] / na) * (ver[0] / nb) + (ro[1] / na) * (ver[1] / nb) +
                           (ro[2] / na) * (ver[2] / nb);
        if (fabs(dot - 1.0) < 1.0e-10) hit = 1;
      }
      if (mask) mask[(size_t)v * c->w + u] = (uint8_t)hit;
      has |= hit;
    }
  }
  return has;
}
