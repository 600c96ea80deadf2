// camera_dev.h -- device-side pinhole camera model (FP64), shared by k_describe.hip and k_match.hip.
// Restates okvis::cameras::PinholeCamera<D>::backProject / project and the distortion models
// (okvis_cv/include/okvis/cameras/implementation/PinholeCamera.hpp:241-283,574-593,
//  RadialTangentialDistortion.hpp:90-135,214-252, EquidistantDistortion.hpp:87-171,319-351).
// Component-wise, sums left to right, no FMA (-ffp-contract=off).
//
// Attribution: the distortion formulas in distort() below (radial-tangential value + Jacobian, and
// the machine-generated equidistant Jacobian with its temporaries t2..t25) keep the reference's
// operation order on purpose -- FP64 bit-exactness depends on it -- and are therefore a close
// transcription of RadialTangentialDistortion.hpp:111-135 and EquidistantDistortion.hpp:128-171,
// which are Copyright (c) 2015 Autonomous Systems Lab / ETH Zurich, (c) 2020 Smart Robotics Lab /
// Imperial College London, (c) 2024 Smart Robotics Lab / Technical University of Munich,
// distributed under the BSD 3-Clause licence (see the licence header of those files; the
// conditions -- retain the copyright notice, the list of conditions and the disclaimer; no
// endorsement with the holders' names -- apply to this fragment).
#pragma once

#include "atan_fixed.h"
#include "equidistant_jacobian.h"
#include "okvfe_internal.h"

namespace okvfe {
namespace cam {

__device__ inline void distort(const DeviceCamera& c, double u0, double u1, double out[2], double J[4]) {
  if (c.distortion == OKVFE_DIST_NONE) {
    out[0] = u0; out[1] = u1;
    J[0] = 1.0; J[1] = 0.0; J[2] = 0.0; J[3] = 1.0;
    return;
  }
  if (c.distortion == OKVFE_DIST_RADTAN) {
    const double k1 = c.d[0], k2 = c.d[1], p1 = c.d[2], p2 = c.d[3];
    const double mx_u = u0 * u0;
    const double my_u = u1 * u1;
    const double mxy_u = u0 * u1;
    const double rho_u = mx_u + my_u;
    const double rad_dist_u = k1 * rho_u + k2 * rho_u * rho_u;
    out[0] = u0 + u0 * rad_dist_u + 2.0 * p1 * mxy_u + p2 * (rho_u + 2.0 * mx_u);
    out[1] = u1 + u1 * rad_dist_u + 2.0 * p2 * mxy_u + p1 * (rho_u + 2.0 * my_u);
    J[0] = 1 + rad_dist_u + k1 * 2.0 * mx_u + k2 * rho_u * 4 * mx_u + 2.0 * p1 * u1 + 6 * p2 * u0;
    J[2] = k1 * 2.0 * u0 * u1 + k2 * 4 * rho_u * u0 * u1 + p1 * 2.0 * u0 + 2.0 * p2 * u1;
    J[1] = J[2];
    J[3] = 1 + rad_dist_u + k1 * 2.0 * my_u + k2 * rho_u * 4 * my_u + 6 * p1 * u1 + 2.0 * p2 * u0;
    return;
  }
  // equidistant; atan_fixed: the same operation sequence as on the host (atan_fixed.h)
  const double k1 = c.d[0], k2 = c.d[1], k3 = c.d[2], k4 = c.d[3];
  const double r = sqrt(u0 * u0 + u1 * u1);
  const double theta = atan_fixed(r);
  const double theta2 = theta * theta;
  const double theta4 = theta2 * theta2;
  const double theta6 = theta4 * theta2;
  const double theta8 = theta4 * theta4;
  const double thetad = theta * (1.0 + k1 * theta2 + k2 * theta4 + k3 * theta6 + k4 * theta8);
  const double scaling = (r > 1e-8) ? thetad / r : 1.0;
  out[0] = scaling * u0;
  out[1] = scaling * u1;
  if (r > 1e-8) {
    equidistant_jacobian(u0, u1, k1, k2, k3, k4, [](double v) { return sqrt(v); }, J);  // equidistant_jacobian.h
  } else {
    J[0] = 1.0; J[1] = 0.0; J[2] = 0.0; J[3] = 1.0;
  }
}

__device__ inline bool backproject(const DeviceCamera& c, double px, double py, double dir[3]) {
  const double pd0 = (px - c.cu) * c.one_over_fu;
  const double pd1 = (py - c.cv) * c.one_over_fv;
  bool success = false;
  double x0 = pd0, x1 = pd1;
  if (c.distortion == OKVFE_DIST_NONE) {
    success = true;
  } else {
    const int n = c.distortion == OKVFE_DIST_RADTAN ? 5 : 20;
    for (int i = 0; i < n; ++i) {
      double xt[2], E[4];
      distort(c, x0, x1, xt, E);
      const double e0 = pd0 - xt[0], e1 = pd1 - xt[1];
      const double a = E[0] * E[0] + E[2] * E[2];
      const double b = E[0] * E[1] + E[2] * E[3];
      const double cc = E[1] * E[0] + E[3] * E[2];
      const double d = E[1] * E[1] + E[3] * E[3];
      const double det = a * d - b * cc;
      const double invdet = 1.0 / det;
      const double i00 = d * invdet, i01 = -b * invdet, i10 = -cc * invdet, i11 = a * invdet;
      const double b00 = i00 * E[0] + i01 * E[1];
      const double b01 = i00 * E[2] + i01 * E[3];
      const double b10 = i10 * E[0] + i11 * E[1];
      const double b11 = i10 * E[2] + i11 * E[3];
      const double du0 = b00 * e0 + b01 * e1;
      const double du1 = b10 * e0 + b11 * e1;
      x0 += du0;
      x1 += du1;
      const double chi2 = e0 * e0 + e1 * e1;
      if (chi2 < 1e-6) success = true;
      if (chi2 < 1e-15) {
        success = true;
        break;
      }
    }
  }
  dir[0] = x0;
  dir[1] = x1;
  dir[2] = 1.0;
  return success;
}


// PinholeCamera::project without Jacobian; status 0 = Successful, 1 = OutsideImage, 3 = Behind,
// 4 = Invalid (PinholeCamera.hpp:241-283, CameraBase.hpp:97-106)
__device__ inline int project(const DeviceCamera& c, int w, int h, const double p[3], double pt[2]) {
  if (fabs(p[2]) < 1.0e-12) return 4;
  const double rz = 1.0 / p[2];
  double dist[2], J[4];
  distort(c, p[0] * rz, p[1] * rz, dist, J);
  pt[0] = c.fu * dist[0] + c.cu;
  pt[1] = c.fv * dist[1] + c.cv;
  if (pt[0] < 0.0 || pt[1] < 0.0) return 1;
  if (pt[0] >= (double)w || pt[1] >= (double)h) return 1;
  return p[2] > 0.0 ? 0 : 3;
}

}  // namespace cam
}  // namespace okvfe
