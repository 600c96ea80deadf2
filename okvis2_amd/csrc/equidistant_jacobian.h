// equidistant_jacobian.h -- the ONE product copy of the equidistant distortion Jacobian.
//
// Attribution: the expression below keeps the operation order of the reference's MATLAB-generated
// code (okvis_cv/include/okvis/cameras/implementation/EquidistantDistortion.hpp:128-171, temporaries
// t2 .. t25) because that order fixes the FP64 rounding bit-exactness depends on -- Copyright (c) 2015
// Autonomous Systems Lab / ETH Zurich, (c) 2020 Smart Robotics Lab / Imperial College London, (c) 2024
// Smart Robotics Lab / Technical University of Munich, BSD 3-Clause (licence text in the header of that
// file; its conditions apply to this fragment).  Included by host_tables.cpp (host) and camera_dev.h
// (device): both call it with their own sqrt; atan comes from atan_fixed.h.  The oracle keeps its own
// copy on purpose (oracle/orc_camera.c: test infrastructure, independent of product headers).
#pragma once

#include "atan_fixed.h"

namespace okvfe {

// J = d(distorted point) / d(u0, u1), row-major {a, b, c, d}; valid for r = |u| > 1e-8
template <class Sqrt>
OKVFE_HD void equidistant_jacobian(double u0, double u1, double k1, double k2, double k3, double k4, Sqrt sq,
                                          double J[4]) {
  double t2, t3, t4, t6, t7, t8, t9, t11, t17, t18, t19, t20, t25;
  t2 = u0 * u0;
  t3 = u1 * u1;
  t4 = t2 + t3;
  t6 = atan_fixed(sq(t4));
  t7 = t6 * t6;
  t8 = 1.0 / sq(t4);
  t9 = t7 * t7;
  t11 = 1.0 / ((t2 + t3) + 1.0);
  t17 = (((k1 * t7 + k2 * t9) + k3 * t7 * t9) + k4 * (t9 * t9)) + 1.0;
  t18 = 1.0 / t4;
  t19 = 1.0 / sq(t4 * t4 * t4);
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
}

}  // namespace okvfe
