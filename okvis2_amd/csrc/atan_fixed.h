// atan_fixed.h -- FP64 arctangent as ONE fixed sequence of IEEE-754 operations, compiled for the
// host (host_tables.cpp: awareness maps, overlap masks) and for gfx950 (camera_dev.h: on-device
// back-projection / projection of equidistant cameras).
//
// Why: the reference's EquidistantDistortion calls libm's atan
// (okvis_cv/include/okvis/cameras/implementation/EquidistantDistortion.hpp:98,138); the device
// math library's atan differs from a host libm in the last bit for some arguments, which made the
// equidistant back-projections agree only to ~1e-14 in round 1.  With the same +,-,*,/ sequence on
// both sides (no FMA: the library is built with -ffp-contract=off; no table look-ups that could be
// reordered) host tables, device kernels and the CPU oracle's own copy (oracle/orc_camera.c) give
// identical bit patterns, so match gates cannot flip between host and device.
//
// Algorithm: the classic argument reduction to [0, 7/16] around the break points 7/16, 11/16,
// 19/16, 39/16 with an odd minimax polynomial of degree 23 (the published fdlibm scheme of
// K.C. Ng, 1993; error < 1 ulp).  Against glibc's (nearly correctly rounded) atan this differs by
// at most 1 ulp (tests/test_oracle_pins.py::test_fixed_atan_within_one_ulp_of_libm).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define OKVFE_HD __host__ __device__ inline
#else
#define OKVFE_HD inline
#endif

namespace okvfe {

OKVFE_HD double atan_fixed(double x) {
  // atan(0.5), atan(1), atan(1.5), atan(inf) as hi + lo pairs
  const double hi0 = 4.63647609000806093515e-01, lo0 = 2.26987774529616870924e-17;
  const double hi1 = 7.85398163397448278999e-01, lo1 = 3.06161699786838301793e-17;
  const double hi2 = 9.82793723247329054082e-01, lo2 = 1.39033110312309984516e-17;
  const double hi3 = 1.57079632679489655800e+00, lo3 = 6.12323399573676603587e-17;
  const double a0 = 3.33333333333329318027e-01, a1 = -1.99999999998764832476e-01,
               a2 = 1.42857142725034663711e-01, a3 = -1.11111104054623557880e-01,
               a4 = 9.09088713343650656196e-02, a5 = -7.69187620504482999495e-02,
               a6 = 6.66107313738753120669e-02, a7 = -5.83357013379057348645e-02,
               a8 = 4.97687799461593236017e-02, a9 = -3.65315727442169155270e-02,
               a10 = 1.62858201153657823623e-02;
  if (x != x) return x;
  const bool neg = x < 0.0;
  double ax = neg ? -x : x;
  if (ax >= 7.378697629483821e19) {  // 2^66: atan = +-pi/2 to double precision
    const double r = hi3 + lo3;
    return neg ? -r : r;
  }
  bool reduced = true;
  double t, h = 0.0, l = 0.0;
  if (ax < 0.4375) {
    if (ax < 1.862645149230957e-09) return x;  // 2^-29
    reduced = false;
    t = ax;
  } else if (ax < 1.1875) {
    if (ax < 0.6875) {
      h = hi0; l = lo0;
      t = (2.0 * ax - 1.0) / (2.0 + ax);
    } else {
      h = hi1; l = lo1;
      t = (ax - 1.0) / (ax + 1.0);
    }
  } else if (ax < 2.4375) {
    h = hi2; l = lo2;
    t = (ax - 1.5) / (1.0 + 1.5 * ax);
  } else {
    h = hi3; l = lo3;
    t = -1.0 / ax;
  }
  const double z = t * t;
  const double w = z * z;
  const double s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
  const double s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
  double r;
  if (!reduced) {
    r = t - t * (s1 + s2);
  } else {
    r = h - ((t * (s1 + s2) - l) - t);
  }
  return neg ? -r : r;
}

// FP64 arccosine as one fixed sequence (the published fdlibm scheme: rational approximation of
// asin on [0, 1/2], acos(x) = 2 asin(sqrt((1 - x) / 2)) above it; error < 1 ulp).  Used where the
// reference calls libm's acos in a comparison that decides an outcome (descriptor-view pooling
// score, okvis_frontend/src/Frontend.cpp:1312); host code and kernels share it.
OKVFE_HD double acos_fixed(double x) {
  const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
               pi = 3.14159265358979311600e+00;
  const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
               pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
               pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
               qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
               qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
  if (x != x) return x;
  const double ax = x < 0.0 ? -x : x;
  if (ax >= 1.0) {
    if (x == 1.0) return 0.0;
    if (x == -1.0) return pi + 2.0 * pio2_lo;
    return (x - x) / (x - x);  // NaN
  }
  if (ax < 0.5) {
    if (ax < 6.938893903907228e-18) return pio2_hi + pio2_lo;  // 2^-57
    const double z = x * x;
    const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const double r = p / q;
    return pio2_hi - (x - (pio2_lo - x * r));
  }
  if (x < 0.0) {
    const double z = (1.0 + x) * 0.5;
    const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const double s = sqrt(z);
    const double r = p / q;
    const double w = r * s - pio2_lo;
    return pi - 2.0 * (s + w);
  }
  const double z = (1.0 - x) * 0.5;
  const double s = sqrt(z);
  // df = s with the low 32 bits of its mantissa cleared
  unsigned long long bits;
  __builtin_memcpy(&bits, &s, 8);
  bits &= 0xFFFFFFFF00000000ull;
  double df;
  __builtin_memcpy(&df, &bits, 8);
  const double c = (z - df * df) / (s + df);
  const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
  const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
  const double r = p / q;
  const double w = r * s + c;
  return 2.0 * (df + w);
}

}  // namespace okvfe
