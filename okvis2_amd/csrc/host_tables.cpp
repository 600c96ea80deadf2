// host_tables.cpp -- host-side tables and camera input preparation of libokvfe.so (product code).
//
//   build_pattern          BRISK2 sampling pattern: recovered pair table + published ring geometry (data consumed by k_describe.hip).
//   build_uniformity_lut   31x31 radial stamp of the uniformity enforcement (k_select.hip).
//   build_awareness_maps   = okvis::cameras::PinholeCamera<D>::initialiseCameraAwarenessMaps
//                            (okvis_cv/include/okvis/cameras/implementation/PinholeCamera.hpp:180-208):
//                            per-pixel unit ray (CV_32FC3) and 2x3 image Jacobian (CV_32FC6).
//   host_backproject       = PinholeCamera<D>::backProject (PinholeCamera.hpp:574-593) with the
//                            Gauss-Newton undistortion of RadialTangentialDistortion.hpp:214-252 /
//                            EquidistantDistortion.hpp:319-351.
// FP64, explicit evaluation order, compiled with -ffp-contract=off.
//
// Attribution: distort() keeps the reference's operation order (bit-exact FP64 depends on it) and
// is a close transcription of RadialTangentialDistortion.hpp:111-135 and
// EquidistantDistortion.hpp:128-171 -- Copyright (c) 2015 Autonomous Systems Lab / ETH Zurich,
// (c) 2020 Smart Robotics Lab / Imperial College London, (c) 2024 Smart Robotics Lab / Technical
// University of Munich, BSD 3-Clause (licence text in the header of those files; its conditions
// apply to that fragment).  atan comes from atan_fixed.h (same sequence as the device).
#include <cmath>
#include <cstring>

#include "atan_fixed.h"
#include "brisk2_pairs.h"
#include "equidistant_jacobian.h"
#include "okvfe_internal.h"

namespace okvfe {

float pattern_reach(const Pattern& p) {
  double reach = 0.0;
  for (int i = 0; i < p.n_points && i < kPatternPoints; ++i)
    reach = std::fmax(reach, std::sqrt(static_cast<double>(p.px[i]) * p.px[i] + static_cast<double>(p.py[i]) * p.py[i]) +
                                 static_cast<double>(p.sigma_half[i]));
  return static_cast<float>(reach * 1.0001 + 1.0e-3);  // (rounded up: only ever used for a superset)
}

void fill_aware_lanes(Pattern* p) {
  const int extra = p->n_points > 64 ? p->n_points - 64 : 0;
  for (int l = 0; l < 64; ++l) {
    const int li = extra + l < p->n_points ? extra + l : 0;
    int32_t* r = p->aware_lane[l];
    std::memcpy(&r[0], &p->px[li], 4);
    std::memcpy(&r[1], &p->py[li], 4);
    std::memcpy(&r[2], &p->sigma_half[li], 4);
    r[3] = p->box_scaling[li];
    r[4] = p->box_scaling2[li];
    for (int j = 0; j < 3; ++j) {
      uint32_t v = 0;
      for (int half = 0; half < 2; ++half) {
        const int t = (2 * j + half) * 64 + l;  // bit t of the descriptor: word 2 j + half, lane l
        const uint32_t e = t < p->n_short ? (uint32_t)p->short_i[t] | ((uint32_t)p->short_j[t] << 8) : 0u;
        v |= e << (16 * half);
      }
      r[5 + j] = (int32_t)v;
    }
  }
}

void build_pattern(Pattern* p) {
  // The BRISK2 pattern as recovered from the 819 real node descriptors of the reference's vocabulary
  // (resources/small_voc.yml.gz; tools/pattern/README.md): 66 sample points -- centre, hexagon, rings of
  // 10 / 14 / 15 / 20 -- and the 384 short pairs of brisk2_pairs.h in the generator's loop order (all 384
  // bits live).  Assumed, not recoverable from bits: ring radii {0,1.4,2.9,4.9,7.4,10.8}*0.85 and box
  // half-side 1.3 * r * sin(pi/n) (published BRISK constants + a hexagon radius inside the interval the pair
  // rule allows), at the fixed scale of the non-scale-invariant extractor; pairs further apart than 8.2
  // feed the gradient orientation.
  const double ring_radius[6] = {0.0, 1.4, 2.9, 4.9, 7.4, 10.8};
  const int ring_points[6] = {1, 6, 10, 14, 15, 20};
  const double lb_range = std::log(30.0) / std::log(2.0);
  const int scale_index = static_cast<int>(64.0 / lb_range * (std::log(1.45 / 0.6) / std::log(2.0)) + 0.5);
  const double scale = std::pow(2.0, static_cast<double>(scale_index) * (lb_range / 64.0));
  std::memset(p, 0, sizeof(*p));
  double ux[kPatternPoints], uy[kPatternPoints];
  int n = 0;
  double reach = 0.0;
  for (int ring = 0; ring < 6; ++ring) {
    const double r = ring_radius[ring] * 0.85;
    for (int j = 0; j < ring_points[ring]; ++j, ++n) {
      const double alpha = static_cast<double>(j) * 2.0 * M_PI / static_cast<double>(ring_points[ring]);
      ux[n] = r * std::cos(alpha);
      uy[n] = r * std::sin(alpha);
      p->px[n] = static_cast<float>(scale * ux[n]);
      p->py[n] = static_cast<float>(scale * uy[n]);
      const double sigma = ring == 0 ? 1.3 * scale * 0.5
                                     : 1.3 * scale * r * std::sin(M_PI / static_cast<double>(ring_points[ring]));
      p->sigma_half[n] = static_cast<float>(sigma);
      reach = std::fmax(reach, scale * r + sigma);
    }
  }
  p->n_points = n;
  p->border = static_cast<int>(std::ceil(reach)) + 1;
  p->reach = pattern_reach(*p);
  for (int i = 0; i < kPatternPoints; ++i) {
    const float sg = i < n ? p->sigma_half[i] : 1.0f;
    float area = 4.0f * sg;
    area = area * sg;
    const int scaling = static_cast<int>(4194304.0f / area);
    const float s2 = static_cast<float>(scaling) * area;
    p->box_scaling[i] = scaling;
    p->box_scaling2[i] = static_cast<int>(s2 / 1024.0f);
  }
  static_assert(OKVFE_BRISK2_PAIR_N_PAIRS == 384, "48-byte rows");
  p->n_short = OKVFE_BRISK2_PAIR_N_PAIRS;
  for (int b = 0; b < OKVFE_BRISK2_PAIR_N_PAIRS; ++b) {
    p->short_i[b] = okvfe_brisk2_pair_i[b];
    p->short_j[b] = okvfe_brisk2_pair_j[b];
  }
  for (int i = 1; i < n; ++i) {
    for (int j = 0; j < i; ++j) {
      const double dx = ux[j] - ux[i], dy = uy[j] - uy[i];
      const double norm_sq = dx * dx + dy * dy;
      if (std::sqrt(norm_sq) > 8.2 && p->n_long < kMaxLongPairs) {
        p->long_i[p->n_long] = static_cast<uint8_t>(i);
        p->long_j[p->n_long] = static_cast<uint8_t>(j);
        p->long_wdx[p->n_long] = static_cast<int32_t>(std::floor((dx / norm_sq) * 2048.0 + 0.5));
        p->long_wdy[p->n_long] = static_cast<int32_t>(std::floor((dy / norm_sq) * 2048.0 + 0.5));
        ++p->n_long;
      }
    }
  }
  for (int k = 0; k < kRot; ++k) {
    const double a = static_cast<double>(k) * 2.0 * M_PI / 1024.0;
    p->rot_cos[k] = static_cast<int32_t>(std::lround(32768.0 * std::cos(a)));
    p->rot_sin[k] = static_cast<int32_t>(std::lround(32768.0 * std::sin(a)));
    p->rot_cosf[k] = static_cast<float>(std::cos(a));
    p->rot_sinf[k] = static_cast<float>(std::sin(a));
  }
  fill_aware_lanes(p);
}

// okvfe_config.box_scale: every smoothing box of the pattern `s` times wider (half-side in double, rounded once);
// the rim a keypoint must keep from the image border follows the farthest box
void scale_pattern_boxes(Pattern* p, float s) {
  double reach = 0.0;
  for (int i = 0; i < p->n_points; ++i) {
    p->sigma_half[i] = static_cast<float>(static_cast<double>(p->sigma_half[i]) * static_cast<double>(s));
    reach = std::fmax(reach, std::sqrt(static_cast<double>(p->px[i]) * p->px[i] + static_cast<double>(p->py[i]) * p->py[i]) +
                                 static_cast<double>(p->sigma_half[i]));
  }
  p->border = static_cast<int>(std::ceil(reach)) + 1;
  p->reach = pattern_reach(*p);
  for (int i = 0; i < kPatternPoints; ++i) {  // same float sequence as build_pattern
    const float sg = i < p->n_points ? p->sigma_half[i] : 1.0f;
    float area = 4.0f * sg;
    area = area * sg;
    const int scaling = static_cast<int>(4194304.0f / area);
    const float s2 = static_cast<float>(scaling) * area;
    p->box_scaling[i] = scaling;
    p->box_scaling2[i] = static_cast<int>(s2 / 1024.0f);
  }
  fill_aware_lanes(p);
}

int pattern_scale_index(float size) {
  const double lb_range = std::log(30.0) / std::log(2.0);
  if (!(size > 0.0f)) return 0;
  const double v = 64.0 / lb_range * (std::log(static_cast<double>(size) / (0.6 * 12.0)) / std::log(2.0)) + 0.5;
  if (!(v > 0.0)) return 0;
  if (v >= static_cast<double>(kPatternScales)) return kPatternScales - 1;
  return static_cast<int>(v);
}

void build_pattern_scales(const Pattern& base, PatternScales* out) {
  const double lb_range = std::log(30.0) / std::log(2.0);
  std::memset(out, 0, sizeof(*out));
  for (int s = 0; s < kPatternScales; ++s) {
    const double rel = std::pow(2.0, static_cast<double>(s - kBasicScale) * (lb_range / 64.0));
    for (int i = 0; i < kPatternPoints; ++i) {
      const bool on = i < base.n_points;
      if (s == kBasicScale) {
        out->px[s][i] = base.px[i];
        out->py[s][i] = base.py[i];
        out->sigma_half[s][i] = base.sigma_half[i];
      } else {
        out->px[s][i] = static_cast<float>(static_cast<double>(base.px[i]) * rel);
        out->py[s][i] = static_cast<float>(static_cast<double>(base.py[i]) * rel);
        out->sigma_half[s][i] = static_cast<float>(static_cast<double>(base.sigma_half[i]) * rel);
      }
      const float sg = on ? out->sigma_half[s][i] : 1.0f;  // same float sequence as build_pattern
      float area = 4.0f * sg;
      area = area * sg;
      const int scaling = static_cast<int>(4194304.0f / area);
      const float s2 = static_cast<float>(scaling) * area;
      out->box_scaling[s][i] = scaling;
      out->box_scaling2[s][i] = static_cast<int>(s2 / 1024.0f);
    }
    out->border[s] = s == kBasicScale ? base.border
                                      : static_cast<int>(std::ceil(rel * static_cast<double>(base.border - 1))) + 1;
    // smallest float size whose index reaches s: start at the analytic boundary, then step
    // float by float on the formula itself (it is what the oracle evaluates per keypoint)
    if (s == 0) {
      out->size_from[0] = 0.0f;
      continue;
    }
    float b = static_cast<float>(7.2 * std::pow(2.0, (static_cast<double>(s) - 0.5) * (lb_range / 64.0)));
    while (pattern_scale_index(b) >= s) b = std::nextafterf(b, 0.0f);
    while (pattern_scale_index(b) < s) b = std::nextafterf(b, INFINITY);
    out->size_from[s] = b;
  }
}

void build_uniformity_lut(float lut[kLutFloats]) {
  std::memset(lut, 0, sizeof(float) * kLutFloats);
  for (int y = 0; y < 31; ++y)
    for (int x = 0; x < 31; ++x) {
      const int d2 = (15 - x) * (15 - x) + (15 - y) * (15 - y);
      const double v = 1.0 - static_cast<double>(d2) / 225.0;
      lut[y * 31 + x] = static_cast<float>(v > 0.0 ? v : 0.0);
    }
  // compacted stamp: only the cells with a non-zero weight, raster order, padded to kStampSlots
  // with weight 0 (k_select_grid.hip: 11 cells per lane instead of 16)
  int j = 0;
  for (int t = 0; t < 31 * 31; ++t) {
    if (!(lut[t] > 0.0f)) continue;
    const uint32_t pos = (static_cast<uint32_t>(t / 31) << 8) | static_cast<uint32_t>(t % 31);
    std::memcpy(&lut[kStampTableOffset + 2 * j], &pos, 4);
    lut[kStampTableOffset + 2 * j + 1] = lut[t];
    ++j;
  }
  const uint32_t centre = (15u << 8) | 15u;
  for (; j < kStampSlots; ++j) std::memcpy(&lut[kStampTableOffset + 2 * j], &centre, 4);
}

namespace {

struct Vec2 {
  double x, y;
};
struct Mat2 {
  double a, b, c, d;  // [a b; c d]
};

void distort(const okvfe_camera& cam, Vec2 u, Vec2* out, Mat2* J) {
  if (cam.distortion == OKVFE_DIST_NONE) {
    *out = u;
    if (J) *J = {1.0, 0.0, 0.0, 1.0};
    return;
  }
  const double u0 = u.x, u1 = u.y;
  if (cam.distortion == OKVFE_DIST_RADTAN) {
    const double k1 = cam.d[0], k2 = cam.d[1], p1 = cam.d[2], p2 = cam.d[3];
    const double mx_u = u0 * u0;
    const double my_u = u1 * u1;
    const double mxy_u = u0 * u1;
    const double rho_u = mx_u + my_u;
    const double rad_dist_u = k1 * rho_u + k2 * rho_u * rho_u;
    out->x = u0 + u0 * rad_dist_u + 2.0 * p1 * mxy_u + p2 * (rho_u + 2.0 * mx_u);
    out->y = u1 + u1 * rad_dist_u + 2.0 * p2 * mxy_u + p1 * (rho_u + 2.0 * my_u);
    if (J) {
      J->a = 1 + rad_dist_u + k1 * 2.0 * mx_u + k2 * rho_u * 4 * mx_u + 2.0 * p1 * u1 + 6 * p2 * u0;
      J->c = k1 * 2.0 * u0 * u1 + k2 * 4 * rho_u * u0 * u1 + p1 * 2.0 * u0 + 2.0 * p2 * u1;
      J->b = J->c;
      J->d = 1 + rad_dist_u + k1 * 2.0 * my_u + k2 * rho_u * 4 * my_u + 6 * p1 * u1 + 2.0 * p2 * u0;
    }
    return;
  }
  const double k1 = cam.d[0], k2 = cam.d[1], k3 = cam.d[2], k4 = cam.d[3];
  const double r = std::sqrt(u0 * u0 + u1 * u1);
  const double theta = atan_fixed(r);
  const double theta2 = theta * theta;
  const double theta4 = theta2 * theta2;
  const double theta6 = theta4 * theta2;
  const double theta8 = theta4 * theta4;
  const double thetad = theta * (1.0 + k1 * theta2 + k2 * theta4 + k3 * theta6 + k4 * theta8);
  const double scaling = (r > 1e-8) ? thetad / r : 1.0;
  out->x = scaling * u0;
  out->y = scaling * u1;
  if (!J) return;
  if (r > 1e-8) {
    // generated expression of the reference (EquidistantDistortion.hpp:128-171): equidistant_jacobian.h
    double j4[4];
    equidistant_jacobian(u0, u1, k1, k2, k3, k4, [](double v) { return std::sqrt(v); }, j4);
    J->a = j4[0];
    J->b = j4[1];
    J->c = j4[2];
    J->d = j4[3];
  } else {
    *J = {1.0, 0.0, 0.0, 1.0};
  }
}

bool undistort(const okvfe_camera& cam, Vec2 pd, Vec2* out) {
  if (cam.distortion == OKVFE_DIST_NONE) {
    *out = pd;
    return true;
  }
  const int iterations = cam.distortion == OKVFE_DIST_RADTAN ? 5 : 20;
  Vec2 x_bar = pd;
  bool success = false;
  for (int it = 0; it < iterations; ++it) {
    Vec2 x_tmp;
    Mat2 E;
    distort(cam, x_bar, &x_tmp, &E);
    const double e0 = pd.x - x_tmp.x, e1 = pd.y - x_tmp.y;
    // E2 = E^T E ; du = (inv(E2) * E^T) * e
    const double a = E.a * E.a + E.c * E.c;
    const double b = E.a * E.b + E.c * E.d;
    const double c = E.b * E.a + E.d * E.c;
    const double d = E.b * E.b + E.d * E.d;
    const double det = a * d - b * c;
    const double invdet = 1.0 / det;
    const double i00 = d * invdet, i01 = -b * invdet, i10 = -c * invdet, i11 = a * invdet;
    const double b00 = i00 * E.a + i01 * E.b;
    const double b01 = i00 * E.c + i01 * E.d;
    const double b10 = i10 * E.a + i11 * E.b;
    const double b11 = i10 * E.c + i11 * E.d;
    x_bar.x += b00 * e0 + b01 * e1;
    x_bar.y += b10 * e0 + b11 * e1;
    const double chi2 = e0 * e0 + e1 * e1;
    if (chi2 < 1e-6) success = true;
    if (chi2 < 1e-15) {
      success = true;
      break;
    }
  }
  *out = x_bar;
  return success;
}

// status 0 = Successful (only that one matters here)
int project_point(const okvfe_camera& cam, const double p[3], double* out_x, double* out_y, double J23[6]) {
  if (std::fabs(p[2]) < 1.0e-12) return 4;
  const double rz = 1.0 / p[2];
  const double rz2 = rz * rz;
  Vec2 und{p[0] * rz, p[1] * rz}, dist;
  Mat2 D;
  distort(cam, und, &dist, &D);
  J23[0] = cam.fu * D.a * rz;
  J23[1] = cam.fu * D.b * rz;
  J23[2] = -cam.fu * (p[0] * D.a + p[1] * D.b) * rz2;
  J23[3] = cam.fv * D.c * rz;
  J23[4] = cam.fv * D.d * rz;
  J23[5] = -cam.fv * (p[0] * D.c + p[1] * D.d) * rz2;
  const double px = cam.fu * dist.x + cam.cu;
  const double py = cam.fv * dist.y + cam.cv;
  *out_x = px;
  *out_y = py;
  if (px < 0.0 || py < 0.0) return 1;
  if (px >= static_cast<double>(cam.width) || py >= static_cast<double>(cam.height)) return 1;
  return p[2] > 0.0 ? 0 : 3;
}

}  // namespace

bool host_backproject(const okvfe_camera& cam, double px, double py, double dir[3]) {
  const double one_over_fu = 1.0 / cam.fu, one_over_fv = 1.0 / cam.fv;
  Vec2 p2{(px - cam.cu) * one_over_fu, (py - cam.cv) * one_over_fv}, und;
  const bool ok = undistort(cam, p2, &und);
  dir[0] = und.x;
  dir[1] = und.y;
  dir[2] = 1.0;
  return ok;
}

// = NCameraSystem::computeOverlaps for one ordered camera pair (okvis_cv/src/NCameraSystem.cpp:48-119)
bool camera_overlap(const okvfe_camera& cam, const okvfe_camera& other, const double R[9], uint8_t* mask) {
  bool any = false;
  for (int u = 0; u < cam.width; ++u) {
    for (int v = 0; v < cam.height; ++v) {
      double ray[3], ro[3], ver[3], J[6];
      host_backproject(cam, static_cast<double>(u), static_cast<double>(v), ray);
      for (int i = 0; i < 3; ++i) ro[i] = R[3 * i] * ray[0] + R[3 * i + 1] * ray[1] + R[3 * i + 2] * ray[2];
      bool hit = false;
      double px = 0.0, py = 0.0;
      if (project_point(other, ro, &px, &py, J) == 0) {
        host_backproject(other, px, py, ver);
        const double na = std::sqrt(ro[0] * ro[0] + ro[1] * ro[1] + ro[2] * ro[2]);
        const double nb = std::sqrt(ver[0] * ver[0] + ver[1] * ver[1] + ver[2] * ver[2]);
        const double dot = (ro[0] / na) * (ver[0] / nb) + (ro[1] / na) * (ver[1] / nb) + (ro[2] / na) * (ver[2] / nb);
        hit = std::fabs(dot - 1.0) < 1.0e-10;
      }
      if (mask) mask[static_cast<size_t>(v) * cam.width + u] = hit ? 1 : 0;
      any = any || hit;
    }
  }
  return any;
}

void build_awareness_maps(const okvfe_camera& cam, float* rays, float* jac) {
  for (int v = 0; v < cam.height; ++v) {
    for (int u = 0; u < cam.width; ++u) {
      double ray[3];
      if (host_backproject(cam, static_cast<double>(u), static_cast<double>(v), ray)) {
        const double n = std::sqrt(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
        ray[0] /= n;
        ray[1] /= n;
        ray[2] /= n;
      } else {
        ray[0] = ray[1] = ray[2] = 0.0;
      }
      const size_t px = static_cast<size_t>(v) * cam.width + u;
      for (int i = 0; i < 3; ++i) rays[px * 3 + i] = static_cast<float>(ray[i]);
      double J[6];
      double qx, qy;
      const bool ok = project_point(cam, ray, &qx, &qy, J) == 0;
      // the reference leaves failed entries uninitialised; they are defined as zero here and
      // never read for a kept keypoint (a zero ray removes the keypoint)
      for (int i = 0; i < 6; ++i) jac[px * 6 + i] = ok ? static_cast<float>(J[i]) : 0.0f;
    }
  }
}

}  // namespace okvfe
