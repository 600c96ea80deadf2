/*
 * orc_describe.c -- oracle restatement of the extractor behind
 *   brisk::BriskDescriptorExtractor(rotationInvariant, scaleInvariant)
 * constructed at okvis_frontend/src/Frontend.cpp:2410-2412 (defaults true/false
 * at :142-143), configured through setCameraProperties(rays, imageJacobians, fu)
 * (:239-242) and setExtractionDirection(Vec3f) (:249-251), and invoked through
 * cv::DescriptorExtractor::compute at
 * okvis_cv/include/okvis/implementation/Frame.hpp:167 ("some keypoints are
 * removed there", :146).  Output rows are 48 bytes (FBrisk.hpp:35).
 *
 * TEST INFRASTRUCTURE ONLY (see okvfe_oracle.h).  PARITY UNPINNED: the brisk
 * submodule (and with it the BRISK2 pattern file) is absent.  This restates
 * the published BRISK descriptor arithmetic: a ring pattern of sample points,
 * each smoothed by a box of half-side sigma with sub-pixel edge weights, short
 * point pairs compared into bits.  The pattern is DATA (orc_pattern).  Its
 * PAIR TABLE AND BIT ORDER are pinned on the 819 real BRISK2 descriptors of
 * resources/small_voc.yml.gz (tools/pattern/README.md; 7 impossible outcomes
 * over 740 implied point triangles, 384 live bits); radii and box sizes remain
 * the published BRISK constants (assumed).
 *
 * Orientation modes:
 *   UPRIGHT       rotationInvariant=false: M = I.
 *   GRADIENT      rotationInvariant=true without camera properties: published
 *                 long-pair intensity-gradient direction, quantised to 1024
 *                 steps by an exact integer arg-max (no atan2).
 *   CAMERA_AWARE  OKVIS2 production: the pattern is laid out on the tangent
 *                 plane of the keypoint's viewing ray with its +y axis along
 *                 the extraction direction (gravity in the camera frame) and
 *                 mapped to the image by the local 2x3 image Jacobian:
 *                 M = J * [e_x e_y] / fu   (rays / Jacobians from
 *                 okvis_cv .../cameras/implementation/PinholeCamera.hpp:180-208).
 * Sample i is taken at kp + M * p_i.
 */
#include "okvfe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- pattern ------------------------------------------------------------------------------ */
#include "brisk2_pairs.h"

/* geometry shared by both builders: rings of `number[]` points at `radius[] * 0.85`, box half-side
 * 1.3 * r * sin(pi / n) (centre: 1.3 * 0.5), everything at the fixed scale of the non-scale-invariant
 * extractor: index 17 of 64 scales over a range of 30
 * = max(int(64/lb(30) * lb(1.45*12/(0.6*12)) + 0.5), 0) */
static void pattern_points(orc_pattern* p, const double* radius, const int* number, int rings, double* ux,
                           double* uy) {
  const double pattern_scale = 0.85, sigma_scale = 1.3;
  const double lb_scalerange = log(30.0) / log(2.0);
  const int basicscale = (int)(64.0 / lb_scalerange * (log(1.45 / 0.6) / log(2.0)) + 0.5);
  const double scale = pow(2.0, (double)basicscale * (lb_scalerange / 64.0));
  memset(p, 0, sizeof(*p));
  int n = 0;
  double border = 0.0;
  for (int ring = 0; ring < rings; ++ring) {
    const double r = radius[ring] * pattern_scale;
    for (int j = 0; j < number[ring]; ++j) {
      const double alpha = (double)j * 2.0 * M_PI / (double)number[ring];
      ux[n] = r * cos(alpha);
      uy[n] = r * sin(alpha);
      p->px[n] = (float)(scale * ux[n]);
      p->py[n] = (float)(scale * uy[n]);
      double sigma;
      if (radius[ring] == 0.0)
        sigma = sigma_scale * scale * 0.5;
      else
        sigma = sigma_scale * scale * r * sin(M_PI / (double)number[ring]);
      p->sigma_half[n] = (float)sigma;
      const double ext = scale * r + sigma;
      if (ext > border) border = ext;
      ++n;
    }
  }
  p->n_points = n;
  p->border = (int)ceil(border) + 1;
  for (int k = 0; k < 1024; ++k) {
    const double a = (double)k * 2.0 * M_PI / 1024.0;
    p->rot_cos[k] = (int32_t)lround(32768.0 * cos(a));
    p->rot_sin[k] = (int32_t)lround(32768.0 * sin(a));
    p->rot_cosf[k] = (float)cos(a);
    p->rot_sinf[k] = (float)sin(a);
  }
}
/* long pairs (gradient orientation): every pair further apart than d_min, published weights */
static void pattern_long_pairs(orc_pattern* p, const double* ux, const double* uy, double d_min) {
  for (int i = 1; i < p->n_points; ++i) {
    for (int j = 0; j < i; ++j) {
      const double dx = ux[j] - ux[i], dy = uy[j] - uy[i];
      const double norm_sq = dx * dx + dy * dy;
      if (sqrt(norm_sq) > d_min && p->n_long < ORC_MAX_LONG_PAIRS) {
        p->long_i[p->n_long] = (uint8_t)i;
        p->long_j[p->n_long] = (uint8_t)j;
        p->long_wdx[p->n_long] = (int32_t)floor((dx / norm_sq) * 2048.0 + 0.5);
        p->long_wdy[p->n_long] = (int32_t)floor((dy / norm_sq) * 2048.0 + 0.5);
        p->n_long++;
      }
    }
  }
}

/* Default: the BRISK2 pattern as recovered from the reference's vocabulary (tools/pattern/README.md):
 * 66 points -- centre, hexagon, rings of 10 / 14 / 15 / 20 -- and the 384 pairs of brisk2_pairs.h in the
 * generator's loop order.  Radii and box sizes are the published BRISK constants plus a hexagon radius
 * from the interval the pair rule leaves (assumptions, stated in the README). */
void orc_pattern_build(orc_pattern* p) {
  static const double radius[6] = {0.0, 1.4, 2.9, 4.9, 7.4, 10.8};
  static const int number[6] = {1, 6, 10, 14, 15, 20};
  double ux[ORC_PATTERN_POINTS], uy[ORC_PATTERN_POINTS]; /* unscaled */
  pattern_points(p, radius, number, 6, ux, uy);
  p->n_short = ORC_BRISK2_PAIR_N_PAIRS;
  for (int b = 0; b < ORC_BRISK2_PAIR_N_PAIRS; ++b) {
    p->short_i[b] = orc_brisk2_pair_i[b];
    p->short_j[b] = orc_brisk2_pair_j[b];
  }
  pattern_long_pairs(p, ux, uy, 8.2);
}

/* The published BRISK form used as the default until round 5: 60 points on {1,10,14,15,20} rings and the
 * 383 pairs closer than 5.10 (bit 383 stays 0).  Kept as a second pattern for the pattern-is-data tests. */
void orc_pattern_build_published(orc_pattern* p) {
  static const double radius[5] = {0.0, 2.9, 4.9, 7.4, 10.8};
  static const int number[5] = {1, 10, 14, 15, 20};
  double ux[ORC_PATTERN_POINTS], uy[ORC_PATTERN_POINTS];
  pattern_points(p, radius, number, 5, ux, uy);
  for (int i = 1; i < p->n_points; ++i)
    for (int j = 0; j < i; ++j) {
      const double dx = ux[j] - ux[i], dy = uy[j] - uy[i];
      if (sqrt(dx * dx + dy * dy) < 5.10 && p->n_short < 384) {
        p->short_i[p->n_short] = (uint8_t)i;
        p->short_j[p->n_short] = (uint8_t)j;
        p->n_short++;
      }
    }
  pattern_long_pairs(p, ux, uy, 8.2);
}

/* ---- scale invariance (scaleInvariant = true, Frontend.hpp:235-237 / Frontend.cpp:2410-2412) ----
 * Published BRISK: the pattern exists at 64 scales spanning a factor of 30, scale step
 * 2^(lb(30)/64); a keypoint of diameter `size` uses index
 *   max(int(64 / lb(30) * lb(size / (0.6 * 12)) + 0.5), 0), at most 63
 * (the non-scale-invariant extractor is the same formula at size = 1.45 * 12: index 17).  The
 * pattern at index i is the base pattern (index 17: orc_pattern_build, or any pattern installed as
 * data) with offsets, box half-sides and reach multiplied by 2^((i - 17) * lb(30) / 64); pair
 * tables and gradient weights are defined on the unit pattern and do not change. */
#define ORC_SCALES 64
#define ORC_BASIC_SCALE 17
int orc_scale_index(float size) {
  const double lb_scalerange = log(30.0) / log(2.0);
  if (!(size > 0.0f)) return 0;
  const double v = 64.0 / lb_scalerange * (log((double)size / (0.6 * 12.0)) / log(2.0)) + 0.5;
  if (!(v > 0.0)) return 0;
  if (v >= (double)ORC_SCALES) return ORC_SCALES - 1;
  return (int)v;
}
void orc_pattern_scaled(const orc_pattern* base, int index, orc_pattern* out) {
  const double lb_scalerange = log(30.0) / log(2.0);
  const double rel = pow(2.0, (double)(index - ORC_BASIC_SCALE) * (lb_scalerange / 64.0));
  *out = *base;
  if (index == ORC_BASIC_SCALE) return;
  for (int i = 0; i < base->n_points; ++i) {
    out->px[i] = (float)((double)base->px[i] * rel);
    out->py[i] = (float)((double)base->py[i] * rel);
    out->sigma_half[i] = (float)((double)base->sigma_half[i] * rel);
  }
  out->border = (int)ceil(rel * (double)(base->border - 1)) + 1;
}

/* ---- integral image (exclusive: I[y][x] = sum of rows < y, cols < x) ----------------------- */
void orc_integral(const uint8_t* img, int w, int h, int stride, int32_t* integral) {
  const int iw = w + 1;
  for (int x = 0; x <= w; ++x) integral[x] = 0;
  for (int y = 0; y < h; ++y) {
    int32_t row = 0;
    integral[(size_t)(y + 1) * iw] = 0;
    for (int x = 0; x < w; ++x) {
      row += img[(size_t)y * stride + x];
      integral[(size_t)(y + 1) * iw + x + 1] = integral[(size_t)y * iw + x + 1] + row;
    }
  }
}

/* ---- smoothed intensity ------------------------------------------------------------------- */
/* Box of half-side sigma_half centred at (xf, yf): interior pixels weigh
 * `scaling`, rim pixels weigh by their covered fraction; the result is
 * 1024 * mean intensity.  The caller guarantees the box lies inside the image.
 * `integral` may be NULL (sums are then taken directly; identical result). */
int orc_smoothed_intensity(const uint8_t* img, const int32_t* integral, int w, int h, int stride,
                           float xf, float yf, float sigma_half) {
  (void)h;
  if (sigma_half < 0.5f) {
    const int x = (int)xf, y = (int)yf;
    const int r_x = (int)((xf - (float)x) * 1024.0f);
    const int r_y = (int)((yf - (float)y) * 1024.0f);
    const int r_x_1 = 1024 - r_x, r_y_1 = 1024 - r_y;
    const uint8_t* ptr = img + (size_t)y * stride + x;
    int ret = r_x_1 * r_y_1 * (int)ptr[0];
    ret += r_x * r_y_1 * (int)ptr[1];
    ret += r_x * r_y * (int)ptr[stride + 1];
    ret += r_x_1 * r_y * (int)ptr[stride];
    return (ret + 512) / 1024;
  }
  float area = 4.0f * sigma_half;
  area = area * sigma_half;
  const int scaling = (int)(4194304.0f / area);
  float s2 = (float)scaling * area;
  const int scaling2 = (int)(s2 / 1024.0f);
  const float x_1 = xf - sigma_half, x1 = xf + sigma_half;
  const float y_1 = yf - sigma_half, y1 = yf + sigma_half;
  const int x_left = (int)(x_1 + 0.5f), y_top = (int)(y_1 + 0.5f);
  const int x_right = (int)(x1 + 0.5f), y_bottom = (int)(y1 + 0.5f);
  float r_x_1 = (float)x_left - x_1; r_x_1 = r_x_1 + 0.5f;
  float r_y_1 = (float)y_top - y_1;  r_y_1 = r_y_1 + 0.5f;
  float r_x1 = x1 - (float)x_right;  r_x1 = r_x1 + 0.5f;
  float r_y1 = y1 - (float)y_bottom; r_y1 = r_y1 + 0.5f;
  const float fs = (float)scaling;
  float t;
  t = r_x_1 * r_y_1; const int A = (int)(t * fs);
  t = r_x1 * r_y_1;  const int B = (int)(t * fs);
  t = r_x1 * r_y1;   const int C = (int)(t * fs);
  t = r_x_1 * r_y1;  const int D = (int)(t * fs);
  const int r_x_1_i = (int)(r_x_1 * fs), r_y_1_i = (int)(r_y_1 * fs);
  const int r_x1_i = (int)(r_x1 * fs), r_y1_i = (int)(r_y1 * fs);
  int ret = A * (int)img[(size_t)y_top * stride + x_left];
  ret += B * (int)img[(size_t)y_top * stride + x_right];
  ret += C * (int)img[(size_t)y_bottom * stride + x_right];
  ret += D * (int)img[(size_t)y_bottom * stride + x_left];
  int upper = 0, middle = 0, left = 0, right = 0, bottom = 0;
  if (integral) {
    const int iw = w + 1;
#define RECT(xa, ya, xb, yb)                                                          \
  (integral[(size_t)(yb)*iw + (xb)] - integral[(size_t)(ya)*iw + (xb)] -             \
   integral[(size_t)(yb)*iw + (xa)] + integral[(size_t)(ya)*iw + (xa)])
    upper = RECT(x_left + 1, y_top, x_right, y_top + 1);
    middle = RECT(x_left + 1, y_top + 1, x_right, y_bottom);
    left = RECT(x_left, y_top + 1, x_left + 1, y_bottom);
    right = RECT(x_right, y_top + 1, x_right + 1, y_bottom);
    bottom = RECT(x_left + 1, y_bottom, x_right, y_bottom + 1);
#undef RECT
  } else {
    for (int x = x_left + 1; x < x_right; ++x) {
      upper += img[(size_t)y_top * stride + x];
      bottom += img[(size_t)y_bottom * stride + x];
    }
    for (int y = y_top + 1; y < y_bottom; ++y) {
      left += img[(size_t)y * stride + x_left];
      right += img[(size_t)y * stride + x_right];
      for (int x = x_left + 1; x < x_right; ++x) middle += img[(size_t)y * stride + x];
    }
  }
  ret += upper * r_y_1_i + middle * scaling + left * r_x_1_i + right * r_x1_i + bottom * r_y1_i;
  return (ret + scaling2 / 2) / scaling2;
}

/* all boxes of a keypoint inside the image?  NaN-safe (NaN compares false). */
static int sample_positions(const orc_pattern* pat, const float M[4], float kx, float ky, int w,
                            int h, float* xs, float* ys) {
  int ok = 1;
  for (int i = 0; i < pat->n_points; ++i) {
    float a = M[0] * pat->px[i];
    float b = M[1] * pat->py[i];
    a = a + b;
    const float xf = kx + a;
    float c = M[2] * pat->px[i];
    float d = M[3] * pat->py[i];
    c = c + d;
    const float yf = ky + c;
    const float sg = pat->sigma_half[i];
    const float x_1 = xf - sg, x1 = xf + sg, y_1 = yf - sg, y1 = yf + sg;
    if (!(x_1 >= 0.0f && y_1 >= 0.0f && x1 < (float)(w - 1) && y1 < (float)(h - 1))) ok = 0;
    xs[i] = xf;
    ys[i] = yf;
  }
  return ok;
}

/* M for the camera-aware mode; returns 0 when the ray at the keypoint is unusable */
static int camera_aware_matrix(const float* rays, const float* jac, int w, float fu,
                               const float dir[3], float kx, float ky, float M[4]) {
  const int u = (int)(kx + 0.5f), v = (int)(ky + 0.5f);
  const float* r = rays + ((size_t)v * w + u) * 3;
  const float* J = jac + ((size_t)v * w + u) * 6;
  if (r[0] == 0.0f && r[1] == 0.0f && r[2] == 0.0f) return 0;
  float ey[3];
  float n2 = 0.0f;
  /* candidates for the in-plane "down" direction: dir, then camera +y, then camera +x */
  const float cand[3][3] = {{dir[0], dir[1], dir[2]}, {0.0f, 1.0f, 0.0f}, {1.0f, 0.0f, 0.0f}};
  for (int c = 0; c < 3; ++c) {
    const float* g = cand[c];
    float gr = g[0] * r[0];
    float t = g[1] * r[1];
    gr = gr + t;
    t = g[2] * r[2];
    gr = gr + t;
    for (int i = 0; i < 3; ++i) {
      t = gr * r[i];
      ey[i] = g[i] - t;
    }
    n2 = ey[0] * ey[0];
    t = ey[1] * ey[1];
    n2 = n2 + t;
    t = ey[2] * ey[2];
    n2 = n2 + t;
    if (n2 >= 1.0e-12f) break;
  }
  if (!(n2 >= 1.0e-12f)) return 0;
  const float n = sqrtf(n2);
  ey[0] = ey[0] / n;
  ey[1] = ey[1] / n;
  ey[2] = ey[2] / n;
  float ex[3], t1, t2;
  t1 = ey[1] * r[2]; t2 = ey[2] * r[1]; ex[0] = t1 - t2;
  t1 = ey[2] * r[0]; t2 = ey[0] * r[2]; ex[1] = t1 - t2;
  t1 = ey[0] * r[1]; t2 = ey[1] * r[0]; ex[2] = t1 - t2;
  float s;
  s = J[0] * ex[0]; t1 = J[1] * ex[1]; s = s + t1; t1 = J[2] * ex[2]; s = s + t1; M[0] = s / fu;
  s = J[0] * ey[0]; t1 = J[1] * ey[1]; s = s + t1; t1 = J[2] * ey[2]; s = s + t1; M[1] = s / fu;
  s = J[3] * ex[0]; t1 = J[4] * ex[1]; s = s + t1; t1 = J[5] * ex[2]; s = s + t1; M[2] = s / fu;
  s = J[3] * ey[0]; t1 = J[4] * ey[1]; s = s + t1; t1 = J[5] * ey[2]; s = s + t1; M[3] = s / fu;
  return 1;
}

static int describe_impl(const uint8_t* img, int w, int h, int stride, const orc_pattern* base, int mode,
                         const float* rays_hw3, const float* jac_hw6, float fu, const float dir[3],
                         orc_keypoint* kps, int n, uint8_t* desc, int scale_invariant) {
  int32_t* integral = (int32_t*)malloc((size_t)(w + 1) * (h + 1) * sizeof(int32_t));
  orc_integral(img, w, h, stride, integral);
  orc_pattern* scaled = NULL;       /* patterns per scale index, built on first use */
  uint8_t have[ORC_SCALES] = {0};
  if (scale_invariant) scaled = (orc_pattern*)malloc(sizeof(orc_pattern) * ORC_SCALES);
  int kept = 0;
  float xs[ORC_PATTERN_POINTS], ys[ORC_PATTERN_POINTS];
  int values[ORC_PATTERN_POINTS];
  for (int k = 0; k < n; ++k) {
    orc_keypoint kp = kps[k];
    const orc_pattern* pat = base;
    if (scale_invariant) {
      const int idx = orc_scale_index(kp.size);
      if (!have[idx]) {
        orc_pattern_scaled(base, idx, &scaled[idx]);
        have[idx] = 1;
      }
      pat = &scaled[idx];
    }
    const int border = pat->border;
    /* RoI predicate: pattern circle must fit */
    if (kp.x < (float)border || kp.x >= (float)(w - border) || kp.y < (float)border ||
        kp.y >= (float)(h - border))
      continue;
    float M[4] = {1.0f, 0.0f, 0.0f, 1.0f};
    if (mode == ORC_MODE_CAMERA_AWARE) {
      if (!camera_aware_matrix(rays_hw3, jac_hw6, w, fu, dir, kp.x, kp.y, M)) continue;
    } else if (mode == ORC_MODE_GRADIENT) {
      if (!sample_positions(pat, M, kp.x, kp.y, w, h, xs, ys)) continue;
      for (int i = 0; i < pat->n_points; ++i)
        values[i] = orc_smoothed_intensity(img, integral, w, h, stride, xs[i], ys[i],
                                           pat->sigma_half[i]);
      int direction0 = 0, direction1 = 0;
      for (int l = 0; l < pat->n_long; ++l) {
        const int delta_t = values[pat->long_i[l]] - values[pat->long_j[l]];
        direction0 += delta_t * pat->long_wdx[l] / 1024; /* truncating division */
        direction1 += delta_t * pat->long_wdy[l] / 1024;
      }
      int best_k = 0;
      if (direction0 != 0 || direction1 != 0) {
        int64_t best = INT64_MIN;
        for (int r = 0; r < 1024; ++r) {
          const int64_t dot = (int64_t)direction0 * pat->rot_cos[r] +
                              (int64_t)direction1 * pat->rot_sin[r];
          if (dot > best) {
            best = dot;
            best_k = r;
          }
        }
      }
      kp.angle = (float)best_k * 0.3515625f; /* 360/1024, exact */
      M[0] = pat->rot_cosf[best_k];
      M[1] = -pat->rot_sinf[best_k];
      M[2] = pat->rot_sinf[best_k];
      M[3] = pat->rot_cosf[best_k];
    }
    if (!sample_positions(pat, M, kp.x, kp.y, w, h, xs, ys)) continue;
    for (int i = 0; i < pat->n_points; ++i)
      values[i] =
          orc_smoothed_intensity(img, integral, w, h, stride, xs[i], ys[i], pat->sigma_half[i]);
    uint8_t* d = desc + (size_t)kept * ORC_DESC_BYTES;
    memset(d, 0, ORC_DESC_BYTES);
    for (int b = 0; b < pat->n_short; ++b)
      if (values[pat->short_i[b]] > values[pat->short_j[b]]) d[b >> 3] |= (uint8_t)(1u << (b & 7));
    kps[kept++] = kp;
  }
  free(integral);
  free(scaled);
  return kept;
}

int orc_describe(const uint8_t* img, int w, int h, int stride, const orc_pattern* pat, int mode,
                 const float* rays_hw3, const float* jac_hw6, float fu, const float dir[3],
                 orc_keypoint* kps, int n, uint8_t* desc) {
  return describe_impl(img, w, h, stride, pat, mode, rays_hw3, jac_hw6, fu, dir, kps, n, desc, 0);
}
int orc_describe_scaled(const uint8_t* img, int w, int h, int stride, const orc_pattern* pat, int mode,
                        const float* rays_hw3, const float* jac_hw6, float fu, const float dir[3],
                        orc_keypoint* kps, int n, uint8_t* desc) {
  return describe_impl(img, w, h, stride, pat, mode, rays_hw3, jac_hw6, fu, dir, kps, n, desc, 1);
}

int orc_detect_describe(const uint8_t* img, int w, int h, int stride,
                        const orc_frontend_params* prm, const orc_pattern* pat,
                        const float* rays_hw3, const float* jac_hw6, float fu, const float dir[3],
                        orc_keypoint* kps, uint8_t* desc, int cap) {
  int n = orc_detect(img, w, h, stride, prm->uniformity_radius, prm->octaves, prm->abs_threshold,
                     prm->max_kpts, kps, cap, NULL);
  return orc_describe(img, w, h, stride, pat, prm->mode, rays_hw3, jac_hw6, fu, dir, kps, n, desc);
}
