/*
 * orc_detect.c -- oracle restatement of the detector behind
 *   brisk::ScaleSpaceFeatureDetector<brisk::HarrisScoreCalculator>(
 *       uniformityRadius, octaves, absoluteThreshold, maxNumKpt)
 * constructed at okvis_frontend/src/Frontend.cpp:2406-2409 and invoked through
 * cv::FeatureDetector::detect at okvis_cv/include/okvis/implementation/Frame.hpp:152.
 *
 * TEST INFRASTRUCTURE ONLY (see okvfe_oracle.h).  PARITY UNPINNED: the brisk
 * submodule is absent; this restates the published BRISK2 Harris pipeline:
 *   Scharr (3,10,3) gradients -> products scaled to 16 bit -> 3x3 binomial
 *   (1 2 1)^2 sum -> det - trace^2/16 in int32 -> 8-neighbour non-max
 *   suppression with absolute threshold -> score-sorted uniformity enforcement
 *   on an occupancy grid (scaling 15/radius, 31x31 radial stamp, saturating
 *   adds) capped at maxNumKpt -> 2-D quadratic sub-pixel refinement ->
 *   cv::KeyPoint(pt, 12*scale, -1, score, layer).
 * octaves: every shipped config uses 0 = single layer at full resolution
 * (config/euroc.yaml:66, okvis_common/include/okvis/Parameters.hpp:127); octaves > 0 builds the
 * scale space described at detect_scale_space below.
 */
#include "okvfe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* arithmetic shift right with floor semantics, independent of the compiler */
static inline int32_t asr(int32_t v, int s) {
  return v >= 0 ? (v >> s) : -(int32_t)(((uint32_t)(-(v + 1)) >> s) + 1);
}

/* ---- K1: Harris score --------------------------------------------------------------------- */
/* Gradient products G* live on 1..h-2 x 1..w-2 (zero on the rim), the smoothed
 * entries S* and the score on the same range, each computed from the zero-
 * rimmed previous stage exactly as the separate whole-image passes of the
 * library do (GetCovarEntries -> FilterGauss3by316S x3 -> CornerHarris). */
void orc_harris_score(const uint8_t* img, int w, int h, int stride, int32_t* score) {
  size_t n = (size_t)w * (size_t)h;
  int16_t* gxx = (int16_t*)calloc(n, sizeof(int16_t));
  int16_t* gyy = (int16_t*)calloc(n, sizeof(int16_t));
  int16_t* gxy = (int16_t*)calloc(n, sizeof(int16_t));
  memset(score, 0, n * sizeof(int32_t));
  for (int y = 1; y < h - 1; ++y) {
    const uint8_t* r0 = img + (size_t)(y - 1) * stride;
    const uint8_t* r1 = img + (size_t)y * stride;
    const uint8_t* r2 = img + (size_t)(y + 1) * stride;
    for (int x = 1; x < w - 1; ++x) {
      int gx = 3 * ((int)r0[x + 1] - (int)r0[x - 1]) + 10 * ((int)r1[x + 1] - (int)r1[x - 1]) +
               3 * ((int)r2[x + 1] - (int)r2[x - 1]);
      int gy = 3 * ((int)r2[x - 1] - (int)r0[x - 1]) + 10 * ((int)r2[x] - (int)r0[x]) +
               3 * ((int)r2[x + 1] - (int)r0[x + 1]);
      /* |g| <= 4080; (8g)^2 >> 16 >> 4 == g^2 >> 14 <= 1016 */
      gxx[(size_t)y * w + x] = (int16_t)asr(gx * gx, 14);
      gyy[(size_t)y * w + x] = (int16_t)asr(gy * gy, 14);
      gxy[(size_t)y * w + x] = (int16_t)asr(gx * gy, 14);
    }
  }
  for (int y = 1; y < h - 1; ++y) {
    for (int x = 1; x < w - 1; ++x) {
      int32_t s[3];
      const int16_t* src[3] = {gxx, gyy, gxy};
      for (int c = 0; c < 3; ++c) {
        const int16_t* g = src[c];
        const int16_t* a = g + (size_t)(y - 1) * w + x;
        const int16_t* b = g + (size_t)y * w + x;
        const int16_t* d = g + (size_t)(y + 1) * w + x;
        int32_t v = a[-1] + 2 * a[0] + a[1] + 2 * b[-1] + 4 * b[0] + 2 * b[1] + d[-1] + 2 * d[0] +
                    d[1];
        s[c] = (int16_t)v; /* 16-bit container; |v| <= 16256 so no wrap occurs */
      }
      int32_t det = s[0] * s[1] - s[2] * s[2];
      int32_t tq = asr(asr(s[0], 1) + asr(s[1], 1), 1); /* trace / 4; kappa = 1/16 */
      score[(size_t)y * w + x] = det - tq * tq;
    }
  }
  free(gxx);
  free(gyy);
  free(gxy);
}

/* ---- K2: 8-neighbour non-max suppression -------------------------------------------------- */
/* ---- AGAST / FAST 9-16 corner score -----------------------------------------------------------
 * Score calculator of brisk::BriskFeatureDetector (the reference's ARM branch,
 * okvis_cv/test/TestFrame.cpp:71-72: BriskFeatureDetector(34, 2)); [NOT IN TREE] like the rest of
 * the brisk library -- this follows the PUBLISHED definition (Rosten & Drummond FAST-9 on the
 * 16-pixel Bresenham circle of radius 3; AGAST, Mair et al., decides the same predicate with a
 * faster tree): the pixel is a corner at threshold t iff 9 contiguous circle pixels are all
 * brighter than p + t or all darker than p - t (strict), and its score is the LARGEST such t --
 * what the bisection of the published cornerScore converges to --, i.e.
 *   max( max_s min_{k<9} (c[s+k] - p), max_s min_{k<9} (p - c[s+k]) ) - 1, clamped at 0.
 * Pixels closer than 3 px to the image border score 0. */
static const int kCircle16[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},  {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                     {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};
void orc_agast_score(const uint8_t* img, int w, int h, int stride, int32_t* score) {
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int s = 0;
      if (x >= 3 && y >= 3 && x < w - 3 && y < h - 3) {
        const int p = img[(size_t)y * stride + x];
        int d[16];
        for (int i = 0; i < 16; ++i)
          d[i] = (int)img[(size_t)(y + kCircle16[i][1]) * stride + (x + kCircle16[i][0])] - p;
        int bright = -256, dark = -256;
        for (int st = 0; st < 16; ++st) {
          int mn = 256, mx = -256;
          for (int k = 0; k < 9; ++k) {
            const int v = d[(st + k) & 15];
            if (v < mn) mn = v;
            if (v > mx) mx = v;
          }
          if (mn > bright) bright = mn;
          if (-mx > dark) dark = -mx;
        }
        s = (bright > dark ? bright : dark) - 1;
        if (s < 0) s = 0;
      }
      score[(size_t)y * w + x] = s;
    }
}
/* FAST 5-8 score: the same predicate on the 8-pixel ring of radius 1 with arcs of 5 -- what the
 * published BRISK detector evaluates on c0 to stand in for the (virtual) intra-octave below the
 * first octave (Leutenegger et al., ICCV 2011, section 3.1).  Border of 1 px scores 0. */
static const int kRing8[8][2] = {{0, 1}, {1, 1}, {1, 0}, {1, -1}, {0, -1}, {-1, -1}, {-1, 0}, {-1, 1}};
void orc_fast58_score(const uint8_t* img, int w, int h, int stride, int32_t* score) {
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int s = 0;
      if (x >= 1 && y >= 1 && x < w - 1 && y < h - 1) {
        const int p = img[(size_t)y * stride + x];
        int d[8];
        for (int i = 0; i < 8; ++i) d[i] = (int)img[(size_t)(y + kRing8[i][1]) * stride + (x + kRing8[i][0])] - p;
        int bright = -256, dark = -256;
        for (int st = 0; st < 8; ++st) {
          int mn = 256, mx = -256;
          for (int k = 0; k < 5; ++k) {
            const int v = d[(st + k) & 7];
            if (v < mn) mn = v;
            if (v > mx) mx = v;
          }
          if (mn > bright) bright = mn;
          if (-mx > dark) dark = -mx;
        }
        s = (bright > dark ? bright : dark) - 1;
        if (s < 0) s = 0;
      }
      score[(size_t)y * w + x] = s;
    }
}

/* Continuous scale of a scale-space maximum (published BRISK, section 3.1: "a 1D parabola is fitted
 * along the scale axis"): the parabola through (r_b, s_b), (1, s), (r_a, s_a) -- r = scale of the layer
 * below / above relative to the keypoint's layer -- evaluated in double in this fixed order; its
 * vertex, clamped to [r_b, r_a], is the relative scale, the parabola's value there the refined score.
 * A missing neighbour layer, or a parabola that does not open downwards, keeps (1, s). */
void orc_scale_refine(double rb, int have_b, int32_t sb, int32_t s, double ra, int have_a, int32_t sa,
                      double lo, float* rel_scale, float* score) {
  *rel_scale = 1.0f;
  *score = (float)s;
  if (!have_b || !have_a) return;
  const double y0 = (double)sb, y1 = (double)s, y2 = (double)sa;
  double d10 = y1 - y0;
  double h10 = 1.0 - rb;
  d10 = d10 / h10;
  double d21 = y2 - y1;
  double h21 = ra - 1.0;
  d21 = d21 / h21;
  double a = d21 - d10;
  double h20 = ra - rb;
  a = a / h20;
  if (!(a < 0.0)) return;
  double t = 1.0 + rb;
  t = a * t;
  const double b = d10 - t;
  double v = -b;
  double a2 = 2.0 * a;
  v = v / a2;
  v = v < lo ? lo : (v > ra ? ra : v); /* lo = rb except on layer 0 (published: nodes 2/3, 1, 3/2, result in [0.7, 1.5]) */
  /* value at the vertex: y1 + (v - 1) * (d10 + a * (v - rb)) (Newton form) */
  double u = v - rb;
  u = a * u;
  u = d10 + u;
  double dv = v - 1.0;
  u = dv * u;
  u = y1 + u;
  *rel_scale = (float)v;
  *score = (float)u;
}

void orc_score_map(int score_type, const uint8_t* img, int w, int h, int stride, int32_t* score) {
  if (score_type == 1 || score_type == 2)
    orc_agast_score(img, w, h, stride, score);
  else
    orc_harris_score(img, w, h, stride, score);
}

/* Raster scan of rows 2..h-3, columns 2..w-3.  A centre passes when it is >=
 * the absolute threshold and no neighbour is strictly greater; the pixel right
 * after an accepted maximum is skipped (so of two equal horizontal neighbours
 * the left one wins).  Output order: y ascending, x ascending. */
int orc_nms(const int32_t* score, int w, int h, int abs_threshold, orc_point_score* out, int cap) {
  int n = 0;
  for (int y = 2; y < h - 2; ++y) {
    int last = 0;
    for (int x = 2; x < w - 2; ++x) {
      if (last) {
        last = 0;
        continue;
      }
      const int32_t* c = score + (size_t)y * w + x;
      int32_t v = *c;
      if (v < abs_threshold) continue;
      if (c[1] > v || c[-1] > v) continue;
      const int32_t* p1 = c + w;
      const int32_t* p2 = c - w;
      if (p1[0] > v || p2[0] > v) continue;
      if (p1[1] > v || p1[-1] > v || p2[1] > v || p2[-1] > v) continue;
      if (n < cap) {
        out[n].x = x;
        out[n].y = y;
        out[n].score = v;
      }
      ++n;
      last = 1;
    }
  }
  return n;
}

/* ---- K3: uniformity enforcement ----------------------------------------------------------- */
static int cmp_points(const void* pa, const void* pb) {
  const orc_point_score* a = (const orc_point_score*)pa;
  const orc_point_score* b = (const orc_point_score*)pb;
  if (a->score != b->score) return a->score > b->score ? -1 : 1; /* score descending */
  if (a->y != b->y) return a->y < b->y ? -1 : 1;                 /* total order for ties */
  if (a->x != b->x) return a->x < b->x ? -1 : 1;
  return 0;
}

/* Greedy selection in score order.  Each candidate reads the occupancy cell at
 * its scaled position; it is dropped when sqrt(sqrt(score/maxScore))*255 is
 * below that cell, otherwise a 31x31 radial patch (1 - d^2/225, clipped at 0)
 * times 0.99 of that level is added with u8 saturation.  Stops after
 * max_kpts accepted points.  Returns the new count; pts is rewritten in
 * acceptance order.  radius <= 0 disables the stage (points stay in raster
 * order and are not capped). */
int orc_uniformity_select(orc_point_score* pts, int n, int w, int h, float radius, int max_kpts) {
  if (n <= 0) return 0;
  if (!(radius > 0.0f)) return n;
  qsort(pts, (size_t)n, sizeof(orc_point_score), cmp_points);
  float lut[31][31];
  for (int y = 0; y < 31; ++y)
    for (int x = 0; x < 31; ++x) {
      double v = 1.0 - (double)((15 - x) * (15 - x) + (15 - y) * (15 - y)) / 225.0;
      lut[y][x] = (float)(v > 0.0 ? v : 0.0);
    }
  const float max_score = (float)pts[0].score;
  const float scaling = (float)(15.0 / (double)radius);
  const int cs = (int)ceilf(scaling);
  const int orows = h * cs + 32, ocols = w * cs + 32;
  uint8_t* occ = (uint8_t*)calloc((size_t)orows * ocols, 1);
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    if (kept >= max_kpts) break;
    const orc_point_score p = pts[i];
    float fy = (float)p.y * scaling;
    float fx = (float)p.x * scaling;
    const int cy = (int)(fy + 16.0f);
    const int cx = (int)(fx + 16.0f);
    const float s0 = (float)occ[(size_t)cy * ocols + cx];
    float q = (float)p.score / max_score;
    const float nsc1 = sqrtf(sqrtf(q)) * 255.0f;
    if (nsc1 < s0) continue;
    const float nsc = (float)(0.99 * (double)nsc1);
    for (int y = 0; y < 31; ++y) {
      uint8_t* row = occ + (size_t)(cy + y - 15) * ocols + (cx - 15);
      for (int x = 0; x < 31; ++x) {
        float m = lut[y][x] * nsc;
        int add = (int)ceilf(m);
        int v = (int)row[x] + add;
        row[x] = (uint8_t)(v > 255 ? 255 : v);
      }
    }
    pts[kept++] = p;
  }
  free(occ);
  return kept;
}

/* ---- K4: 2-D quadratic sub-pixel refinement ------------------------------------------------ */
/* s = 3x3 score patch, row-major: s[0]=(-1,-1) s[1]=(0,-1) s[2]=(+1,-1) s[3]=(-1,0) ...
 * Least-squares quadratic fit of the published BRISK refinement; coefficients
 * in 64-bit integers (Harris scores overflow 32-bit products), the Hessian
 * determinant and the numerators in double, divisions in float. */
void orc_subpixel2d(const int32_t s[9], float* delta_x, float* delta_y) {
  /* s_i_j of the published formula = score(x-1+i, y-1+j): first index along x */
  const int64_t s00 = s[0], s01 = s[3], s02 = s[6];
  const int64_t s10 = s[1], s11 = s[4], s12 = s[7];
  const int64_t s20 = s[2], s21 = s[5], s22 = s[8];
  const int64_t tmp1 = s00 + s02 - 2 * s11 + s20 + s22;
  const int64_t c1 = 3 * (tmp1 + s01 - ((s10 + s12) * 2) + s21);
  const int64_t c2 = 3 * (tmp1 - ((s01 + s21) * 2) + s10 + s12);
  const int64_t tmp2 = s02 - s20;
  const int64_t tmp3 = s00 + tmp2 - s22;
  const int64_t tmp4 = tmp3 - 2 * tmp2;
  const int64_t c3 = -3 * (tmp3 + s01 - s21);
  const int64_t c4 = -3 * (tmp4 + s10 - s12);
  const int64_t c5 = (s00 - s02 - s20 + s22) * 4;
  const int64_t c6 = -(s00 + s02 - ((s10 + s01 + s12 + s21) * 2) - 5 * s11 + s20 + s22) * 2;
  /* |c| < 2^37, so the products need more than 64 bits: they are formed in
   * IEEE double (each product and difference correctly rounded, no FMA). */
  const double d1 = (double)c1, d2 = (double)c2, d3 = (double)c3, d4 = (double)c4, d5 = (double)c5;
  double ha = 4.0 * d1; ha = ha * d2;
  double hb = d5 * d5;
  const double hdet = ha - hb;
  if (hdet == 0.0) {
    *delta_x = 0.0f;
    *delta_y = 0.0f;
    return;
  }
  if (!(hdet > 0.0 && c1 < 0)) {
    /* maximum on one of the four patch corners */
    int64_t best = c3 + c4 + c5;
    float bx = 1.0f, by = 1.0f;
    int64_t t = -c3 + c4 - c5;
    if (t > best) { best = t; bx = -1.0f; by = 1.0f; }
    t = c3 - c4 - c5;
    if (t > best) { best = t; bx = 1.0f; by = -1.0f; }
    t = -c3 - c4 + c5;
    if (t > best) { best = t; bx = -1.0f; by = -1.0f; }
    *delta_x = bx;
    *delta_y = by;
    return;
  }
  const float fh = -(float)hdet;
  double na = 2.0 * d2; na = na * d3;
  double nb = d4 * d5;
  const float nx = (float)(na - nb);
  na = 2.0 * d1; na = na * d4;
  nb = d3 * d5;
  const float ny = (float)(na - nb);
  float dx = nx / fh;
  float dy = ny / fh;
  const int tx = dx > 1.0f, tx_ = dx < -1.0f, ty = dy > 1.0f, ty_ = dy < -1.0f;
  if (tx || tx_ || ty || ty_) {
    const float f1 = (float)c1, f2 = (float)c2, f3 = (float)c3, f4 = (float)c4, f5 = (float)c5,
                f6 = (float)c6;
    float dx1 = 0.0f, dx2 = 0.0f, dy1 = 0.0f, dy2 = 0.0f;
    if (tx) {
      dx1 = 1.0f;
      dy1 = -(f4 + f5) / (2.0f * f2);
      if (dy1 > 1.0f) dy1 = 1.0f; else if (dy1 < -1.0f) dy1 = -1.0f;
    } else if (tx_) {
      dx1 = -1.0f;
      dy1 = -(f4 - f5) / (2.0f * f2);
      if (dy1 > 1.0f) dy1 = 1.0f; else if (dy1 < -1.0f) dy1 = -1.0f;
    }
    if (ty) {
      dy2 = 1.0f;
      dx2 = -(f3 + f5) / (2.0f * f1);
      if (dx2 > 1.0f) dx2 = 1.0f; else if (dx2 < -1.0f) dx2 = -1.0f;
    } else if (ty_) {
      dy2 = -1.0f;
      dx2 = -(f3 - f5) / (2.0f * f1);
      if (dx2 > 1.0f) dx2 = 1.0f; else if (dx2 < -1.0f) dx2 = -1.0f;
    }
    /* evaluate both options; explicit temporaries fix the summation order */
    float m1 = f1 * dx1; m1 = m1 * dx1;
    float a = f2 * dy1; a = a * dy1; m1 = m1 + a;
    a = f3 * dx1; m1 = m1 + a;
    a = f4 * dy1; m1 = m1 + a;
    a = f5 * dx1; a = a * dy1; m1 = m1 + a;
    m1 = m1 + f6;
    float m2 = f1 * dx2; m2 = m2 * dx2;
    a = f2 * dy2; a = a * dy2; m2 = m2 + a;
    a = f3 * dx2; m2 = m2 + a;
    a = f4 * dy2; m2 = m2 + a;
    a = f5 * dx2; a = a * dy2; m2 = m2 + a;
    m2 = m2 + f6;
    if (m1 > m2) { dx = dx1; dy = dy1; } else { dx = dx2; dy = dy2; }
  }
  *delta_x = dx;
  *delta_y = dy;
}

/* ---- scale space (octaves > 0) -------------------------------------------------------------- */
/* PARITY UNPINNED like the rest of this file.  The reference passes `octaves` to
 * brisk::ScaleSpaceFeatureDetector (Frontend.cpp:2406-2409; its own smoke test uses 2,
 * okvis_cv/test/TestFrame.cpp:75-77); "0 means single-scale at highest resolution"
 * (Parameters.hpp:127).  Restated from the published BRISK scale space:
 *   layers    2*octaves of them.  Layer 0 = the image (scale 1), layer 1 = two-third sampling of
 *             layer 0 (scale 1.5), layer l >= 2 = half sampling of layer l-2 (scales 2, 3, 4, 6...).
 *   half      2x2 box mean, (a+b+c+d+2)>>2;  size w/2 x h/2.
 *   2/3       every 3x3 source block gives 2x2 pixels with the separable weights (2,1,0)/3 and
 *             (0,1,2)/3: (sum w_x w_y s + 4) / 9;  size (w/3)*2 x (h/3)*2.
 *   per layer the single-scale pipeline: Harris score, 8-neighbour NMS with the absolute threshold.
 *   scale-space maximum: a 2-D maximum of layer l survives unless a STRICTLY greater score exists
 *             in layer l-1 or l+1 within +-1 pixel (that layer's pixels) of the same image
 *             location (pixel centres: X = s (x + 1/2) - 1/2).
 *   then per layer: uniformity enforcement (radius in layer pixels, maxNumKpt per layer), sub-pixel
 *             refinement, cv::KeyPoint(pt in image coordinates, 12 * scale, -1, score, layer).
 * Output: layers in ascending order, at most 2*octaves*max_kpts keypoints. */
void orc_halfsample(const uint8_t* src, int w, int h, int stride, uint8_t* dst /* (h/2)*(w/2) */) {
  const int w2 = w / 2, h2 = h / 2;
  for (int y = 0; y < h2; ++y) {
    const uint8_t* r0 = src + (size_t)(2 * y) * stride;
    const uint8_t* r1 = r0 + stride;
    for (int x = 0; x < w2; ++x)
      dst[(size_t)y * w2 + x] = (uint8_t)((r0[2 * x] + r0[2 * x + 1] + r1[2 * x] + r1[2 * x + 1] + 2) >> 2);
  }
}

void orc_twothirdsample(const uint8_t* src, int w, int h, int stride,
                        uint8_t* dst /* ((h/3)*2)*((w/3)*2) */) {
  const int bw = w / 3, bh = h / 3, w2 = bw * 2;
  static const int wt[2][3] = {{2, 1, 0}, {0, 1, 2}};
  for (int by = 0; by < bh; ++by)
    for (int q = 0; q < 2; ++q)
      for (int bx = 0; bx < bw; ++bx)
        for (int p = 0; p < 2; ++p) {
          int acc = 0;
          for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i)
              acc += wt[q][j] * wt[p][i] * (int)src[(size_t)(3 * by + j) * stride + 3 * bx + i];
          dst[(size_t)(2 * by + q) * w2 + 2 * bx + p] = (uint8_t)((acc + 4) / 9);
        }
}

/* scale of layer l as a fraction: even l: 2^(l/2); odd l: 3 * 2^((l-1)/2) / 2 */
void orc_layer_scale(int l, int* num, int* den) {
  if ((l & 1) == 0) {
    *num = 1 << (l / 2);
    *den = 1;
  } else {
    *num = 3 << ((l - 1) / 2);
    *den = 2;
  }
}
void orc_layer_size(int w, int h, int l, int* lw, int* lh) {
  if (l == 0) {
    *lw = w; *lh = h;
  } else if (l == 1) {
    *lw = (w / 3) * 2; *lh = (h / 3) * 2;
  } else {
    int pw, ph;
    orc_layer_size(w, h, l - 2, &pw, &ph);
    *lw = pw / 2; *lh = ph / 2;
  }
}
static int floor_div(int a, int b) { /* b > 0 */
  return a >= 0 ? a / b : -((-a + b - 1) / b);
}
/* 1 when no pixel of `other` (wo x ho) within +-1 px of the location corresponding to (x, y) of a
 * layer whose scale is rn/rd times the other layer's has a strictly greater score than s */
int orc_scale_neighbour_ok(const int32_t* other, int wo, int ho, int x, int y, int32_t s, int rn, int rd) {
  /* x' = ((2x+1) rn/rd - 1) / 2 = N / D,  D = 2 rd */
  const int D = 2 * rd;
  const int Nx = (2 * x + 1) * rn - rd, Ny = (2 * y + 1) * rn - rd;
  int u0 = -floor_div(-(Nx - D), D), u1 = floor_div(Nx + D, D);  /* ceil((N-D)/D) .. floor((N+D)/D) */
  int v0 = -floor_div(-(Ny - D), D), v1 = floor_div(Ny + D, D);
  if (u0 < 0) u0 = 0;
  if (v0 < 0) v0 = 0;
  if (u1 > wo - 1) u1 = wo - 1;
  if (v1 > ho - 1) v1 = ho - 1;
  for (int v = v0; v <= v1; ++v)
    for (int u = u0; u <= u1; ++u)
      if (other[(size_t)v * wo + u] > s) return 0;
  return 1;
}

/* largest score of `other` within the same +-1 px window (the value the scale parabola takes for
 * the neighbouring layer) */
int32_t orc_scale_neighbour_max(const int32_t* other, int wo, int ho, int x, int y, int rn, int rd) {
  const int D = 2 * rd;
  const int Nx = (2 * x + 1) * rn - rd, Ny = (2 * y + 1) * rn - rd;
  int u0 = -floor_div(-(Nx - D), D), u1 = floor_div(Nx + D, D);
  int v0 = -floor_div(-(Ny - D), D), v1 = floor_div(Ny + D, D);
  if (u0 < 0) u0 = 0;
  if (v0 < 0) v0 = 0;
  if (u1 > wo - 1) u1 = wo - 1;
  if (v1 > ho - 1) v1 = ho - 1;
  int32_t m = 0;
  for (int v = v0; v <= v1; ++v)
    for (int u = u0; u <= u1; ++u)
      if (other[(size_t)v * wo + u] > m) m = other[(size_t)v * wo + u];
  return m;
}

#define ORC_MAX_LAYERS 8
static int detect_scale_space(const uint8_t* img, int w, int h, int stride, float uniformity_radius,
                              int octaves, int abs_threshold, int max_kpts, orc_keypoint* kps, int cap,
                              int score_type) {
  const int L = 2 * octaves;
  if (L > ORC_MAX_LAYERS) return -1;
  uint8_t* im[ORC_MAX_LAYERS];
  int32_t* sc[ORC_MAX_LAYERS];
  orc_point_score* pts[ORC_MAX_LAYERS];
  int np[ORC_MAX_LAYERS], lw[ORC_MAX_LAYERS], lh[ORC_MAX_LAYERS], st[ORC_MAX_LAYERS];
  for (int l = 0; l < L; ++l) {
    orc_layer_size(w, h, l, &lw[l], &lh[l]);
    if (lw[l] < 8 || lh[l] < 8) return -1;
    if (l == 0) {
      im[0] = (uint8_t*)img;
      st[0] = stride;
    } else {
      im[l] = (uint8_t*)malloc((size_t)lw[l] * lh[l]);
      st[l] = lw[l];
      if (l == 1)
        orc_twothirdsample(im[0], lw[0], lh[0], st[0], im[1]);
      else
        orc_halfsample(im[l - 2], lw[l - 2], lh[l - 2], st[l - 2], im[l]);
    }
    sc[l] = (int32_t*)malloc((size_t)lw[l] * lh[l] * sizeof(int32_t));
    orc_score_map(score_type, im[l], lw[l], lh[l], st[l], sc[l]);
    int maxc = (lw[l] / 2 + 1) * (lh[l] - 3);
    if (maxc < 16) maxc = 16;
    pts[l] = (orc_point_score*)malloc((size_t)maxc * sizeof(orc_point_score));
    np[l] = orc_nms(sc[l], lw[l], lh[l], abs_threshold, pts[l], maxc);
  }
  /* score_type 2 = the published BRISK scale-space detector (brisk::BriskFeatureDetector(threshold,
   * octaves), okvis_cv/test/TestFrame.cpp:71-72): AGAST 9-16 scores on every layer, the FAST 5-8 score
   * of c0 as the virtual layer below it, 2-D maxima that are also maxima against the +-1 px patches of
   * the layers below and above, strongest first up to max_kpts per layer (no uniformity), 2-D sub-pixel
   * fit in the layer, and a 1-D parabola over the three layers' scores for the continuous scale
   * (size = 12 * layer scale * relative scale; response = the parabola's value). */
  int32_t* sc_virtual = NULL;
  if (score_type == 2) {
    sc_virtual = (int32_t*)malloc((size_t)lw[0] * lh[0] * sizeof(int32_t));
    orc_fast58_score(im[0], lw[0], lh[0], st[0], sc_virtual);
  }
  int nout = 0;
  for (int l = 0; l < L; ++l) {
    int sn, sd;
    orc_layer_scale(l, &sn, &sd);
    /* scale-space maxima: compare with the layers below and above */
    int kept = 0;
    for (int i = 0; i < np[l]; ++i) {
      const orc_point_score p = pts[l][i];
      int ok = 1;
      if (score_type == 2 && l == 0) ok = orc_scale_neighbour_ok(sc_virtual, lw[0], lh[0], p.x, p.y, p.score, 1, 1);
      for (int dl = -1; dl <= 1 && ok; dl += 2) {
        const int m = l + dl;
        if (m < 0 || m >= L) continue;
        int mn, md;
        orc_layer_scale(m, &mn, &md);
        /* ratio scale_l / scale_m = (sn/sd) / (mn/md), reduced */
        int rn = sn * md, rd = sd * mn;
        for (int g = 2; g <= 3; ++g)
          while (rn % g == 0 && rd % g == 0) { rn /= g; rd /= g; }
        ok = orc_scale_neighbour_ok(sc[m], lw[m], lh[m], p.x, p.y, p.score, rn, rd);
      }
      if (ok) pts[l][kept++] = p;
    }
    if (score_type == 2) { /* strongest first: (score desc, y, x) -- the total order of the uniformity stage */
      qsort(pts[l], (size_t)kept, sizeof(orc_point_score), cmp_points);
      if (kept > max_kpts) kept = max_kpts;
    } else {
      kept = orc_uniformity_select(pts[l], kept, lw[l], lh[l], uniformity_radius, max_kpts);
    }
    const float scale = (float)sn / (float)sd;
    for (int i = 0; i < kept && nout < cap; ++i) {
      const int u = pts[l][i].x, v = pts[l][i].y;
      int32_t patch[9];
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx)
          patch[(dy + 1) * 3 + (dx + 1)] = sc[l][(size_t)(v + dy) * lw[l] + (u + dx)];
      float ddx, ddy;
      orc_subpixel2d(patch, &ddx, &ddy);
      const float xl = (float)u + ddx, yl = (float)v + ddy; /* layer coordinates */
      orc_keypoint* k = &kps[nout++];
      float t = xl + 0.5f;
      t = scale * t;
      k->x = t - 0.5f;
      t = yl + 0.5f;
      t = scale * t;
      k->y = t - 0.5f;
      k->size = 12.0f * scale;
      k->angle = -1.0f;
      k->response = (float)pts[l][i].score;
      k->octave = l;
      k->class_id = -1;
      if (score_type == 2) { /* continuous scale from the three layers' scores */
        double rb = 0.75, ra = 1.5; /* octave c_i: d_(i-1) below (3/4), d_i above (3/2) */
        if (l & 1) { rb = 2.0 / 3.0; ra = 4.0 / 3.0; } /* intra-octave d_i: c_i below, c_(i+1) above */
        double lo = rb;
        if (l == 0) { rb = 2.0 / 3.0; lo = 0.7; } /* c_0: the virtual FAST 5-8 layer sits at 2/3 (published refine1D_2:
                                                   * coefficients 18 -30 12 = nodes 2/3, 1, 3/2; result clamped to [0.7, 1.5]) */
        int have_b = 1, have_a = l + 1 < L;
        int32_t sb = 0, sa = 0;
        if (l == 0) {
          sb = orc_scale_neighbour_max(sc_virtual, lw[0], lh[0], u, v, 1, 1);
        } else {
          int mn, md;
          orc_layer_scale(l - 1, &mn, &md);
          int rn = sn * md, rd = sd * mn;
          for (int g = 2; g <= 3; ++g)
            while (rn % g == 0 && rd % g == 0) { rn /= g; rd /= g; }
          sb = orc_scale_neighbour_max(sc[l - 1], lw[l - 1], lh[l - 1], u, v, rn, rd);
        }
        if (have_a) {
          int mn, md;
          orc_layer_scale(l + 1, &mn, &md);
          int rn = sn * md, rd = sd * mn;
          for (int g = 2; g <= 3; ++g)
            while (rn % g == 0 && rd % g == 0) { rn /= g; rd /= g; }
          sa = orc_scale_neighbour_max(sc[l + 1], lw[l + 1], lh[l + 1], u, v, rn, rd);
        }
        float rel, resp;
        orc_scale_refine(rb, have_b, sb, pts[l][i].score, ra, have_a, sa, lo, &rel, &resp);
        float sz = 12.0f * rel; /* basic size 12 at the layer's own scale, times the relative scale ... */
        k->size = sz * scale;   /* ... times the layer scale */
        k->response = resp;
      }
    }
  }
  free(sc_virtual);
  for (int l = 0; l < L; ++l) {
    if (l) free(im[l]);
    free(sc[l]);
    free(pts[l]);
  }
  return nout;
}

/* ---- detect(): the whole detector for one image -------------------------------------------- */
int orc_detect(const uint8_t* img, int w, int h, int stride, float uniformity_radius, int octaves,
               int abs_threshold, int max_kpts, orc_keypoint* kps, int cap, int32_t* score_out) {
  return orc_detect_scored(img, w, h, stride, uniformity_radius, octaves, abs_threshold, max_kpts, kps, cap,
                           score_out, 0);
}

/* score_type: 0 = Harris (the x86 reference path), 1 = AGAST 9-16 score (orc_agast_score); everything
 * after the score map -- NMS, scale-space maxima, uniformity, cap, sub-pixel -- is shared. */
int orc_detect_scored(const uint8_t* img, int w, int h, int stride, float uniformity_radius, int octaves,
                      int abs_threshold, int max_kpts, orc_keypoint* kps, int cap, int32_t* score_out,
                      int score_type) {
  if (octaves > 0) {
    const int n_ss = detect_scale_space(img, w, h, stride, uniformity_radius, octaves, abs_threshold,
                                        max_kpts, kps, cap, score_type);
    if (score_out) orc_score_map(score_type, img, w, h, stride, score_out);
    return n_ss;
  }
  size_t n = (size_t)w * (size_t)h;
  int32_t* score = score_out ? score_out : (int32_t*)malloc(n * sizeof(int32_t));
  orc_score_map(score_type, img, w, h, stride, score);
  int maxc = (w / 2 + 1) * (h - 3);
  if (maxc < 16) maxc = 16;
  orc_point_score* pts = (orc_point_score*)malloc((size_t)maxc * sizeof(orc_point_score));
  int np = orc_nms(score, w, h, abs_threshold, pts, maxc);
  np = orc_uniformity_select(pts, np, w, h, uniformity_radius, max_kpts);
  int nout = 0;
  for (int i = 0; i < np && nout < cap; ++i) {
    const int u = pts[i].x, v = pts[i].y;
    int32_t patch[9];
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx)
        patch[(dy + 1) * 3 + (dx + 1)] = score[(size_t)(v + dy) * w + (u + dx)];
    float ddx, ddy;
    orc_subpixel2d(patch, &ddx, &ddy);
    orc_keypoint* k = &kps[nout++];
    k->x = (float)u + ddx;
    k->y = (float)v + ddy;
    k->size = 12.0f;
    k->angle = -1.0f;
    k->response = (float)pts[i].score;
    k->octave = 0;
    k->class_id = -1;
  }
  free(pts);
  if (!score_out) free(score);
  return nout;
}
