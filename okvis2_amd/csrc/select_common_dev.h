// select_common_dev.h -- device helpers shared by the selection / sort / refinement kernels
// (k_sort.hip, k_select_grid.hip, k_select.hip, k_brisk_refine.hip): the 64-bit ordering key, the nine
// Harris scores around a kept point recomputed from the image, the published 2-D sub-pixel fit.
#pragma once
#include "okvfe_internal.h"

namespace okvfe {
namespace {

constexpr int kThreads = 1024;
constexpr int kLdsSortKeys = 8192;
constexpr int kMaxKp = 4096;  // okvfe_create enforces max_keypoints <= 4096

__device__ __forceinline__ uint64_t make_key(const Candidate& c) {
  return ((uint64_t)(uint32_t)(0x7FFFFFFF - c.score) << 32) | ((uint32_t)c.y << 16) |
         (uint32_t)c.x;
}

// Launched twice when the candidate capacity exceeds kLdsSortKeys: first with 64 KiB of LDS for the
// images whose (padded) candidate count fits 8192 keys -- two workgroups per CU --, then with
// 128 KiB for the few that need up to 16384 keys; lds_lo_keys / lds_keys bound the range a launch
// handles, every other image is left to the other launch.  Above 16384 keys the network runs in
// the HBM workspace.

// Map-free calls (round 4): the score kernel writes no map -- four of its five bytes per pixel -- and the
// selection recomputes these nine values for the ~230 keypoints per image it keeps (7 x 7 pixels, ~1 k
// integer operations each; +18 us on the selection of 1536 EuRoC images against -105 us on the score kernel.
// A kernel of its own for this -- one thread per keypoint of the batch -- was measured and is slower (111 us:
// it is all scattered line fetches, which hide behind other images' arithmetic in here).
// xx | yy << 16 share a register (both < 2^14 after the binomial).
__device__ __forceinline__ void harris_scores_3x3(const uint8_t* __restrict__ im, int w, int h, int u, int v,
                                                  int32_t out[9]) {
  // (requires w % 4 == 0 and a dword-aligned image, like the fused score kernel this stands in for)
  // Pixels u - 3 .. u + 3 of rows v - 3 .. v + 3 as three aligned dwords per row (one keypoint per lane: every
  // load instruction of the wave touches 64 different lines, so the count of loads is what this costs); the
  // dword before the row / past its end is not read -- the pixel it would supply only feeds gradient products
  // on the image rim, which are zero by definition.
  const int base = (u - 3) & ~3;  // -4 for u = 2
  const int sh = (u - 3) - base;  // 0..3
  uint32_t q0[7], q1[7];          // pixels 0..3 and 4..6 of each row
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const int y = v - 3 + r;
    const uint32_t* row = reinterpret_cast<const uint32_t*>(im + (size_t)(y < 0 ? 0 : (y > h - 1 ? h - 1 : y)) * w);
    const uint32_t d0 = base >= 0 ? row[base >> 2] : 0u;
    const uint32_t d1 = row[(base >> 2) + 1];
    const uint32_t d2 = base + 8 < w ? row[(base >> 2) + 2] : 0u;
    q0[r] = __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)sh);
    q1[r] = __builtin_amdgcn_alignbyte(d2, d1, (uint32_t)sh);
  }
  auto px = [&](int r, int k) { return (int)(((k < 4 ? q0[r] : q1[r]) >> (8 * (k & 3))) & 0xFFu); };
  int acc_p[9], acc_xy[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc_p[k] = acc_xy[k] = 0;
  // The Scharr pair is separable: a = 3 p(y-1) + 10 p(y) + 3 p(y+1), b = p(y+1) - p(y-1) per pixel column,
  // then gx = a(x+1) - a(x-1), gy = 3 b(x-1) + 10 b(x) + 3 b(x+1).
#pragma unroll
  for (int r = -2; r <= 2; ++r) {  // gradient row v + r: pixel rows r + 2, r + 3, r + 4 of the window
    const bool yin = v + r >= 1 && v + r <= h - 2;
    int a[7], b[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const int t0 = px(r + 2, k), t1 = px(r + 3, k), t2 = px(r + 4, k);
      a[k] = 3 * (t0 + t2) + 10 * t1;
      b[k] = t2 - t0;
    }
    int gp[5], gxy[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int x = u - 2 + k;
      const int gx = a[k + 2] - a[k];
      const int gy = 3 * (b[k] + b[k + 2]) + 10 * b[k + 1];
      const bool in = yin && x >= 1 && x <= w - 2;  // gradient products are zero on the image rim
      gp[k] = in ? ((gx * gx) >> 14) | (((gy * gy) >> 14) << 16) : 0;
      gxy[k] = in ? (gx * gy) >> 14 : 0;  // arithmetic shift: floor, like the reference's 16-bit products
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int hp = gp[c] + 2 * gp[c + 1] + gp[c + 2];
      const int hxy = gxy[c] + 2 * gxy[c + 1] + gxy[c + 2];
#pragma unroll
      for (int j = -1; j <= 1; ++j) {
        const int d = j - r;
        if (d >= -1 && d <= 1) {
          acc_p[(j + 1) * 3 + c] += d == 0 ? 2 * hp : hp;
          acc_xy[(j + 1) * 3 + c] += d == 0 ? 2 * hxy : hxy;
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int s0 = acc_p[k] & 0xFFFF, s1 = (int)((unsigned)acc_p[k] >> 16), s2 = acc_xy[k];
    const int tq = ((s0 >> 1) + (s1 >> 1)) >> 1;
    out[k] = s0 * s1 - s2 * s2 - tq * tq;
  }
}

__device__ void subpixel2d(const int32_t s[9], float* delta_x, float* delta_y) {
  // s_i_j of the published formula = score(x-1+i, y-1+j): first index along x
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
  const double d1 = (double)c1, d2 = (double)c2, d3 = (double)c3, d4 = (double)c4, d5 = (double)c5;
  double ha = 4.0 * d1;
  ha = ha * d2;
  double hb = d5 * d5;
  const double hdet = ha - hb;
  if (hdet == 0.0) {
    *delta_x = 0.0f;
    *delta_y = 0.0f;
    return;
  }
  if (!(hdet > 0.0 && c1 < 0)) {
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
  double na = 2.0 * d2;
  na = na * d3;
  double nb = d4 * d5;
  const float nx = (float)(na - nb);
  na = 2.0 * d1;
  na = na * d4;
  nb = d3 * d5;
  const float ny = (float)(na - nb);
  float dx = nx / fh;
  float dy = ny / fh;
  const bool tx = dx > 1.0f, tx_ = dx < -1.0f, ty = dy > 1.0f, ty_ = dy < -1.0f;
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

}  // namespace
}  // namespace okvfe
