// describe_setup_dev.h -- the per-keypoint preparation of the extractor (border test, scale index,
// camera-aware matrix M), shared by describe_setup_kernel (k_describe.hip) and the tail of
// select_lazy_kernel (k_select.hip), which runs it for the keypoints it has just emitted when
// detection and description are one call (one launch and ~15 us per batch less).
#pragma once
#include "okvfe_internal.h"

namespace okvfe {

// M = J * [e_x e_y] / fu on the tangent plane of the keypoint's ray, e_y along `dir`
__device__ __forceinline__ bool camera_aware_matrix(const float* __restrict__ rays,
                                                    const float* __restrict__ jac, int w, float fu,
                                                    const float dir[3], float kx, float ky,
                                                    float M[4]) {
  const int u = (int)(kx + 0.5f), v = (int)(ky + 0.5f);
  const float* r = rays + ((size_t)v * w + u) * 3;
  const float* J = jac + ((size_t)v * w + u) * 6;
  const float r0 = r[0], r1 = r[1], r2 = r[2];
  if (r0 == 0.0f && r1 == 0.0f && r2 == 0.0f) return false;
  float ey0 = 0.f, ey1 = 0.f, ey2 = 0.f, n2 = 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float g0 = c == 0 ? dir[0] : (c == 1 ? 0.0f : 1.0f);
    const float g1 = c == 0 ? dir[1] : (c == 1 ? 1.0f : 0.0f);
    const float g2 = c == 0 ? dir[2] : 0.0f;
    if (c > 0 && n2 >= 1.0e-12f) break;
    float gr = g0 * r0;
    float t = g1 * r1;
    gr = gr + t;
    t = g2 * r2;
    gr = gr + t;
    t = gr * r0; ey0 = g0 - t;
    t = gr * r1; ey1 = g1 - t;
    t = gr * r2; ey2 = g2 - t;
    n2 = ey0 * ey0;
    t = ey1 * ey1;
    n2 = n2 + t;
    t = ey2 * ey2;
    n2 = n2 + t;
  }
  if (!(n2 >= 1.0e-12f)) return false;
  const float n = sqrtf(n2);
  ey0 = ey0 / n;
  ey1 = ey1 / n;
  ey2 = ey2 / n;
  float t1, t2;
  t1 = ey1 * r2; t2 = ey2 * r1; const float ex0 = t1 - t2;
  t1 = ey2 * r0; t2 = ey0 * r2; const float ex1 = t1 - t2;
  t1 = ey0 * r1; t2 = ey1 * r0; const float ex2 = t1 - t2;
  float s;
  s = J[0] * ex0; t1 = J[1] * ex1; s = s + t1; t1 = J[2] * ex2; s = s + t1; M[0] = s / fu;
  s = J[0] * ey0; t1 = J[1] * ey1; s = s + t1; t1 = J[2] * ey2; s = s + t1; M[1] = s / fu;
  s = J[3] * ex0; t1 = J[4] * ex1; s = s + t1; t1 = J[5] * ex2; s = s + t1; M[2] = s / fu;
  s = J[3] * ey0; t1 = J[4] * ey1; s = s + t1; t1 = J[5] * ey2; s = s + t1; M[3] = s / fu;
  return true;
}

// Extra sample e (a pattern point beyond the 64 lanes of describe_aware_kernel) of one keypoint, by ONE THREAD and
// straight from the image: the published BRISK smoothedIntensity with sub-pixel rim weights (same float / integer
// sequence as box_mean, k_describe_aware.hip), boxes of at most MAXB2 + 1 pixels a side.  All rows are loaded up
// front (aligned dword windows, one memory round trip); false = the box leaves the image (the keypoint is dropped).
template <int MAXB2>
__device__ __forceinline__ bool extra_sample_value(const uint8_t* __restrict__ im, int w, int h, float xf, float yf,
                                                   float sigma_half, int scaling, int scaling2, int* value) {
  const float x_1 = xf - sigma_half, x1 = xf + sigma_half;
  const float y_1 = yf - sigma_half, y1 = yf + sigma_half;
  if (!(x_1 >= 0.0f && y_1 >= 0.0f && x1 < (float)(w - 1) && y1 < (float)(h - 1))) return false;
  const int x_left = (int)(x_1 + 0.5f), y_top = (int)(y_1 + 0.5f);
  const int x_right = (int)(x1 + 0.5f), y_bottom = (int)(y1 + 0.5f);
  float r_x_1 = (float)x_left - x_1;  r_x_1 = r_x_1 + 0.5f;
  float r_y_1 = (float)y_top - y_1;   r_y_1 = r_y_1 + 0.5f;
  float r_x1 = x1 - (float)x_right;   r_x1 = r_x1 + 0.5f;
  float r_y1 = y1 - (float)y_bottom;  r_y1 = r_y1 + 0.5f;
  const float fs = (float)scaling;
  float t;
  t = r_x_1 * r_y_1; const int A = (int)(t * fs);
  t = r_x1 * r_y_1;  const int B = (int)(t * fs);
  t = r_x1 * r_y1;   const int C = (int)(t * fs);
  t = r_x_1 * r_y1;  const int D = (int)(t * fs);
  const int r_x_1_i = (int)(r_x_1 * fs), r_y_1_i = (int)(r_y_1 * fs);
  const int r_x1_i = (int)(r_x1 * fs), r_y1_i = (int)(r_y1 * fs);
  const int bw = x_right - x_left, bh = y_bottom - y_top;
  if (bw > MAXB2 || bh > MAXB2) return false;  // (cannot happen: the host checked the pattern's half-widths)
  // row window: NDW dwords from the dword that holds x_left (byte b of it); x_right is byte b + bw <= 3 + MAXB2
  constexpr int NDW = (3 + MAXB2) / 4 + 1;
  const int b = x_left & 3;
  const uint8_t* p0 = im + (size_t)y_top * w + (x_left & ~3);
  auto byte_at = [&](const uint32_t (&r)[NDW], int pos) -> int {  // byte `pos` of a row window
    uint32_t v = r[0];
#pragma unroll
    for (int j = 1; j < NDW; ++j) v = (pos >> 2) == j ? r[j] : v;
    return (int)((v >> (8 * (pos & 3))) & 0xFFu);
  };
  uint32_t msk[NDW];  // bytes b + 1 .. b + bw - 1 of a row window
#pragma unroll
  for (int j = 0; j < NDW; ++j) {
    int lo = b + 1 - 4 * j, hi = b + bw - 4 * j;
    lo = lo < 0 ? 0 : (lo > 4 ? 4 : lo);
    hi = hi < 0 ? 0 : (hi > 4 ? 4 : hi);
    const uint32_t mlo = lo >= 4 ? 0xFFFFFFFFu : ((1u << (8 * lo)) - 1u);
    const uint32_t mhi = hi >= 4 ? 0xFFFFFFFFu : ((1u << (8 * hi)) - 1u);
    msk[j] = mhi & ~mlo;
  }
  int upper = 0, middle = 0, left = 0, right = 0, bottom = 0, pl_t = 0, pr_t = 0, pl_b = 0, pr_b = 0;
  // five rows per memory round trip (registers: the selection kernel's tail has 80)
  constexpr int kChunk = 5;
#pragma unroll
  for (int c0 = 0; c0 <= MAXB2; c0 += kChunk) {
    uint32_t d[kChunk][NDW];
#pragma unroll
    for (int r = 0; r < kChunk; ++r) {
      const int dy = c0 + r;
      if (dy <= MAXB2) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(p0 + (size_t)(dy < bh ? dy : bh) * w);
#pragma unroll
        for (int j = 0; j < NDW; ++j) d[r][j] = q[j];
      }
    }
#pragma unroll
    for (int r = 0; r < kChunk; ++r) {
      const int dy = c0 + r;
      if (dy <= MAXB2) {
        const int pl = byte_at(d[r], b), pr = byte_at(d[r], b + bw);
        int in = 0;
#pragma unroll
        for (int j = 0; j < NDW; ++j) in += (int)__builtin_amdgcn_sad_u8(d[r][j] & msk[j], 0u, 0u);
        if (dy == 0) {
          pl_t = pl;
          pr_t = pr;
          upper = in;
        } else if (dy < bh) {
          left += pl;
          right += pr;
          middle += in;
        } else if (dy == bh) {
          pl_b = pl;
          pr_b = pr;
          bottom = in;
        }
      }
    }
  }
  int ret = A * pl_t;
  ret += B * pr_t;
  ret += C * pr_b;
  ret += D * pl_b;
  ret += upper * r_y_1_i + middle * scaling + left * r_x_1_i + right * r_x1_i + bottom * r_y1_i;
  *value = (ret + scaling2 / 2) / scaling2;  // (non-negative: floor)
  return true;
}

// extra sample e of the keypoint at (kx, ky) under M -> bytes 24 + 4 e of its descriptor slot; false: box outside the image
__device__ __forceinline__ bool extra_sample_one(const Pattern* __restrict__ pat, const uint8_t* __restrict__ im, int w,
                                                 int h, int extra_box, int e, const float M[4], float kx, float ky,
                                                 uint8_t* __restrict__ slot_bytes) {
  const float px = pat->px[e], py = pat->py[e], sg = pat->sigma_half[e];
  float a = M[0] * px, b2 = M[1] * py;  // (the sequence of sample_pos, k_describe_aware.hip)
  a = a + b2;
  const float xf = kx + a;
  float c = M[2] * px, d2 = M[3] * py;
  c = c + d2;
  const float yf = ky + c;
  int v = 0;
  const bool ok = extra_box <= 4
                      ? extra_sample_value<4>(im, w, h, xf, yf, sg, pat->box_scaling[e], pat->box_scaling2[e], &v)
                      : extra_sample_value<9>(im, w, h, xf, yf, sg, pat->box_scaling[e], pat->box_scaling2[e], &v);
  *reinterpret_cast<int*>(slot_bytes + 24 + 4 * e) = v;
  return ok;
}

// One keypoint: valid byte (bit 0 = inside the rim and a usable ray, bits 1..6 = scale index of the
// scale-invariant extractor), M into the first 16 bytes of the (not yet written) descriptor slot, the
// record into kps_tmp.
__device__ __forceinline__ void describe_setup_one(const DescribeSetup& ds, int w, int h, int img, size_t slot,
                                                   const okvfe_keypoint& kp) {
  const ImageParams ip = ds.prm[img];
  int scale = 0;
  if (ds.scales) {
    for (int i = 1; i < kPatternScales; ++i) scale += kp.size >= ds.scales->size_from[i] ? 1 : 0;
  }
  const int border = ds.scales ? ds.scales->border[scale] : ds.pat->border;
  bool valid = !(kp.x < (float)border || kp.x >= (float)(w - border) || kp.y < (float)border ||
                 kp.y >= (float)(h - border));
  float M[4] = {1.0f, 0.0f, 0.0f, 1.0f};
  if (valid && ip.mode == kCameraAware) {
    const float dir[3] = {ip.dir[0], ip.dir[1], ip.dir[2]};
    valid = camera_aware_matrix(ds.rays[ip.cam], ds.jac[ip.cam], w, ip.fu, dir, kp.x, kp.y, M);
  }
  *reinterpret_cast<float4*>(ds.desc_tmp + slot * OKVFE_DESC_BYTES) = make_float4(M[0], M[1], M[2], M[3]);
  if (ip.mode == kCameraAware && !ds.scales) {
    // patch geometry for describe_aware_kernel (k_describe_aware.hip), bytes 16..23 of the slot: a superset of every
    // sample box under M -- |M p|_x <= |row_x(M)| |p|, boxes are not scaled by M, a box spans at most half a pixel
    // beyond x -+ sigma_half (the 0.75 leaves a quarter pixel for the float rounding of the positions) -- clipped to
    // the image (boxes that leave it drop the keypoint).  Class 0 / 1: rows of 64 / 80 bytes in LDS; 3: neither.
    int g0 = 0, g1 = 3 << 29;
    if (valid) {
      const float reach = ds.pat->reach;
      float nx = M[0] * M[0], t = M[1] * M[1];
      nx = sqrtf(nx + t) * 1.001f;
      float ny = M[2] * M[2];
      t = M[3] * M[3];
      ny = sqrtf(ny + t) * 1.001f;
      const float ex = fmaxf(nx, 1.0f) * reach + 0.75f, ey = fmaxf(ny, 1.0f) * reach + 0.75f;
      if (ex < 1024.0f && ey < 1024.0f) {  // (false for NaN)
        int bx0 = (int)floorf(kp.x - ex), bx1 = (int)ceilf(kp.x + ex);
        int by0 = (int)floorf(kp.y - ey), by1 = (int)ceilf(kp.y + ey);
        bx0 = bx0 < 0 ? 0 : bx0;
        by0 = by0 < 0 ? 0 : by0;
        bx1 = bx1 > w - 1 ? w - 1 : bx1;
        by1 = by1 > h - 1 ? h - 1 : by1;
        const int px0 = bx0 & ~3, pw = bx1 - px0 + 1, ph = by1 - by0 + 1;
#ifndef OKVFE_AWARE_PITCH80
#define OKVFE_AWARE_PITCH80 0  // A/B: every patch in the 80-byte-pitch class (rows step through 16 bank phases instead of 4)
#endif
        const int cls = (!OKVFE_AWARE_PITCH80 && pw <= 64 && ph <= 64) ? 0 : ((pw <= 80 && ph <= 72) ? 1 : 3);
        if (pw >= 1 && ph >= 1 && px0 < 4096 && by0 < 4096) {
          g0 = by0 * w + px0;
          g1 = (px0 >> 2) | (by0 << 10) | ((cls == 3 ? 0 : ph) << 22) | (cls << 29);
        }
      }
    }
    *reinterpret_cast<int2*>(ds.desc_tmp + slot * OKVFE_DESC_BYTES + 16) = make_int2(g0, g1);
    // ... and the samples beyond its 64 lanes, bytes 24.. of the slot (extra_box == 0: describe_extras_kernel does it)
    if (valid && ds.extra_box > 0) {
      const int extra = ds.pat->n_points - 64;  // 1 .. kAwareMaxExtra (host-checked)
      for (int e = 0; e < extra && valid; ++e)
        valid = extra_sample_one(ds.pat, ds.images + (size_t)img * w * h, w, h, ds.extra_box, e, M, kp.x, kp.y,
                                 ds.desc_tmp + slot * OKVFE_DESC_BYTES);
    }
  }
  ds.valid_tmp[slot] = (uint8_t)((valid ? 1 : 0) | (scale << 1));
  ds.kps_tmp[slot] = kp;  // the record travels on from here; describe_kernel only rewrites the angle
}

}  // namespace okvfe
