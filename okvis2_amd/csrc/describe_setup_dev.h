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
        const int cls = (pw <= 64 && ph <= 64) ? 0 : ((pw <= 80 && ph <= 72) ? 1 : 3);
        if (pw >= 1 && ph >= 1 && px0 < 4096 && by0 < 4096) {
          g0 = by0 * w + px0;
          g1 = (px0 >> 2) | (by0 << 10) | ((cls == 3 ? 0 : ph) << 22) | (cls << 29);
        }
      }
    }
    *reinterpret_cast<int2*>(ds.desc_tmp + slot * OKVFE_DESC_BYTES + 16) = make_int2(g0, g1);
  }
  ds.valid_tmp[slot] = (uint8_t)((valid ? 1 : 0) | (scale << 1));
  ds.kps_tmp[slot] = kp;  // the record travels on from here; describe_kernel only rewrites the angle
}

}  // namespace okvfe
