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
  ds.valid_tmp[slot] = (uint8_t)((valid ? 1 : 0) | (scale << 1));
  ds.kps_tmp[slot] = kp;  // the record travels on from here; describe_kernel only rewrites the angle
}

}  // namespace okvfe
