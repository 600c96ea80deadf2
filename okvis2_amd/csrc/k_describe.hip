// k_describe.hip -- K5 integral image, K6 BRISK2 descriptor, compaction + back-projection.
//
// Replaces brisk::BriskDescriptorExtractor::compute (behind cv::DescriptorExtractor::compute,
// okvis_cv/include/okvis/implementation/Frame.hpp:167; extractor built at
// okvis_frontend/src/Frontend.cpp:2410-2412, configured by setCameraProperties /
// setExtractionDirection at :239-251) and Frame::computeBackProjections
// (okvis_cv/include/okvis/implementation/Frame.hpp:178-193 ->
// cameras/implementation/PinholeCamera.hpp:574-593).
//
//   integral_kernel  inclusive integral image J[y][x] = sum_{r<=y, c<=x} img, int32.  One
//                    workgroup per image walks the rows (4 px per lane, wave scan + LDS carry,
//                    column accumulators in registers): every pixel is read once and every J
//                    written once (5 B/px, HBM-bound in large batches).
//   describe_kernel  one wave per keypoint, lane i = pattern point i (60 of 64 lanes): sample
//                    position kp + M p_i, box-smoothed intensity from J (13 taps) and 4 rim
//                    pixels, 383 pair comparisons as 6 wave ballots -> 6 x u64 = 48 bytes.
//                    L2-resident gathers; ALU/latency-bound, no HBM roofline.
//   compact_kernel   removes the keypoints the extractor dropped (order preserved) and
//                    back-projects the survivors in FP64 (Gauss-Newton undistortion).
#include <limits.h>

#include "okvfe_internal.h"

namespace okvfe {
namespace {

// ---- K5 -----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void integral_kernel(const uint8_t* __restrict__ images, int w,
                                                       int h, int32_t* __restrict__ integral) {
  __shared__ int wave_tot[2][4];
  const int img = blockIdx.x;
  const uint8_t* src = images + (size_t)img * w * h;
  int32_t* dst = integral + (size_t)img * w * h;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool vec = (w % 4 == 0) && ((reinterpret_cast<uintptr_t>(images) & 3) == 0) &&
                   ((reinterpret_cast<uintptr_t>(integral) & 15) == 0);
  const int nseg = (w + 1023) / 1024;
  // column accumulators: up to 4 segments of 1024 columns (w <= 4096), 4 columns per lane each
  int acc[4][4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[s][i] = 0;
  int buf = 0;
  for (int y = 0; y < h; ++y) {
    int carry = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < nseg) {
        const int x0 = s * 1024 + tid * 4;
        int p[4] = {0, 0, 0, 0};
        if (vec) {
          if (x0 < w) {
            const uint32_t d = *reinterpret_cast<const uint32_t*>(src + (size_t)y * w + x0);
            p[0] = d & 255; p[1] = (d >> 8) & 255; p[2] = (d >> 16) & 255; p[3] = d >> 24;
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (x0 + i < w) p[i] = src[(size_t)y * w + x0 + i];
        }
        p[1] += p[0]; p[2] += p[1]; p[3] += p[2];
        // inclusive wave scan of the lane totals
        int t = p[3];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int o = __shfl_up(t, d);
          if (lane >= d) t += o;
        }
        if (lane == 63) wave_tot[buf][wv] = t;
        __syncthreads();
        int base = carry + t - p[3];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int wt = wave_tot[buf][k];
          if (k < wv) base += wt;
          carry += wt;
        }
        buf ^= 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[s][i] += base + p[i];
        if (vec) {
          if (x0 < w)
            *reinterpret_cast<int4*>(dst + (size_t)y * w + x0) =
                make_int4(acc[s][0], acc[s][1], acc[s][2], acc[s][3]);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (x0 + i < w) dst[(size_t)y * w + x0 + i] = acc[s][i];
        }
      }
    }
  }
}

// ---- K6 -----------------------------------------------------------------------------------
__device__ __forceinline__ int isum(const int32_t* __restrict__ J, int w, int y, int x) {
  // exclusive integral I[y][x] = sum rows < y, cols < x
  return (x > 0 && y > 0) ? J[(size_t)(y - 1) * w + (x - 1)] : 0;
}
#define RECT(xa, ya, xb, yb) \
  (isum(J, w, (yb), (xb)) - isum(J, w, (ya), (xb)) - isum(J, w, (yb), (xa)) + isum(J, w, (ya), (xa)))

// Box of half-side sigma_half centred at (xf, yf); returns 1024 * mean intensity.  Same integer /
// float operation sequence as the published BRISK smoothedIntensity (sub-pixel rim weights).
__device__ __forceinline__ int smoothed_intensity(const uint8_t* __restrict__ img,
                                                  const int32_t* __restrict__ J, int w, float xf,
                                                  float yf, float sigma_half) {
  if (sigma_half < 0.5f) {
    const int x = (int)xf, y = (int)yf;
    const int r_x = (int)((xf - (float)x) * 1024.0f);
    const int r_y = (int)((yf - (float)y) * 1024.0f);
    const int r_x_1 = 1024 - r_x, r_y_1 = 1024 - r_y;
    const uint8_t* ptr = img + (size_t)y * w + x;
    int ret = r_x_1 * r_y_1 * (int)ptr[0];
    ret += r_x * r_y_1 * (int)ptr[1];
    ret += r_x * r_y * (int)ptr[w + 1];
    ret += r_x_1 * r_y * (int)ptr[w];
    return (ret + 512) / 1024;
  }
  float area = 4.0f * sigma_half;
  area = area * sigma_half;
  const int scaling = (int)(4194304.0f / area);
  const float s2 = (float)scaling * area;
  const int scaling2 = (int)(s2 / 1024.0f);
  const float x_1 = xf - sigma_half, x1 = xf + sigma_half;
  const float y_1 = yf - sigma_half, y1 = yf + sigma_half;
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
  int ret = A * (int)img[(size_t)y_top * w + x_left];
  ret += B * (int)img[(size_t)y_top * w + x_right];
  ret += C * (int)img[(size_t)y_bottom * w + x_right];
  ret += D * (int)img[(size_t)y_bottom * w + x_left];
  const int upper = RECT(x_left + 1, y_top, x_right, y_top + 1);
  const int middle = RECT(x_left + 1, y_top + 1, x_right, y_bottom);
  const int left = RECT(x_left, y_top + 1, x_left + 1, y_bottom);
  const int right = RECT(x_right, y_top + 1, x_right + 1, y_bottom);
  const int bottom = RECT(x_left + 1, y_bottom, x_right, y_bottom + 1);
  ret += upper * r_y_1_i + middle * scaling + left * r_x_1_i + right * r_x1_i + bottom * r_y1_i;
  return (ret + scaling2 / 2) / scaling2;
}
#undef RECT

// sample position of this lane's pattern point under M; ok = box inside the image (NaN-safe)
__device__ __forceinline__ bool sample_pos(const float M[4], float kx, float ky, float px, float py,
                                           float sg, int w, int h, float* xf, float* yf) {
  float a = M[0] * px;
  float b = M[1] * py;
  a = a + b;
  *xf = kx + a;
  float c = M[2] * px;
  float d = M[3] * py;
  c = c + d;
  *yf = ky + c;
  const float x_1 = *xf - sg, x1 = *xf + sg, y_1 = *yf - sg, y1 = *yf + sg;
  return (x_1 >= 0.0f && y_1 >= 0.0f && x1 < (float)(w - 1) && y1 < (float)(h - 1));
}

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

constexpr int kDescWaves = 4;

__global__ __launch_bounds__(64 * kDescWaves) void describe_kernel(
    const uint8_t* __restrict__ images, const int32_t* __restrict__ integral, int w, int h,
    const Pattern* __restrict__ pat, const ImageParams* __restrict__ prm,
    const float* const* __restrict__ rays, const float* const* __restrict__ jac,
    const okvfe_keypoint* __restrict__ kps_in, int kp_cap, const int32_t* __restrict__ kp_count_in,
    okvfe_keypoint* __restrict__ kps_tmp, uint8_t* __restrict__ desc_tmp,
    uint8_t* __restrict__ valid_tmp) {
  __shared__ int values[kDescWaves][64];
  const int img = blockIdx.y;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int k = blockIdx.x * kDescWaves + wv;
  const int n = kp_count_in[img];
  if (k >= n) return;  // whole wave exits; no block-wide barriers below
  const uint8_t* im = images + (size_t)img * w * h;
  const int32_t* J = integral + (size_t)img * w * h;
  const size_t slot = (size_t)img * kp_cap + k;
  okvfe_keypoint kp = kps_in[slot];
  const ImageParams ip = prm[img];
  const int border = pat->border;
  bool valid = !(kp.x < (float)border || kp.x >= (float)(w - border) || kp.y < (float)border ||
                 kp.y >= (float)(h - border));
  const int li = lane < kPatternPoints ? lane : 0;
  const float px = pat->px[li], py = pat->py[li], sg = pat->sigma_half[li];
  float M[4] = {1.0f, 0.0f, 0.0f, 1.0f};
  float xf, yf;
  int* vals = values[wv];
  if (valid && ip.mode == kCameraAware) {
    const float dir[3] = {ip.dir[0], ip.dir[1], ip.dir[2]};
    valid = camera_aware_matrix(rays[ip.cam], jac[ip.cam], w, ip.fu, dir, kp.x, kp.y, M);
  } else if (valid && ip.mode == kGradient) {
    bool ok = sample_pos(M, kp.x, kp.y, px, py, sg, w, h, &xf, &yf);
    valid = __all(ok || lane >= kPatternPoints);
    if (valid) {
      vals[lane] = lane < kPatternPoints ? smoothed_intensity(im, J, w, xf, yf, sg) : 0;
      __builtin_amdgcn_wave_barrier();
      int d0 = 0, d1 = 0;
      for (int l = lane; l < pat->n_long; l += 64) {
        const int delta_t = vals[pat->long_i[l]] - vals[pat->long_j[l]];
        d0 += delta_t * pat->long_wdx[l] / 1024;
        d1 += delta_t * pat->long_wdy[l] / 1024;
      }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        d0 += __shfl_xor(d0, d);
        d1 += __shfl_xor(d1, d);
      }
      int best_k = 0;
      if (d0 != 0 || d1 != 0) {
        long long best = LLONG_MIN;
        int bk = 0;
        for (int r = lane * 16; r < lane * 16 + 16; ++r) {
          const long long dot = (long long)d0 * pat->rot_cos[r] + (long long)d1 * pat->rot_sin[r];
          if (dot > best) {
            best = dot;
            bk = r;
          }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
          const long long ob = __shfl_xor(best, d);
          const int ok2 = __shfl_xor(bk, d);
          if (ob > best || (ob == best && ok2 < bk)) {
            best = ob;
            bk = ok2;
          }
        }
        best_k = bk;
      }
      kp.angle = (float)best_k * 0.3515625f;
      M[0] = pat->rot_cosf[best_k];
      M[1] = -pat->rot_sinf[best_k];
      M[2] = pat->rot_sinf[best_k];
      M[3] = pat->rot_cosf[best_k];
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (valid) {
    const bool ok = sample_pos(M, kp.x, kp.y, px, py, sg, w, h, &xf, &yf);
    valid = __all(ok || lane >= kPatternPoints);
  }
  if (valid) {
    vals[lane] = lane < kPatternPoints ? smoothed_intensity(im, J, w, xf, yf, sg) : 0;
    __builtin_amdgcn_wave_barrier();
    unsigned long long words[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int b = j * 64 + lane;
      bool bit = false;
      if (b < pat->n_short) bit = vals[pat->short_i[b]] > vals[pat->short_j[b]];
      words[j] = __ballot(bit);
    }
    if (lane < 6) {
      unsigned long long wsel = words[0];
#pragma unroll
      for (int j = 1; j < 6; ++j)
        if (lane == j) wsel = words[j];
      reinterpret_cast<unsigned long long*>(desc_tmp + slot * OKVFE_DESC_BYTES)[lane] = wsel;
    }
  }
  if (lane == 0) {
    kps_tmp[slot] = kp;
    valid_tmp[slot] = valid ? 1 : 0;
  }
}

// ---- compaction + back-projection -----------------------------------------------------------
__device__ void distort(const DeviceCamera& c, double u0, double u1, double out[2], double J[4]) {
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
  // equidistant (device atan; see DESIGN.md on its last-ulp caveat)
  const double k1 = c.d[0], k2 = c.d[1], k3 = c.d[2], k4 = c.d[3];
  const double r = sqrt(u0 * u0 + u1 * u1);
  const double theta = atan(r);
  const double theta2 = theta * theta;
  const double theta4 = theta2 * theta2;
  const double theta6 = theta4 * theta2;
  const double theta8 = theta4 * theta4;
  const double thetad = theta * (1.0 + k1 * theta2 + k2 * theta4 + k3 * theta6 + k4 * theta8);
  const double scaling = (r > 1e-8) ? thetad / r : 1.0;
  out[0] = scaling * u0;
  out[1] = scaling * u1;
  if (r > 1e-8) {
    double t2, t3, t4, t6, t7, t8, t9, t11, t17, t18, t19, t20, t25;
    t2 = u0 * u0;
    t3 = u1 * u1;
    t4 = t2 + t3;
    t6 = atan(sqrt(t4));
    t7 = t6 * t6;
    t8 = 1.0 / sqrt(t4);
    t9 = t7 * t7;
    t11 = 1.0 / ((t2 + t3) + 1.0);
    t17 = (((k1 * t7 + k2 * t9) + k3 * t7 * t9) + k4 * (t9 * t9)) + 1.0;
    t18 = 1.0 / t4;
    t19 = 1.0 / sqrt(t4 * t4 * t4);
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
  } else {
    J[0] = 1.0; J[1] = 0.0; J[2] = 0.0; J[3] = 1.0;
  }
}

__device__ bool backproject(const DeviceCamera& c, double px, double py, double dir[3]) {
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

__global__ __launch_bounds__(256) void compact_kernel(
    const DeviceCamera* __restrict__ cams, const ImageParams* __restrict__ prm,
    const okvfe_keypoint* __restrict__ kps_tmp, const uint8_t* __restrict__ desc_tmp,
    const uint8_t* __restrict__ valid_tmp, const int32_t* __restrict__ kp_count_in, int kp_cap,
    okvfe_keypoint* __restrict__ kps, uint8_t* __restrict__ desc, double* __restrict__ bp,
    uint8_t* __restrict__ bpv, int32_t* __restrict__ kp_count) {
  __shared__ int wave_cnt[4];
  __shared__ int base_s;
  const int img = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = kp_count_in[img];
  const size_t off = (size_t)img * kp_cap;
  const int cam = prm[img].cam;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int k0 = 0; k0 < n; k0 += 256) {
    const int k = k0 + tid;
    const bool v = k < n && valid_tmp[off + k] != 0;
    const unsigned long long b = __ballot(v);
    if (lane == 0) wave_cnt[wv] = __popcll(b);
    __syncthreads();
    int pos = base_s;
    for (int i = 0; i < wv; ++i) pos += wave_cnt[i];
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    pos += __popcll(b & lt);
    if (v) {
      const okvfe_keypoint kp = kps_tmp[off + k];
      kps[off + pos] = kp;
      const uint4* s = reinterpret_cast<const uint4*>(desc_tmp + (off + k) * OKVFE_DESC_BYTES);
      uint4* d = reinterpret_cast<uint4*>(desc + (off + pos) * OKVFE_DESC_BYTES);
      d[0] = s[0];
      d[1] = s[1];
      d[2] = s[2];
      double dir[3] = {0.0, 0.0, 0.0};
      bool ok = false;
      if (cam >= 0 && cams[cam].fu > 0.0) ok = backproject(cams[cam], (double)kp.x, (double)kp.y, dir);
      bp[(off + pos) * 3 + 0] = dir[0];
      bp[(off + pos) * 3 + 1] = dir[1];
      bp[(off + pos) * 3 + 2] = dir[2];
      bpv[off + pos] = ok ? 1 : 0;
    }
    __syncthreads();
    if (tid == 0) base_s += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (tid == 0) kp_count[img] = base_s;
}

}  // namespace

void launch_integral(const uint8_t* img, int w, int h, int n_images, int32_t* integral,
                     hipStream_t stream) {
  if (n_images <= 0) return;
  hipLaunchKernelGGL(integral_kernel, dim3(n_images), dim3(256), 0, stream, img, w, h, integral);
}

void launch_describe(const uint8_t* img, const int32_t* integral, int w, int h, int n_images,
                     const Pattern* pat, const ImageParams* prm, const float* const* rays,
                     const float* const* jac, const okvfe_keypoint* kps_in, int kp_cap,
                     const int32_t* kp_count_in, okvfe_keypoint* kps_tmp, uint8_t* desc_tmp,
                     uint8_t* valid_tmp, hipStream_t stream) {
  if (n_images <= 0) return;
  const dim3 grid((kp_cap + kDescWaves - 1) / kDescWaves, n_images);
  hipLaunchKernelGGL(describe_kernel, grid, dim3(64 * kDescWaves), 0, stream, img, integral, w, h,
                     pat, prm, rays, jac, kps_in, kp_cap, kp_count_in, kps_tmp, desc_tmp,
                     valid_tmp);
}

void launch_compact(int n_images, const DeviceCamera* cams, const ImageParams* prm,
                    const okvfe_keypoint* kps_tmp, const uint8_t* desc_tmp,
                    const uint8_t* valid_tmp, const int32_t* kp_count_in, int kp_cap,
                    okvfe_keypoint* kps, uint8_t* desc, double* bp, uint8_t* bpv,
                    int32_t* kp_count, hipStream_t stream) {
  if (n_images <= 0) return;
  hipLaunchKernelGGL(compact_kernel, dim3(n_images), dim3(256), 0, stream, cams, prm, kps_tmp,
                     desc_tmp, valid_tmp, kp_count_in, kp_cap, kps, desc, bp, bpv, kp_count);
}

}  // namespace okvfe
