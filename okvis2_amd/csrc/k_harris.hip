// k_harris.hip -- K1: Harris corner score map, u8 image -> int32 score (gfx950).
//
// Replaces the whole-image passes of brisk::HarrisScoreCalculator (behind
// cv::FeatureDetector::detect, okvis_cv/include/okvis/implementation/Frame.hpp:152; detector
// built at okvis_frontend/src/Frontend.cpp:2406-2409) by ONE streaming pass:
//   Scharr (3,10,3) gradients -> gx^2, gy^2, gx*gy >> 14 (16-bit covariance entries, zero on the
//   image rim) -> 3x3 binomial sum -> det - (trace/4)^2, zero on the rim.
//
// Roofline: HBM.  Algorithmic bytes per pixel = 1 (u8 in) + 4 (int32 out) = 5.
// Mapping: one wave = a 256-pixel-wide column strip (4 px per lane, one aligned dword load per
// lane per row = 256 B coalesced per wave, one 16 B store per lane per row = 1 KiB per wave);
// the wave walks TH rows down the strip keeping a rolling 3-row window of pixels and of
// horizontally smoothed covariance entries in registers, so every pixel is fetched once per
// strip (+4 halo rows per TH) and nothing is staged through LDS.  All products fit 24 bits
// (|g| <= 4080, entries <= 16256), so the multiplies are full-rate v_mul_i32_i24 / v_mad_i32_i24.
#include "okvfe_internal.h"

namespace okvfe {

namespace {

constexpr int kTH = 32;          // output rows per wave
constexpr int kWavesPerBlock = 4;

__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }

// pixels of columns x0-2 .. x0+5 of one row into p[0..7]
template <bool ALIGNED>
__device__ __forceinline__ void load_row(const uint8_t* __restrict__ img, int w, int h, int row,
                                         int x0, int p[8]) {
  row = row < 0 ? 0 : (row > h - 1 ? h - 1 : row);
  const uint8_t* rp = img + (size_t)row * w;
  if (ALIGNED) {
    // w % 4 == 0 and the image base is 4-byte aligned: three aligned dwords, indices clamped
    // (clamped lanes only feed masked rim entries)
    const uint32_t* rq = reinterpret_cast<const uint32_t*>(rp);
    const int nd = w >> 2;
    int dc = x0 >> 2;
    dc = dc > nd - 1 ? nd - 1 : dc;
    const int dl = dc > 0 ? dc - 1 : 0;
    const int dr = dc < nd - 1 ? dc + 1 : nd - 1;
    const uint32_t L = rq[dl], C = rq[dc], R = rq[dr];
    p[0] = (L >> 16) & 255;
    p[1] = (L >> 24);
    p[2] = C & 255;
    p[3] = (C >> 8) & 255;
    p[4] = (C >> 16) & 255;
    p[5] = (C >> 24);
    p[6] = R & 255;
    p[7] = (R >> 8) & 255;
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int c = x0 - 2 + i;
      c = c < 0 ? 0 : (c > w - 1 ? w - 1 : c);
      p[i] = rp[c];
    }
  }
}

// Covariance entries of row g at columns x0-1 .. x0+4 from pixel rows a (g-1), b (g), c (g+1),
// then the horizontal binomial [1 2 1] for the 4 output columns -> hs[3][4].
__device__ __forceinline__ void cov_row(const int a[8], const int b[8], const int c[8], int g, int x0,
                                        int w, int h, int hs[3][4]) {
  int vs[8], vd[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    vs[i] = 3 * (a[i] + c[i]) + 10 * b[i];  // vertical (3,10,3)
    vd[i] = c[i] - a[i];                     // vertical difference
  }
  int gxx[6], gyy[6], gxy[6];
  const bool row_ok = (g >= 1) && (g <= h - 2);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int col = x0 - 1 + i;
    const int gx = vs[i + 2] - vs[i];
    const int gy = 3 * (vd[i] + vd[i + 2]) + 10 * vd[i + 1];
    const bool ok = row_ok && (col >= 1) && (col <= w - 2);
    gxx[i] = ok ? (mul24(gx, gx) >> 14) : 0;
    gyy[i] = ok ? (mul24(gy, gy) >> 14) : 0;
    gxy[i] = ok ? (mul24(gx, gy) >> 14) : 0;  // arithmetic shift = floor
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hs[0][i] = gxx[i] + 2 * gxx[i + 1] + gxx[i + 2];
    hs[1][i] = gyy[i] + 2 * gyy[i + 1] + gyy[i + 2];
    hs[2][i] = gxy[i] + 2 * gxy[i + 1] + gxy[i + 2];
  }
}

template <bool ALIGNED>
__global__ __launch_bounds__(64 * kWavesPerBlock) void harris_kernel(
    const uint8_t* __restrict__ images, int w, int h, int32_t* __restrict__ scores) {
  const int lane = threadIdx.x;
  const int x0 = (blockIdx.x * 64 + lane) * 4;
  const int ys = (blockIdx.y * kWavesPerBlock + threadIdx.y) * kTH;
  if (ys >= h) return;
  const int ye = ys + kTH < h ? ys + kTH : h;
  const size_t img_off = (size_t)blockIdx.z * (size_t)w * (size_t)h;
  const uint8_t* img = images + img_off;
  int32_t* out = scores + img_off;

  int pr[3][8];      // rolling pixel rows, slot = row mod 3 (relative)
  int hsr[3][3][4];  // rolling horizontally smoothed entries, slot = row mod 3 (relative)

  // prologue: pixel rows ys-2, ys-1 -> slots 0, 1 ; the loop loads row r = ys-2+k into slot k%3
  load_row<ALIGNED>(img, w, h, ys - 2, x0, pr[0]);
  load_row<ALIGNED>(img, w, h, ys - 1, x0, pr[1]);

  // iteration k (k = 2, 3, ...): load pixel row r = ys-2+k, produce covariance row g = r-1
  // into hs slot (k-2)%3... relative numbering j = k-2 = 0,1,2,...: g = ys-1+j ; once j >= 2 the
  // score row y = g-1 = ys+j-2 is complete.
  auto step = [&](int j, int s_new, int s_a, int s_b, int h_new, int h_a, int h_b) {
    const int r = ys + j;  // pixel row loaded this step
    load_row<ALIGNED>(img, w, h, r, x0, pr[s_new]);
    const int g = r - 1;
    cov_row(pr[s_a], pr[s_b], pr[s_new], g, x0, w, h, hsr[h_new]);
    if (j >= 2) {
      const int y = g - 1;
      if (y < ye) {
        int sc[4];
        const bool yrow = (y >= 1) && (y <= h - 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int A = hsr[h_a][0][i] + 2 * hsr[h_b][0][i] + hsr[h_new][0][i];
          const int B = hsr[h_a][1][i] + 2 * hsr[h_b][1][i] + hsr[h_new][1][i];
          const int Cc = hsr[h_a][2][i] + 2 * hsr[h_b][2][i] + hsr[h_new][2][i];
          const int det = mul24(A, B) - mul24(Cc, Cc);
          const int tq = ((A >> 1) + (B >> 1)) >> 1;
          const int x = x0 + i;
          const bool ok = yrow && (x >= 1) && (x <= w - 2);
          sc[i] = ok ? det - mul24(tq, tq) : 0;
        }
        int32_t* op = out + (size_t)y * w + x0;
        if (ALIGNED && x0 + 3 < w) {
          *reinterpret_cast<int4*>(op) = make_int4(sc[0], sc[1], sc[2], sc[3]);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (x0 + i < w) op[i] = sc[i];
        }
      }
    }
  };

  // j = 0 .. (ye-ys)+1 ; unrolled by 3 so that all rolling-buffer slots are compile-time
  const int jn = (ye - ys) + 2;
  for (int j = 0; j < jn; j += 3) {
    // slots: pixel row ys+j lives in slot (j+2)%3 ; with j % 3 == 0: new=2, a=0, b=1
    step(j, 2, 0, 1, 0, 1, 2);
    if (j + 1 < jn) step(j + 1, 0, 1, 2, 1, 2, 0);
    if (j + 2 < jn) step(j + 2, 1, 2, 0, 2, 0, 1);
  }
}

}  // namespace

void launch_harris(const uint8_t* img, int w, int h, int n_images, int32_t* score,
                   hipStream_t stream) {
  if (n_images <= 0) return;
  const dim3 block(64, kWavesPerBlock, 1);
  const dim3 grid((w + 255) / 256, (h + kTH * kWavesPerBlock - 1) / (kTH * kWavesPerBlock),
                  n_images);
  const bool aligned = (w % 4 == 0) && ((reinterpret_cast<uintptr_t>(img) & 3) == 0) &&
                       ((reinterpret_cast<uintptr_t>(score) & 15) == 0);
  if (aligned)
    hipLaunchKernelGGL(harris_kernel<true>, grid, block, 0, stream, img, w, h, score);
  else
    hipLaunchKernelGGL(harris_kernel<false>, grid, block, 0, stream, img, w, h, score);
}

}  // namespace okvfe
