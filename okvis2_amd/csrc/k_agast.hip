// k_agast.hip -- AGAST / FAST 9-16 corner score map (okvfe_config.score_type = OKVFE_SCORE_AGAST_9_16).
//
// Score calculator of brisk::BriskFeatureDetector, the detector the reference instantiates on ARM
// (okvis_cv/test/TestFrame.cpp:71-72: BriskFeatureDetector(34, 2)); the x86 path and every shipped
// configuration use the Harris calculator (k_harris.hip).  The brisk library is not vendored with
// the reference, so this follows the PUBLISHED predicate (FAST-9 on the 16-pixel Bresenham circle
// of radius 3; AGAST evaluates the same predicate with a different decision tree): corner at
// threshold t iff 9 contiguous circle pixels are all > p + t or all < p - t; score = the largest
// such t = max(max_s min_k (c - p), max_s min_k (p - c)) - 1, clamped at 0; 0 within 3 px of the
// border.  The rest of the detector (NMS, scale-space maxima, uniformity, cap, sub-pixel) is the
// shared pipeline; the oracle counterpart is orc_agast_score (oracle/orc_detect.c).
//
// One workgroup = 64 x 16 pixels staged in LDS with a 3-pixel apron; a thread scores 4 pixels of
// one column.  The 16 windows of 9 contiguous differences are built from 3-windows:
// m3[i] = min3(d[i], d[i+1], d[i+2]), m9[i] = min3(m3[i], m3[i+3], m3[i+6]) -- 32 v_min3 + 32
// v_max3 per pixel instead of 256 compares.  Integer-VALU bound (~110 instructions per pixel);
// HBM traffic is the same 1 B in + 4 B out per pixel as the Harris kernel.
#include "okvfe_internal.h"

namespace okvfe {
namespace {

constexpr int kTileW = 64, kTileH = 16, kApron = 3;
constexpr int kLdsPitch = 72;  // 18 dwords: columns x0 - 4 .. x0 + 67 (>= kTileW + 2 * kApron)
constexpr int kLdsX = 4;       // LDS column of image column x0: the row starts one aligned dword to the left

__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int max3i_(int a, int b, int c) { return max(max(a, b), c); }

__global__ __launch_bounds__(256) void agast_score_kernel(const uint8_t* __restrict__ images, int w, int h,
                                                          int32_t* __restrict__ scores, int tiles_x,
                                                          int tiles_y, int n_images) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[kTileH + 2 * kApron][kLdsPitch];
  int image, t;
  xcd_tile(tiles_x * tiles_y, n_images, &image, &t);
  const int ty0 = t / tiles_x, tx0 = t - ty0 * tiles_x;
  const int x0 = tx0 * kTileW, y0 = ty0 * kTileH;
  const uint8_t* img = images + (size_t)image * w * h;
  int32_t* out = scores + (size_t)image * w * h;
  const int tid = threadIdx.x;
  // stage the tile; coordinates outside the image are clamped (those values only reach pixels
  // whose score is 0 by the border rule).  Rows of 4-aligned widths move as 18 aligned dwords (the
  // byte-wise loop below was the kernel's actual bound: 6 byte loads + LDS byte stores per thread)
  if ((w & 3) == 0 && (reinterpret_cast<uintptr_t>(img) & 3) == 0) {
    const int ndw = w >> 2;
    for (int i = tid; i < (kTileH + 2 * kApron) * (kLdsPitch / 4); i += 256) {
      const int r = i / (kLdsPitch / 4), c = i - r * (kLdsPitch / 4);
      int yy = y0 - kApron + r, dq = (x0 >> 2) - 1 + c;
      yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
      dq = dq < 0 ? 0 : (dq > ndw - 1 ? ndw - 1 : dq);
      reinterpret_cast<uint32_t*>(&tile[r][0])[c] = reinterpret_cast<const uint32_t*>(img + (size_t)yy * w)[dq];
    }
  } else {
    for (int i = tid; i < (kTileH + 2 * kApron) * (kTileW + 2 * kApron); i += 256) {
      const int r = i / (kTileW + 2 * kApron), c = i - r * (kTileW + 2 * kApron);
      int yy = y0 - kApron + r, xx = x0 - kApron + c;
      yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
      xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
      tile[r][c + kLdsX - kApron] = img[(size_t)yy * w + xx];
    }
  }
  __syncthreads();
  const int tx = tid & 63, tyy = tid >> 6;
  const int x = x0 + tx;
  if (x >= w) return;
#pragma unroll
  for (int k = 0; k < kTileH / 4; ++k) {
    const int ly = tyy + 4 * k, y = y0 + ly;
    if (y >= h) break;
    int s = 0;
    if (x >= 3 && y >= 3 && x < w - 3 && y < h - 3) {
      const uint8_t* c = &tile[ly + kApron][tx + kLdsX];
      const int p = c[0];
      // circle in the order of the oracle's table: (0,3) (1,3) (2,2) (3,1) (3,0) (3,-1) (2,-2) (1,-3)
      // (0,-3) (-1,-3) (-2,-2) (-3,-1) (-3,0) (-3,1) (-2,2) (-1,3)
      int d[16];
      d[0] = c[3 * kLdsPitch + 0] - p;    d[1] = c[3 * kLdsPitch + 1] - p;
      d[2] = c[2 * kLdsPitch + 2] - p;    d[3] = c[1 * kLdsPitch + 3] - p;
      d[4] = c[3] - p;                    d[5] = c[-1 * kLdsPitch + 3] - p;
      d[6] = c[-2 * kLdsPitch + 2] - p;   d[7] = c[-3 * kLdsPitch + 1] - p;
      d[8] = c[-3 * kLdsPitch + 0] - p;   d[9] = c[-3 * kLdsPitch - 1] - p;
      d[10] = c[-2 * kLdsPitch - 2] - p;  d[11] = c[-1 * kLdsPitch - 3] - p;
      d[12] = c[-3] - p;                  d[13] = c[1 * kLdsPitch - 3] - p;
      d[14] = c[2 * kLdsPitch - 2] - p;   d[15] = c[3 * kLdsPitch - 1] - p;
      int mn3[16], mx3[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        mn3[i] = min3i(d[i], d[(i + 1) & 15], d[(i + 2) & 15]);
        mx3[i] = max3i_(d[i], d[(i + 1) & 15], d[(i + 2) & 15]);
      }
      int bright = -256, most = 256;  // max over arcs of the arc minimum; min over arcs of the arc maximum
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        bright = max(bright, min3i(mn3[i], mn3[(i + 3) & 15], mn3[(i + 6) & 15]));
        most = min(most, max3i_(mx3[i], mx3[(i + 3) & 15], mx3[(i + 6) & 15]));
      }
      const int dark = -most;
      s = max(bright, dark) - 1;
      s = s < 0 ? 0 : s;
    }
    out[(size_t)y * w + x] = s;
  }
}

// FAST 5-8 score of c0: the virtual intra-octave below the first octave of the published BRISK
// scale space (oracle: orc_fast58_score).  One thread per pixel, direct reads (a small fraction of
// the 9-16 kernel's work; the 3 x 3 neighbourhood comes out of L1 / L2).
__global__ __launch_bounds__(256) void fast58_score_kernel(const uint8_t* __restrict__ images, int w, int h,
                                                           int32_t* __restrict__ scores) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const uint8_t* img = images + (size_t)blockIdx.z * w * h;
  int s = 0;
  if (x >= 1 && y >= 1 && x < w - 1 && y < h - 1) {
    const uint8_t* c = img + (size_t)y * w + x;
    const int p = c[0];
    // ring in the oracle's order: (0,1) (1,1) (1,0) (1,-1) (0,-1) (-1,-1) (-1,0) (-1,1)
    int d[8];
    d[0] = c[w] - p;       d[1] = c[w + 1] - p;  d[2] = c[1] - p;   d[3] = c[-w + 1] - p;
    d[4] = c[-w] - p;      d[5] = c[-w - 1] - p; d[6] = c[-1] - p;  d[7] = c[w - 1] - p;
    int bright = -256, most = 256;
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const int mn = min(min3i(d[st], d[(st + 1) & 7], d[(st + 2) & 7]), min(d[(st + 3) & 7], d[(st + 4) & 7]));
      const int mx = max(max3i_(d[st], d[(st + 1) & 7], d[(st + 2) & 7]), max(d[(st + 3) & 7], d[(st + 4) & 7]));
      bright = max(bright, mn);
      most = min(most, mx);
    }
    s = max(bright, -most) - 1;
    s = s < 0 ? 0 : s;
  }
  scores[(size_t)blockIdx.z * w * h + (size_t)y * w + x] = s;
}

}  // namespace

void launch_fast58_score(const uint8_t* img, int w, int h, int n_images, int32_t* score, hipStream_t stream) {
  if (n_images <= 0) return;
  hipLaunchKernelGGL(fast58_score_kernel, dim3((w + 255) / 256, h, n_images), dim3(256), 0, stream, img, w, h, score);
}

void launch_agast_score(const uint8_t* img, int w, int h, int n_images, int32_t* score, hipStream_t stream) {
  if (n_images <= 0) return;
  const int tiles_x = (w + kTileW - 1) / kTileW, tiles_y = (h + kTileH - 1) / kTileH;
  hipLaunchKernelGGL(agast_score_kernel, dim3(tiles_x * tiles_y * n_images), dim3(256), 0, stream, img, w, h,
                     score, tiles_x, tiles_y, n_images);
}

}  // namespace okvfe
