// k_nms.hip -- K2: 8-neighbour non-max suppression + compaction of the Harris score map.
//
// Replaces HarrisScoreCalculator::Get2dMaxima of the brisk library (behind
// cv::FeatureDetector::detect, okvis_cv/include/okvis/implementation/Frame.hpp:152).
// A centre (2 <= x < w-2, 2 <= y < h-2) is a maximum when score >= absoluteThreshold and no
// neighbour is strictly greater; of a horizontal run of equal passing pixels every second one,
// starting with the leftmost, is kept (the raster scan skips the pixel after a hit).
//
// Roofline: HBM, 4 B/px read + 12 B per maximum written.  One lane owns 4 consecutive pixels
// (one 16 B load per row for the centre row; neighbour rows are only fetched by lanes that hold
// a pixel above the threshold).  Maxima are appended per image through one wave-aggregated
// atomicAdd per wave; the set, not its order, is the contract (K3 sorts with a total order).
#include "okvfe_internal.h"

namespace okvfe {
namespace {

__device__ __forceinline__ bool passes(const int32_t* __restrict__ s, int w, int x, int y, int thr) {
  const int32_t* c = s + (size_t)y * w + x;
  const int v = c[0];
  if (v < thr) return false;
  if (c[1] > v || c[-1] > v) return false;
  const int32_t* p1 = c + w;
  const int32_t* p2 = c - w;
  if (p1[0] > v || p2[0] > v) return false;
  if (p1[1] > v || p1[-1] > v || p2[1] > v || p2[-1] > v) return false;
  return true;
}

__global__ __launch_bounds__(256) void nms_kernel(const int32_t* __restrict__ scores, int w, int h,
                                                  int thr, Candidate* __restrict__ cand,
                                                  int cand_cap, int32_t* __restrict__ cand_count) {
  const int img = blockIdx.z;
  const int32_t* s = scores + (size_t)img * w * h;
  const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
  const int y = blockIdx.y * 4 + threadIdx.y;
  bool acc[4] = {false, false, false, false};
  int val[4] = {0, 0, 0, 0};
  if (y >= 2 && y < h - 2 && x0 < w) {
    bool any = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int x = x0 + i;
      if (x >= 2 && x < w - 2) {
        val[i] = s[(size_t)y * w + x];
        any |= val[i] >= thr;
      }
    }
    if (any) {
      bool p[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int x = x0 + i;
        p[i] = (x >= 2 && x < w - 2 && val[i] >= thr) ? passes(s, w, x, y, thr) : false;
      }
      // run parity to the left of this lane's first pixel (only when the left neighbour can
      // pass too, i.e. carries the same score)
      bool left_acc = false;
      if (p[0] && x0 - 1 >= 2) {
        int run = 0;
        int xx = x0 - 1;
        while (xx >= 2 && passes(s, w, xx, y, thr)) {
          ++run;
          --xx;
        }
        left_acc = (run & 1) != 0;  // the pixel left of x0 is accepted iff its run index is even
      }
      bool prev = left_acc;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = p[i] && !prev;
        prev = acc[i];
      }
    }
  }
  // wave-aggregated append
  unsigned long long b[4];
  int total = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    b[i] = __ballot(acc[i]);
    total += __popcll(b[i]);
  }
  if (total == 0) return;
  const int lane = (threadIdx.y * 64 + threadIdx.x) & 63;
  int base = 0;
  if (lane == 0) base = atomicAdd(&cand_count[img], total);
  base = __shfl(base, 0);
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int off = base;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (acc[i]) {
      const int pos = off + __popcll(b[i] & lt);
      if (pos < cand_cap) {
        Candidate c;
        c.x = x0 + i;
        c.y = y;
        c.score = val[i];
        cand[(size_t)img * cand_cap + pos] = c;
      }
    }
    off += __popcll(b[i]);
  }
}

}  // namespace

void launch_nms(const int32_t* score, int w, int h, int n_images, int abs_threshold,
                Candidate* cand, int cand_cap, int32_t* cand_count, hipStream_t stream) {
  if (n_images <= 0) return;
  const dim3 block(64, 4, 1);
  const dim3 grid((w + 255) / 256, (h + 3) / 4, n_images);
  hipLaunchKernelGGL(nms_kernel, grid, block, 0, stream, score, w, h, abs_threshold, cand,
                     cand_cap, cand_count);
}

}  // namespace okvfe
