// k_nms.hip -- K2: 8-neighbour non-max suppression + compaction of the Harris score map.
//
// Replaces HarrisScoreCalculator::Get2dMaxima of the brisk library (behind
// cv::FeatureDetector::detect, okvis_cv/include/okvis/implementation/Frame.hpp:152).
// A centre (2 <= x < w-2, 2 <= y < h-2) is a maximum when score >= absoluteThreshold and no
// neighbour is strictly greater; of a horizontal run of equal passing pixels every second one,
// starting with the leftmost, is kept (the raster scan skips the pixel after a hit).
//
// Roofline: HBM, 4 B/px read + 12 B per maximum written.
// Fast path (w % 4 == 0): same strip mapping as K1 -- lane = one 16-byte column group (4 px),
// 62 valid lanes per wave, the wave walks down its strip with a rolling window of three rows:
// every score is loaded exactly once (1 KiB per wave per row); horizontal neighbours come from
// the adjacent lanes through DPP wave shifts; the per-row horizontal 3-max is computed once and
// reused for the row above and below.  Maxima are collected in LDS and appended to the image's
// candidate list with ONE global atomic per workgroup.  Equal-score horizontal runs (which need
// the serial parity rule) are detected per wave and resolved on a slow path.
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

// accepted(x) of the raster scan for a pixel that passes: parity of the run of passing pixels
// immediately to its left
__device__ __forceinline__ bool accepted_slow(const int32_t* __restrict__ s, int w, int x, int y,
                                              int thr) {
  if (x < 2 || x >= w - 2 || !passes(s, w, x, y, thr)) return false;
  int run = 0;
  int xx = x - 1;
  while (xx >= 2 && passes(s, w, xx, y, thr)) {
    ++run;
    --xx;
  }
  return (run & 1) == 0;
}

// ---- generic kernel (any width): one lane = 4 pixels of one row ---------------------------------
__global__ __launch_bounds__(256) void nms_generic_kernel(const int32_t* __restrict__ scores, int w,
                                                          int h, int thr,
                                                          Candidate* __restrict__ cand,
                                                          int cand_cap,
                                                          int32_t* __restrict__ cand_count) {
  const int img = blockIdx.z;
  const int32_t* s = scores + (size_t)img * w * h;
  const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
  const int y = blockIdx.y * 4 + threadIdx.y;
  bool acc[4] = {false, false, false, false};
  int val[4] = {0, 0, 0, 0};
  if (y >= 2 && y < h - 2 && x0 < w) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int x = x0 + i;
      if (x >= 2 && x < w - 2) {
        val[i] = s[(size_t)y * w + x];
        if (val[i] >= thr) acc[i] = accepted_slow(s, w, x, y, thr);
      }
    }
  }
  unsigned long long b[4];
  int total = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    b[i] = __ballot(acc[i]);
    total += __popcll(b[i]);
  }
  if (total == 0) return;
  const int lane = threadIdx.x;
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

// ---- fast kernel --------------------------------------------------------------------------------
constexpr int kStripLanes = 62;
constexpr int kRows = 32;   // centre rows per wave
constexpr int kWaves = 4;
constexpr int kLdsCap = 1024;

__device__ __forceinline__ int from_left(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true);
}
__device__ __forceinline__ int from_right(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true);
}
__device__ __forceinline__ int max3(int a, int b, int c) { return max(max(a, b), c); }

__global__ __launch_bounds__(64 * kWaves) void nms_kernel(const int32_t* __restrict__ scores, int w,
                                                          int h, int thr,
                                                          Candidate* __restrict__ cand,
                                                          int cand_cap,
                                                          int32_t* __restrict__ cand_count,
                                                          int strips, int ytiles, int n_images) {
  __shared__ Candidate buf[kLdsCap];
  __shared__ int lds_cnt, lds_base;
  int img, tile;
  xcd_tile(strips * ytiles, n_images, &img, &tile);
  const int ytile = tile / strips;
  const int strip = tile - ytile * strips;
  const int32_t* s = scores + (size_t)img * w * h;
  const int lane = threadIdx.x;
  const int nd = w >> 2;  // 16-byte groups per row
  const int d = strip * kStripLanes + lane;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int ys = (ytile * kWaves + wave) * kRows;  // first centre row of this wave
  const bool last_strip = strip * kStripLanes + 64 >= nd;
  const bool own = d < nd && (strip == 0 || lane >= 1) && (last_strip || lane <= kStripLanes);
  const int dcl = d < nd ? d : nd - 1;
  const int x0 = dcl * 4;
  if (threadIdx.x == 0 && threadIdx.y == 0) lds_cnt = 0;
  __syncthreads();

  if (ys < h) {  // wave-uniform
    const int ye = ys + kRows < h ? ys + kRows : h;
    const int4* rows = reinterpret_cast<const int4*>(s) + dcl;
    auto load_row = [&](int row) -> int4 {
      row = row < 0 ? 0 : (row > h - 1 ? h - 1 : row);
      return rows[(size_t)row * (size_t)nd];
    };
    // horizontal 3-max of a row at this lane's 4 columns, plus the row's edge neighbours
    auto hmax = [&](const int4& r, int hm[4], int& left, int& right) {
      left = from_left(r.w);
      right = from_right(r.x);
      hm[0] = max3(left, r.x, r.y);
      hm[1] = max3(r.x, r.y, r.z);
      hm[2] = max3(r.y, r.z, r.w);
      hm[3] = max3(r.z, r.w, right);
    };
    int4 rt = load_row(ys - 1), rc = load_row(ys), rb;
    int4 nxt = load_row(ys + 1);
    int hmt[4], hmc[4], hmb[4], cl, cr, tl, tr;
    hmax(rt, hmt, tl, tr);
    hmax(rc, hmc, cl, cr);
    for (int y = ys; y < ye; ++y) {
      rb = nxt;
      nxt = load_row(y + 2);
      int bl, br;
      hmax(rb, hmb, bl, br);
      const bool row_ok = y >= 2 && y < h - 2;
      const int c[4] = {rc.x, rc.y, rc.z, rc.w};
      const int lft[4] = {cl, rc.x, rc.y, rc.z};
      const int rgt[4] = {rc.y, rc.z, rc.w, cr};
      bool p[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int x = x0 + i;
        const int nb = max(max3(hmt[i], hmb[i], lft[i]), rgt[i]);
        p[i] = row_ok && x >= 2 && x < w - 2 && c[i] >= thr && nb <= c[i];  // halo lanes too
      }
      // raster-scan parity only matters when two horizontally adjacent pixels both pass
      const int p0_right = from_right(p[0] ? 1 : 0);
      const bool adj = (p[0] && p[1]) || (p[1] && p[2]) || (p[2] && p[3]) || (p[3] && p0_right);
      if (__builtin_expect(__any(adj), 0)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (p[i]) p[i] = accepted_slow(s, w, x0 + i, y, thr);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) p[i] = p[i] && own;  // halo lanes only feed the adjacency test
      unsigned long long b[4];
      int total = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        b[i] = __ballot(p[i]);
        total += __popcll(b[i]);
      }
      if (total) {  // wave-uniform
        int base = 0;
        if (lane == 0) base = atomicAdd(&lds_cnt, total);
        base = __shfl(base, 0);
        const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        int off = base;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (p[i]) {
            const int pos = off + __popcll(b[i] & lt);
            Candidate cd;
            cd.x = x0 + i;
            cd.y = y;
            cd.score = c[i];
            if (pos < kLdsCap) {
              buf[pos] = cd;
            } else {  // LDS staging full: append directly (rare)
              const int gp = atomicAdd(&cand_count[img], 1);
              if (gp < cand_cap) cand[(size_t)img * cand_cap + gp] = cd;
            }
          }
          off += __popcll(b[i]);
        }
      }
      // roll the window
      rt = rc;
      rc = rb;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        hmt[i] = hmc[i];
        hmc[i] = hmb[i];
      }
      tl = cl; tr = cr;
      cl = bl; cr = br;
    }
  }
  __syncthreads();
  const int n = lds_cnt < kLdsCap ? lds_cnt : kLdsCap;
  if (n > 0) {
    if (threadIdx.x == 0 && threadIdx.y == 0) lds_base = atomicAdd(&cand_count[img], n);
    __syncthreads();
    const int tid = threadIdx.y * 64 + threadIdx.x;
    for (int i = tid; i < n; i += 64 * kWaves) {
      const int gp = lds_base + i;
      if (gp < cand_cap) cand[(size_t)img * cand_cap + gp] = buf[i];
    }
  }
}

}  // namespace

void launch_nms(const int32_t* score, int w, int h, int n_images, int abs_threshold,
                Candidate* cand, int cand_cap, int32_t* cand_count, hipStream_t stream) {
  if (n_images <= 0) return;
  const bool aligned = (w % 4 == 0) && ((reinterpret_cast<uintptr_t>(score) & 15) == 0);
  if (aligned) {
    const int nd = w >> 2;
    int strips = 1;
    while ((strips - 1) * kStripLanes + 64 < nd) ++strips;
    const int ytiles = (h + kRows * kWaves - 1) / (kRows * kWaves);
    hipLaunchKernelGGL(nms_kernel, dim3(strips * ytiles * n_images), dim3(64, kWaves, 1), 0, stream,
                       score, w, h, abs_threshold, cand, cand_cap, cand_count, strips, ytiles,
                       n_images);
  } else {
    const dim3 grid((w + 255) / 256, (h + 3) / 4, n_images);
    hipLaunchKernelGGL(nms_generic_kernel, grid, dim3(64, 4, 1), 0, stream, score, w, h,
                       abs_threshold, cand, cand_cap, cand_count);
  }
}

}  // namespace okvfe
