// k_pyramid.hip -- scale space of the detector for octaves > 0 (gfx950).
//
// brisk::ScaleSpaceFeatureDetector<HarrisScoreCalculator>(uniformityRadius, OCTAVES, absThreshold,
// maxNumKpt) (okvis_frontend/src/Frontend.cpp:2406-2409; the reference's own smoke test passes
// octaves = 2, okvis_cv/test/TestFrame.cpp:75-77) builds 2 * octaves layers; the layer arithmetic is
// in the un-vendored brisk library, so this follows the restatement of oracle/orc_detect.c
// (detect_scale_space; PARITY UNPINNED):
//   layer 1 = two-third sampling of layer 0, layer l >= 2 = half sampling of layer l-2;
//   per layer K1 + K2 (k_harris.hip / k_nms.hip) on the layer image;
//   scale_filter_kernel   a 2-D maximum survives unless a strictly greater score lies within +-1 px
//                         of the corresponding location in the layer below or above;
//   per layer K3 + K4 (k_select.hip);
//   merge_layers_kernel   layer keypoints -> image coordinates, size 12 * scale, octave = layer.
// The samplers are plain streaming kernels (1 B/px in, <= 1 B/px out; a few % of K1's bytes).
#include "okvfe_internal.h"

namespace okvfe {
namespace {

// 2x2 box mean, (a+b+c+d+2)>>2: one thread = 4 output pixels of one row (two aligned 4-byte loads
// per source row when the source width is a multiple of 8, byte loads otherwise)
__global__ __launch_bounds__(256) void halfsample_kernel(const uint8_t* __restrict__ src, int w, int h,
                                                         uint8_t* __restrict__ dst, int w2, int h2) {
  const int img = blockIdx.z;
  const int y = blockIdx.y;
  const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (x0 >= w2) return;
  const uint8_t* r0 = src + (size_t)img * w * h + (size_t)(2 * y) * w + 2 * x0;
  const uint8_t* r1 = r0 + w;
  uint8_t* o = dst + (size_t)img * w2 * h2 + (size_t)y * w2 + x0;
  const int n = min(4, w2 - x0);
  for (int i = 0; i < n; ++i)
    o[i] = (uint8_t)((r0[2 * i] + r0[2 * i + 1] + r1[2 * i] + r1[2 * i + 1] + 2) >> 2);
}

// 3x3 -> 2x2 with the separable weights (2,1,0)/3 | (0,1,2)/3: (sum + 4) / 9.  One thread = one
// 3x3 source block.
__global__ __launch_bounds__(256) void twothird_kernel(const uint8_t* __restrict__ src, int w, int h,
                                                       uint8_t* __restrict__ dst, int bw, int bh) {
  const int img = blockIdx.z;
  const int by = blockIdx.y;
  const int bx = blockIdx.x * 256 + threadIdx.x;
  if (bx >= bw) return;
  const uint8_t* s = src + (size_t)img * w * h + (size_t)(3 * by) * w + 3 * bx;
  int p[3][3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int i = 0; i < 3; ++i) p[j][i] = s[(size_t)j * w + i];
  const int w2 = bw * 2;
  uint8_t* o = dst + (size_t)img * w2 * (bh * 2) + (size_t)(2 * by) * w2 + 2 * bx;
  // horizontal (2,1,0) / (0,1,2) per row, then the same vertically
  int hl[3], hr[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    hl[j] = 2 * p[j][0] + p[j][1];
    hr[j] = p[j][1] + 2 * p[j][2];
  }
  o[0] = (uint8_t)((2 * hl[0] + hl[1] + 4) / 9);
  o[1] = (uint8_t)((2 * hr[0] + hr[1] + 4) / 9);
  o[w2] = (uint8_t)((hl[1] + 2 * hl[2] + 4) / 9);
  o[w2 + 1] = (uint8_t)((hr[1] + 2 * hr[2] + 4) / 9);
}

__device__ __forceinline__ int floor_div(int a, int b) {  // b > 0
  return a >= 0 ? a / b : -((-a + b - 1) / b);
}
// no strictly greater score within +-1 px (other layer's pixels) of the corresponding location;
// rn / rd = scale of this layer / scale of the other one (oracle: orc_scale_neighbour_ok)
__device__ bool neighbour_ok(const int32_t* __restrict__ other, const ScoreLayout& lo, int wo, int ho, int x,
                             int y, int32_t s, int rn, int rd) {
  const int D = 2 * rd;
  const int Nx = (2 * x + 1) * rn - rd, Ny = (2 * y + 1) * rn - rd;
  int u0 = -floor_div(-(Nx - D), D), u1 = floor_div(Nx + D, D);
  int v0 = -floor_div(-(Ny - D), D), v1 = floor_div(Ny + D, D);
  u0 = max(u0, 0);
  v0 = max(v0, 0);
  u1 = min(u1, wo - 1);
  v1 = min(v1, ho - 1);
  for (int v = v0; v <= v1; ++v)
    for (int u = u0; u <= u1; ++u)
      if (other[score_index(lo, u, v)] > s) return false;
  return true;
}

// In-place compaction of one layer's candidate list (one workgroup per image): survivors keep
// their relative order (which is arbitrary anyway: the sort that follows fixes the order).
__global__ __launch_bounds__(256) void scale_filter_kernel(
    Candidate* __restrict__ cand, int cand_cap, int32_t* __restrict__ cand_count,
    const int32_t* __restrict__ below, ScoreLayout lb, int wb, int hb, int rn_b, int rd_b,
    const int32_t* __restrict__ above, ScoreLayout la, int wa, int ha, int rn_a, int rd_a) {
  __shared__ int wave_cnt[4];
  __shared__ int s_base;
  const int img = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  Candidate* c = cand + (size_t)img * cand_cap;
  const int total = cand_count[img];
  const int n = total > cand_cap ? 0 : total;  // an overflowed list is dropped downstream anyway
  const int32_t* sb = below ? below + (size_t)img * lb.pitch * hb : nullptr;
  const int32_t* sa = above ? above + (size_t)img * la.pitch * ha : nullptr;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 256) {
    const int i = base + tid;
    Candidate cd{};
    bool keep = false;
    if (i < n) {
      cd = c[i];
      keep = true;
      if (sb) keep = neighbour_ok(sb, lb, wb, hb, cd.x, cd.y, cd.score, rn_b, rd_b);
      if (keep && sa) keep = neighbour_ok(sa, la, wa, ha, cd.x, cd.y, cd.score, rn_a, rd_a);
    }
    const unsigned long long b = __ballot(keep);
    if (lane == 0) wave_cnt[wv] = __popcll(b);
    __syncthreads();  // all candidates of this chunk are in registers before any slot is rewritten
    int pos = s_base;
    for (int k = 0; k < wv; ++k) pos += wave_cnt[k];
    pos += __popcll(b & ((1ull << lane) - 1ull));
    if (keep) c[pos] = cd;
    __syncthreads();
    if (tid == 0) s_base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (tid == 0 && total <= cand_cap) cand_count[img] = s_base;
}

struct MergeLayers {
  const okvfe_keypoint* kps[8];   // per layer [n_images][layer_cap]
  const int32_t* counts[8];       // per layer [n_images]
  float scale[8];
  int n_layers, layer_cap;
};
// layers in ascending order -> out[img][0 .. sum), image coordinates X = s (x + 1/2) - 1/2
__global__ __launch_bounds__(256) void merge_layers_kernel(MergeLayers m, okvfe_keypoint* __restrict__ out,
                                                           int out_cap, int32_t* __restrict__ out_count) {
  const int img = blockIdx.x;
  int off = 0;
  for (int l = 0; l < m.n_layers; ++l) {
    const int n = m.counts[l][img];
    const okvfe_keypoint* src = m.kps[l] + (size_t)img * m.layer_cap;
    const float s = m.scale[l];
    for (int i = threadIdx.x; i < n; i += 256) {
      okvfe_keypoint k = src[i];
      float t = k.x + 0.5f;
      t = s * t;
      k.x = t - 0.5f;
      t = k.y + 0.5f;
      t = s * t;
      k.y = t - 0.5f;
      k.size = k.size * s;  // layer keypoints carry 12 (x the relative scale of the BRISK scale space)
      k.octave = l;
      if (off + i < out_cap) out[(size_t)img * out_cap + off + i] = k;
    }
    off += n;
  }
  if (threadIdx.x == 0) out_count[img] = off < out_cap ? off : out_cap;
}

}  // namespace

void launch_halfsample(const uint8_t* src, int w, int h, int n_images, uint8_t* dst, hipStream_t stream) {
  const int w2 = w / 2, h2 = h / 2;
  if (n_images <= 0 || w2 <= 0 || h2 <= 0) return;
  hipLaunchKernelGGL(halfsample_kernel, dim3((w2 + 1023) / 1024, h2, n_images), dim3(256), 0, stream, src, w,
                     h, dst, w2, h2);
}
void launch_twothird(const uint8_t* src, int w, int h, int n_images, uint8_t* dst, hipStream_t stream) {
  const int bw = w / 3, bh = h / 3;
  if (n_images <= 0 || bw <= 0 || bh <= 0) return;
  hipLaunchKernelGGL(twothird_kernel, dim3((bw + 255) / 256, bh, n_images), dim3(256), 0, stream, src, w, h,
                     dst, bw, bh);
}
void launch_scale_filter(Candidate* cand, int cand_cap, int32_t* cand_count, int n_images,
                         const int32_t* below, ScoreLayout lb, int wb, int hb, int rn_b, int rd_b,
                         const int32_t* above, ScoreLayout la, int wa, int ha, int rn_a, int rd_a,
                         hipStream_t stream) {
  if (n_images <= 0) return;
  hipLaunchKernelGGL(scale_filter_kernel, dim3(n_images), dim3(256), 0, stream, cand, cand_cap, cand_count,
                     below, lb, wb, hb, rn_b, rd_b, above, la, wa, ha, rn_a, rd_a);
}
void launch_merge_layers(const okvfe_keypoint* const* kps, const int32_t* const* counts, const float* scale,
                         int n_layers, int layer_cap, int n_images, okvfe_keypoint* out, int out_cap,
                         int32_t* out_count, hipStream_t stream) {
  if (n_images <= 0) return;
  MergeLayers m{};
  for (int l = 0; l < n_layers; ++l) {
    m.kps[l] = kps[l];
    m.counts[l] = counts[l];
    m.scale[l] = scale[l];
  }
  m.n_layers = n_layers;
  m.layer_cap = layer_cap;
  hipLaunchKernelGGL(merge_layers_kernel, dim3(n_images), dim3(256), 0, stream, m, out, out_cap, out_count);
}

}  // namespace okvfe
