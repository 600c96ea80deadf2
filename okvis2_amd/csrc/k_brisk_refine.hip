// k_brisk_refine.hip -- strongest maxima of a scale-space layer + continuous scale
// (score_type OKVFE_SCORE_BRISK_SCALESPACE = brisk::BriskFeatureDetector(threshold, octaves),
// okvis_cv/test/TestFrame.cpp:71-72).
#include "select_common_dev.h"

namespace okvfe {
namespace {

// ---- published BRISK scale-space detector: strongest maxima of a layer + continuous scale ---------
// (score_type OKVFE_SCORE_BRISK_SCALESPACE = brisk::BriskFeatureDetector(threshold, octaves),
// okvis_cv/test/TestFrame.cpp:71-72; oracle: detect_scale_space with score_type 2.)  Input: the
// layer's candidates after the cross-layer maximum test, sorted (score desc, y, x).  Per kept
// candidate: 2-D sub-pixel fit in the layer, the largest score within +-1 px of the corresponding
// location in the layer below / above (layer 0: the FAST 5-8 map of c0), and the vertex of the
// parabola through the three (relative scale, score) points.  Output in LAYER coordinates with
// size = 12 * relative scale; merge_layers_kernel maps both into the image.
struct ScaleNeighbour {
  const int32_t* map;  // dense score map of the neighbouring layer, null = none
  int w, h, rn, rd;    // its size; scale of this layer / scale of that layer, reduced
};
__device__ __forceinline__ int floor_div_i(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }
__device__ int scale_neighbour_max(const ScaleNeighbour& nb, size_t img_off, int x, int y) {
  const int D = 2 * nb.rd;
  const int Nx = (2 * x + 1) * nb.rn - nb.rd, Ny = (2 * y + 1) * nb.rn - nb.rd;
  int u0 = -floor_div_i(-(Nx - D), D), u1 = floor_div_i(Nx + D, D);
  int v0 = -floor_div_i(-(Ny - D), D), v1 = floor_div_i(Ny + D, D);
  u0 = max(u0, 0);
  v0 = max(v0, 0);
  u1 = min(u1, nb.w - 1);
  v1 = min(v1, nb.h - 1);
  const int32_t* m = nb.map + img_off;
  int best = 0;
  for (int v = v0; v <= v1; ++v)
    for (int u = u0; u <= u1; ++u) best = max(best, m[(size_t)v * nb.w + u]);
  return best;
}
// same operation sequence as orc_scale_refine (oracle/orc_detect.c), FP64, no contraction
__device__ void scale_refine(double rb, bool have_b, int sb, int s, double ra, bool have_a, int sa, double lo, float* rel,
                             float* score) {
  *rel = 1.0f;
  *score = (float)s;
  if (!have_b || !have_a) return;
  const double y0 = (double)sb, y1 = (double)s, y2 = (double)sa;
  double d10 = y1 - y0;
  double h10 = 1.0 - rb;
  d10 = d10 / h10;
  double d21 = y2 - y1;
  double h21 = ra - 1.0;
  d21 = d21 / h21;
  double a = d21 - d10;
  double h20 = ra - rb;
  a = a / h20;
  if (!(a < 0.0)) return;
  double t = 1.0 + rb;
  t = a * t;
  const double b = d10 - t;
  double v = -b;
  double a2 = 2.0 * a;
  v = v / a2;
  v = v < lo ? lo : (v > ra ? ra : v);
  double u = v - rb;
  u = a * u;
  u = d10 + u;
  double dv = v - 1.0;
  u = dv * u;
  u = y1 + u;
  *rel = (float)v;
  *score = (float)u;
}

__global__ __launch_bounds__(256) void brisk_refine_kernel(
    const int32_t* __restrict__ scores, int w, int h, int cand_cap, const int32_t* __restrict__ cand_count,
    const uint64_t* __restrict__ sort_ws, int ws_stride, int max_kpts, ScaleNeighbour below, ScaleNeighbour above,
    double rb, double ra, double lo, okvfe_keypoint* __restrict__ kps, int kp_cap, int32_t* __restrict__ kp_count) {
  const int img = blockIdx.x;
  int n = cand_count[img];
  n = n > cand_cap ? 0 : n;  // overflowed list: no keypoints (okvfe_check_capacity reports it)
  const int kept = min(n, min(max_kpts, kp_cap));
  const uint64_t* keys = sort_ws + (size_t)img * ws_stride;
  const int32_t* sc = scores + (size_t)img * w * h;
  okvfe_keypoint* out = kps + (size_t)img * kp_cap;
  for (int i = threadIdx.x; i < kept; i += 256) {
    const uint64_t k = keys[i];
    const int score = 0x7FFFFFFF - (int32_t)(k >> 32);
    const int v = (int)((k >> 16) & 0xFFFF), u = (int)(k & 0xFFFF);
    int32_t patch[9];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) patch[(dy + 1) * 3 + (dx + 1)] = sc[(size_t)(v + dy) * w + (u + dx)];
    float ddx, ddy;
    subpixel2d(patch, &ddx, &ddy);
    const int sb = below.map ? scale_neighbour_max(below, (size_t)img * below.w * below.h, u, v) : 0;
    const int sa = above.map ? scale_neighbour_max(above, (size_t)img * above.w * above.h, u, v) : 0;
    float rel, resp;
    scale_refine(rb, below.map != nullptr, sb, score, ra, above.map != nullptr, sa, lo, &rel, &resp);
    okvfe_keypoint kp;
    kp.x = (float)u + ddx;
    kp.y = (float)v + ddy;
    kp.size = 12.0f * rel;
    kp.angle = -1.0f;
    kp.response = resp;
    kp.octave = 0;
    kp.class_id = -1;
    out[i] = kp;
  }
  if (threadIdx.x == 0) kp_count[img] = kept;
}

}  // namespace

void launch_brisk_refine(const int32_t* score, int w, int h, int n_images, int cand_cap, const int32_t* cand_count,
                         const uint64_t* sort_ws, int max_kpts, const int32_t* below, int wb, int hb, int rn_b,
                         int rd_b, const int32_t* above, int wa, int ha, int rn_a, int rd_a, double rb, double ra, double lo,
                         okvfe_keypoint* kps, int kp_cap, int32_t* kp_count, hipStream_t stream) {
  if (n_images <= 0) return;
  int ws_stride = 1;
  while (ws_stride < cand_cap) ws_stride <<= 1;
  const ScaleNeighbour nb{below, wb, hb, rn_b, rd_b}, na{above, wa, ha, rn_a, rd_a};
  hipLaunchKernelGGL(brisk_refine_kernel, dim3(n_images), dim3(256), 0, stream, score, w, h, cand_cap, cand_count,
                     sort_ws, ws_stride, max_kpts, nb, na, rb, ra, lo, kps, kp_cap, kp_count);
}

}  // namespace okvfe
