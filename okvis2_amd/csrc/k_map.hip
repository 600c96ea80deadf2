// k_map.hip -- Frontend::matchToMap before its matcher threads start: projection of every landmark
// into the current camera and the descriptor-view pooling (okvis_frontend/src/Frontend.cpp:1219-1359).
//   prepare_landmarks_kernel   one thread per landmark: FoV check through the camera model
//                              (camera_dev.h), 3-D test, view-point / scale pruning, the three-slot
//                              "keep the best" buffer with the reference's exact write / crop rules
//                              (oracle: orc_prepare_landmarks documents the quirks);
//   compact_landmarks_kernel   the 3-D landmarks as the packed set the matcher kernel reads
//                              (projections, <= 2 pooled descriptors each), in landmark order.
// FP64, 3-term sums in the order of okvfe_set_fp64_reduction, no FMA; acos / cos through atan_fixed.h resp. host-computed constants.
#include "camera_dev.h"
#include "okvfe_internal.h"

namespace okvfe {
namespace {

// order of the 3-term sums: as in k_match.hip (okvfe_set_fp64_reduction), this translation unit's copy of the flag
__device__ int g_fp64_tree_map = 1;
__device__ __forceinline__ double sum3m(double p0, double p1, double p2) {
  const bool tree = g_fp64_tree_map != 0;
  const double u = tree ? p1 : p0, v = tree ? p2 : p1, w = tree ? p0 : p2;
  const double s = u + v;
  return tree ? w + s : s + w;
}
__device__ __forceinline__ double dot3m(const double a[3], const double b[3]) {
  const double p0 = a[0] * b[0];
  const double p1 = a[1] * b[1];
  const double p2 = a[2] * b[2];
  return sum3m(p0, p1, p2);
}
__device__ __forceinline__ void normalize3m(const double v[3], double out[3]) {
  const double n = sqrt(dot3m(v, v));
  out[0] = v[0] / n;
  out[1] = v[1] / n;
  out[2] = v[2] / n;
}

__global__ __launch_bounds__(128) void prepare_landmarks_kernel(
    const double* __restrict__ hp_W, const double* __restrict__ quality,
    const int32_t* __restrict__ obs_begin, int n_landmarks, const int32_t* __restrict__ obs_pose,
    const double* __restrict__ obs_bp, const okvfe_pose* __restrict__ poses, okvfe_pose T1,
    const DeviceCamera* __restrict__ camera, int w, int h, double repr, int exclusive, double cos10,
    double cos06, int32_t* __restrict__ status, int32_t* __restrict__ n_desc,
    int32_t* __restrict__ obs_rows, double* __restrict__ projection, double* __restrict__ e_W,
    double* __restrict__ r_W) {
  const int l = blockIdx.x * 128 + threadIdx.x;
  if (l >= n_landmarks) return;
  const DeviceCamera cam = *camera;
  status[l] = 0;
  n_desc[l] = 0;
  obs_rows[3 * l] = obs_rows[3 * l + 1] = obs_rows[3 * l + 2] = -1;
  projection[2 * l] = projection[2 * l + 1] = 0.0;
  for (int i = 0; i < 6; ++i) e_W[6 * l + i] = r_W[6 * l + i] = 0.0;
  const double hp[4] = {hp_W[4 * l], hp_W[4 * l + 1], hp_W[4 * l + 2], hp_W[4 * l + 3]};
  const double p_W[3] = {hp[0] / hp[3], hp[1] / hp[3], hp[2] / hp[3]};
  const double r_Wv[3] = {p_W[0] - T1.r[0], p_W[1] - T1.r[1], p_W[2] - T1.r[2]};
  double e_Wv[3];
  normalize3m(r_Wv, e_Wv);
  const double rn = sqrt(dot3m(r_Wv, r_Wv));
  const double r = 0.01 > rn ? 0.01 : rn;
  // hp_C = T_WC1^-1 * hp_W
  double cr[3], hh[3], hp_C[4];
  for (int i = 0; i < 3; ++i) {
    cr[i] = sum3m(T1.C[i] * T1.r[0], T1.C[3 + i] * T1.r[1], T1.C[6 + i] * T1.r[2]);
    hh[i] = sum3m(T1.C[i] * hp[0], T1.C[3 + i] * hp[1], T1.C[6 + i] * hp[2]);
  }
  for (int i = 0; i < 3; ++i) hp_C[i] = hh[i] + (-cr[i]) * hp[3];
  hp_C[3] = hp[3];
  double head[3], kp[2];
  if (hp_C[3] < 0) {
    head[0] = -hp_C[0]; head[1] = -hp_C[1]; head[2] = -hp_C[2];
  } else {
    head[0] = hp_C[0]; head[1] = hp_C[1]; head[2] = hp_C[2];
  }
  const int st = cam::project(cam, w, h, head, kp);
  if (st == 4 || st == 3) return;  // Invalid, Behind
  const double maxU = (double)w + repr, maxV = (double)h + repr;
  if (kp[0] < -repr || kp[1] < -repr || kp[0] > maxU || kp[1] > maxV) return;
  projection[2 * l] = kp[0];
  projection[2 * l + 1] = kp[1];
  const double focal = cam.fu + cam.fv;
  bool is3d = false;
  int o = 0, rows[3] = {-1, -1, -1};
  double best[3] = {1.0, 1.0, 1.0}, ew[3][3], rw[3][3];
  for (int ob = obs_begin[l]; ob < obs_begin[l + 1]; ++ob) {
    const okvfe_pose& To = poses[obs_pose[ob]];
    const double r_old[3] = {p_W[0] - To.r[0], p_W[1] - To.r[1], p_W[2] - To.r[2]};
    if (!is3d) {
      const double f = 0.2 / focal / quality[l];
      const double rc[3] = {r_Wv[0] - f * r_old[0], r_Wv[1] - f * r_old[1], r_Wv[2] - f * r_old[2]};
      double a[3], b[3];
      normalize3m(r_Wv, a);
      normalize3m(rc, b);
      if (dot3m(a, b) > cos10) is3d = true;
    }
    double eo[3];
    normalize3m(r_old, eo);
    const double cosVC = dot3m(e_Wv, eo);
    if (cosVC < cos06 && !exclusive) continue;
    const double scaleChange = fabs(r - sqrt(dot3m(r_old, r_old))) / r;
    if (scaleChange > 0.5 && !exclusive) continue;
    const double score = 0.5 * (acos_fixed(cosVC) / 0.6 + scaleChange / 0.5);
    double worst = 0.0;
    int wi = 0;
    for (int n = 0; n < 3; ++n)
      if (best[n] > worst) {
        worst = best[n];
        wi = n;
      }
    if (score < best[wi]) {
      const double bpv[3] = {obs_bp[3 * (size_t)ob], obs_bp[3 * (size_t)ob + 1], obs_bp[3 * (size_t)ob + 2]};
      double en[3], ev[3];
      normalize3m(bpv, en);
      ev[0] = dot3m(To.C, en);
      ev[1] = dot3m(To.C + 3, en);
      ev[2] = dot3m(To.C + 6, en);
      // rows / ew / rw are indexed with the run-time value o in {0, 1, 2}
      for (int k = 0; k < 3; ++k)
        if (k == o) {
          rows[k] = ob;
          ew[k][0] = ev[0]; ew[k][1] = ev[1]; ew[k][2] = ev[2];
          rw[k][0] = To.r[0]; rw[k][1] = To.r[1]; rw[k][2] = To.r[2];
        }
      o = o > wi ? o : wi;
      for (int k = 0; k < 3; ++k)
        if (k == wi) best[k] = score;
    }
  }
  if (o == 0) return;
  status[l] = is3d ? 1 : 2;
  n_desc[l] = o;
  for (int k = 0; k < 3; ++k) obs_rows[3 * l + k] = rows[k];
  for (int k = 0; k < o && k < 2; ++k)
    for (int i = 0; i < 3; ++i) {
      e_W[6 * l + 3 * k + i] = ew[k][i];
      r_W[6 * l + 3 * k + i] = rw[k][i];
    }
}

// the landmarks with status == want as a packed set, in landmark order (single workgroup: the
// landmark table of a frame is a few thousand rows)
__global__ __launch_bounds__(1024) void compact_landmarks_kernel(
    const int32_t* __restrict__ status, const int32_t* __restrict__ n_desc,
    const int32_t* __restrict__ obs_rows, const double* __restrict__ projection,
    const uint8_t* __restrict__ obs_desc, int n_landmarks, int want, int32_t* __restrict__ index_out,
    double* __restrict__ proj_out, int32_t* __restrict__ begin_out, uint8_t* __restrict__ pool_out,
    int32_t* __restrict__ n_out /* [0] landmarks, [1] pool rows */) {
  __shared__ int s_lm[1024], s_rows[1024];
  __shared__ int base_lm, base_rows;
  const int tid = threadIdx.x;
  if (tid == 0) base_lm = base_rows = 0;
  __syncthreads();
  for (int c0 = 0; c0 < n_landmarks; c0 += 1024) {
    const int l = c0 + tid;
    const bool take = l < n_landmarks && status[l] == want;
    const int nd = take ? n_desc[l] : 0;
    s_lm[tid] = take ? 1 : 0;
    s_rows[tid] = nd;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // inclusive scans
      const int a = tid >= d ? s_lm[tid - d] : 0, b = tid >= d ? s_rows[tid - d] : 0;
      __syncthreads();
      s_lm[tid] += a;
      s_rows[tid] += b;
      __syncthreads();
    }
    if (take) {
      const int pos = base_lm + s_lm[tid] - 1, row0 = base_rows + s_rows[tid] - nd;
      index_out[pos] = l;
      proj_out[2 * pos] = projection[2 * l];
      proj_out[2 * pos + 1] = projection[2 * l + 1];
      begin_out[pos] = row0;
      for (int k = 0; k < nd; ++k) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(obs_desc + (size_t)obs_rows[3 * l + k] * OKVFE_DESC_BYTES);
        uint32_t* dst = reinterpret_cast<uint32_t*>(pool_out + (size_t)(row0 + k) * OKVFE_DESC_BYTES);
        for (int i = 0; i < 12; ++i) dst[i] = src[i];
      }
    }
    __syncthreads();
    if (tid == 1023) {
      base_lm += s_lm[1023];
      base_rows += s_rows[1023];
    }
    __syncthreads();
  }
  if (tid == 0) {
    begin_out[base_lm] = base_rows;
    n_out[0] = base_lm;
    n_out[1] = base_rows;
  }
}

}  // namespace

void launch_prepare_landmarks(const double* hp_W, const double* quality, const int32_t* obs_begin,
                              int n_landmarks, const int32_t* obs_pose, const double* obs_bp,
                              const okvfe_pose* poses, const okvfe_pose& T_WC1, const DeviceCamera* camera,
                              int w, int h, double repr, int exclusive, double cos10, double cos06,
                              int32_t* status, int32_t* n_desc, int32_t* obs_rows, double* projection,
                              double* e_W, double* r_W, hipStream_t stream) {
  if (n_landmarks <= 0) return;
  hipLaunchKernelGGL(prepare_landmarks_kernel, dim3((n_landmarks + 127) / 128), dim3(128), 0, stream, hp_W,
                     quality, obs_begin, n_landmarks, obs_pose, obs_bp, poses, T_WC1, camera, w, h, repr,
                     exclusive, cos10, cos06, status, n_desc, obs_rows, projection, e_W, r_W);
}
void launch_compact_landmarks(const int32_t* status, const int32_t* n_desc, const int32_t* obs_rows,
                              const double* projection, const uint8_t* obs_desc, int n_landmarks, int want,
                              int32_t* index_out, double* proj_out, int32_t* begin_out, uint8_t* pool_out,
                              int32_t* n_out, hipStream_t stream) {
  hipLaunchKernelGGL(compact_landmarks_kernel, dim3(1), dim3(1024), 0, stream, status, n_desc, obs_rows,
                     projection, obs_desc, n_landmarks, want, index_out, proj_out, begin_out, pool_out, n_out);
}

bool set_fp64_tree_map(int tree) {
  const int v = tree != 0;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_fp64_tree_map), &v, sizeof(v)) == hipSuccess;
}

}  // namespace okvfe
